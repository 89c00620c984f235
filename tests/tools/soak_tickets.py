"""Randomised soak of the persistent duo launches at LARGE batches: ticketed launches (>= three instances per team: instances handed out in order of
completion) against striped launches of slices of the same batch (bit for bit, every instance) and against the CPU oracle (sampled instances).
Random team shapes: two-wave teams (P <= 128), four-wave teams (P > 128), odd group counts, ragged particle counts, short and long horizons,
control table in LDS or global memory, fp32 only (the exact path)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from sde4mbrl_px4_amd import MPCConfig, synthetic_multirotor, prng, workload as W
from sde4mbrl_px4_amd.solver import SdeMpcSolver
import orc
from cases import bits_differ
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0; t0 = time.time(); names = {}
for it in range(n):
    rng = np.random.default_rng(7000 + seed0 + it)
    m = int(rng.choice([2, 4, 6, 8])); H = int(rng.choice([3, 8, 20, 50, 90, 140, 200])); P = int(rng.choice([33, 64, 70, 128, 130, 200, 256]))
    cfg = MPCConfig(horizon=H, num_short_dt=int(rng.integers(0, H + 1)), num_particles=P, input_id=list(range(m)), input_bound=[[1e-4, 1.0]] * m, uref=[0.6] * m,
                    res_mult=float(rng.choice([0.0, 0.01])), u_slew_coeff=float(rng.choice([0.0, 1.0])), max_iter=int(rng.integers(1, 4)), max_no_improvement_iter=3,
                    ls_init_stepsize=float(rng.choice([0.01, 1e-4])), ls_maxls=int(rng.integers(1, 5)))
    model = synthetic_multirotor(m, seed=it)
    opts = {"ustg": 1} if rng.random() < 0.3 else {}
    cus = 256
    B = int(3 * 6 * cus + rng.integers(0, 700))            # >= three rounds for every team shape (at most 6 teams per CU)
    x0 = W.random_initial_states(B, 50 + it); xref = np.stack([W.reference_window(0.2 * (b % 40), cfg.time_steps) for b in range(B)])
    keys = prng.split(prng.PRNGKey(100 + it), B)
    u = np.clip(0.6 + 0.1 * rng.standard_normal((B, H, m)), 1e-4, 1).astype(np.float32)
    s0 = np.full(B, cfg.ls_init_stepsize, np.float32)
    S = SdeMpcSolver(cfg, model, max_batch=B, options=opts)
    S.work_counters(reset=True)
    uo, xe, info = S.solve_keys(x0, xref, keys, u, s0)
    kname = S.last_kernel_name().split("sdempc_solve_kernel")[-1]; names[kname] = names.get(kname, 0) + 1
    nb = int(S.work_counters()[0] != B)
    for lo in (0, B // 2 - 300, B - 700):                  # striped launches (fewer than three rounds) of slices of the same batch
        sl = slice(lo, lo + 700)
        u2, x2, i2 = S.solve_keys(x0[sl], xref[sl], keys[sl], u[sl], s0[sl])
        nb += bits_differ(uo[sl], u2) + bits_differ(xe[sl], x2) + bits_differ(info[sl], i2)
    O = orc.Oracle(cfg, model)
    for b in (int(rng.integers(0, B)), B - 1):
        a, c, d, _ = O.solve(x0[b], xref[b], orc.noise_from_key(keys[b], P, H), u[b], float(s0[b]))
        nb += bits_differ(uo[b], a) + bits_differ(xe[b], c) + bits_differ(info[b], d)
    if nb: print(f"MISMATCH case {seed0 + it}: m={m} H={H} P={P} B={B} opts={opts} kernel={kname} words={nb}"); bad += 1
    S.close()
print(f"ticket soak: {n} configurations, {bad} with mismatches, {time.time() - t0:.1f} s; kernels {names}")
sys.exit(1 if bad else 0)
