"""Randomised soak: GPU (C ABI) vs CPU oracle, bit for bit, over many random configurations. Every configuration also draws one of the
execution layouts / kernel instantiations (handle options of include/sdempc.h, SDEMPC_OPT_*): all must give the same bits."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from sde4mbrl_px4_amd import MPCConfig, synthetic_multirotor, workload as W
from sde4mbrl_px4_amd.solver import SdeMpcSolver
import orc
from cases import bits_differ
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0; nonfinite = 0; t0 = time.time()
LAYOUTS = [("auto", {}), ("coop", {"spec": 0}), ("coop-rt", {"coop_launch": 1}), ("auto-fence", {"coop_fence": 1}), ("coop-fence", {"spec": 0, "coop_fence": 1}),
           ("tile-pk", {"coop": 0, "pk": 1}), ("tile", {"coop": 0, "pk": 0}), ("tile-duo", {"coop": 0, "pk": 0, "duo": 1}), ("tile-gtab", {"coop": 0, "pk": 0, "ustg": 1, "duo": 1}),
           ("tile-nolane", {"coop": 0, "lane": 0, "pk": 0}), ("tile-noduo", {"coop": 0, "pk": 0, "duo": 0}), ("tile-noduo-gtab", {"coop": 0, "pk": 0, "duo": 0, "ustg": 1})]
from sde4mbrl_px4_amd import _abi
if not (_abi.load_library().sdempc_build_flags() & 1):       # default build: no packed-tanh instantiations (a layout is drawn by index: keep the list's length, so the draws of a case number do not move)
    LAYOUTS = [(n_, ({"coop": 0, "pk": 0} if n_ == "tile-pk" else o_)) for n_, o_ in LAYOUTS]
MLPS = (sys.argv[3].split(",") if len(sys.argv) > 3 else ["f32", "f32", "f16", "f32x3", "f32x3"])     # contraction modes to draw from (SPEC.md 9, 9b): all bit-exact
FAST_SHARE = float(sys.argv[4]) if len(sys.argv) > 4 else 0.4
DEBUG = bool(os.environ.get("SOAK_DEBUG"))          # which outputs differ, per instance
ONLY = set(int(x) for x in os.environ.get("SOAK_ONLY", "").split(",") if x)      # case numbers to run (the others are skipped; draws unchanged)
used = {}
for it in range(n):
    rng = np.random.default_rng(seed0 + it)
    m = int(rng.integers(1, 9)); H = int(rng.choice([1, 2, 3, 5, 8, 13, 21, 34, 55, 70])); P = int(rng.choice([1, 2, 7, 31, 32, 33, 64, 65, 100, 128, 130, 257]))
    kw = dict(horizon=H, num_short_dt=int(rng.integers(0, H + 1)), short_step_dt=float(rng.choice([0.01, 0.05])), long_step_dt=float(rng.choice([0.05, 0.1])),
              num_particles=P, discount=float(rng.choice([1.0, 0.95])), input_id=list(range(m)), input_bound=[[1e-4, 1.0]] * m, uref=[float(rng.uniform(0.3, 0.8))] * m,
              uerr=float(rng.uniform(0, 2)), perr=list(rng.uniform(0, 200, 3)), verr=list(rng.uniform(0, 10, 3)), qerr=list(rng.uniform(0, 100, 3)), werr=list(rng.uniform(0, 3, 3)),
              res_mult=float(rng.choice([0.0, 0.01, 0.5])), u_slew_coeff=float(rng.choice([0.0, 1.0])), max_iter=int(rng.integers(0, 7)), max_no_improvement_iter=int(rng.integers(1, 7)),
              beta_init=float(rng.uniform(0, 0.9)), rtol=float(rng.choice([1e-6, 1e-3])), ls_init_stepsize=float(rng.choice([0.01, 1e-4])),
              ls_max_stepsize=float(rng.choice([1.0, 1e-3])), ls_coef=float(rng.choice([0.01, 0.3])), ls_decrease_factor=float(rng.uniform(0.2, 0.9)),
              ls_increase_factor=float(rng.uniform(1.0, 3.0)), ls_maxls=int(rng.integers(0, 6)), stepsize=1e-4, ls_reset_option=str(rng.choice(["increase", "conservative"])),
              enforce_ubound=bool(rng.random() < 0.85), mlp_dtype=str(rng.choice(MLPS)))
    if rng.random() < 0.4: kw.update(u_slew_constr=[[-float(rng.uniform(0.01, 0.1)), float(rng.uniform(0.01, 0.1))]] * m, u_slew_constr_coeff=float(rng.uniform(1, 20)))
    if rng.random() < 0.3: kw.update(moment_scale=float(rng.uniform(0.1, 1.0)))
    if rng.random() < 0.3:
        ids = sorted(int(i) for i in rng.choice(13, size=int(rng.integers(1, 7)), replace=False))
        kw.update(state_id=ids, state_penalty=[float(rng.uniform(0.1, 30)) for _ in ids], constr_pen=float(rng.choice([1.0, 0.1])),
                  state_bound=[[-float(rng.uniform(0.05, 1.0)), float(rng.uniform(0.05, 1.0))] for _ in ids])
    if FAST_SHARE and rng.random() < FAST_SHARE: kw.update(math_mode="fast")          # SPEC.md 10: the hardware-instruction arithmetic, checked against oracle/transc_model.c
    if ONLY and (seed0 + it) not in ONLY: continue
    cfg = MPCConfig(**kw); model = synthetic_multirotor(m, seed=it)
    lname, lopts = LAYOUTS[int(rng.integers(0, len(LAYOUTS)))]
    if cfg.mlp_dtype != "f32":
        lname += "/" + cfg.mlp_dtype          # (the lane / cooperative layouts and the packed-tanh instantiation exist for the f32 chain: the library falls back to tiles by itself)
    if cfg.math_mode != "exact": lname += "/" + cfg.math_mode
    used[lname] = used.get(lname, 0) + 1
    B = int(rng.integers(1, 6))
    x0 = W.random_initial_states(B, 1000 + it); xref = np.stack([W.reference_window(0.2 * b, cfg.time_steps) for b in range(B)]); noise = W.make_noise(B, P, H, it)
    u = np.clip(np.asarray(cfg.uref, np.float32) + 0.15 * rng.standard_normal((B, H, m)), 1e-4, 1).astype(np.float32)
    S = SdeMpcSolver(cfg, model, max_batch=B, options=lopts); O = orc.Oracle(cfg, model)
    cost, traj, xm = S.rollout(x0, u, xref, noise, True, True); gc, g = S.grad(x0, u, xref, noise)
    uopt, xe, info = S.solve(x0, xref, noise, u, np.full(B, cfg.ls_init_stepsize if cfg.ls_maxls else cfg.stepsize, np.float32))
    keys = rng.integers(0, 2 ** 32, size=(B, 2), dtype=np.uint32)
    nk = S.noise_from_keys(keys)
    uk, xk, ik = S.solve_keys(x0, xref, keys, u, np.full(B, cfg.ls_init_stepsize if cfg.ls_maxls else cfg.stepsize, np.float32))
    nb = 0; nonfinite += int(not (np.isfinite(traj).all() and np.isfinite(g).all()))
    for b in range(B):
        c, t, mm = O.rollout(x0[b], u[b], xref[b], noise[b], True, True); c2, g2 = O.grad(x0[b], u[b], xref[b], noise[b])
        uo, xo, io, _ = O.solve(x0[b], xref[b], noise[b], u[b], float(cfg.ls_init_stepsize if cfg.ls_maxls else cfg.stepsize))
        parts = dict(cost=bits_differ(cost[b], c), traj=bits_differ(traj[b], t), xmean=bits_differ(xm[b], mm), grad_cost=bits_differ(gc[b], c2), grad=bits_differ(g[b], g2.astype(np.float32)),
                     uopt=bits_differ(uopt[b], uo), xevol=bits_differ(xe[b], xo), info=bits_differ(info[b], io))
        nb += sum(parts.values())
        if DEBUG and sum(parts.values()):
            print(f"   case {seed0+it} instance {b}: differing words {parts}", flush=True)
            if parts["info"]: print(f"      info GPU {info[b]}\n      info CPU {np.asarray(io)}\n      kernel {S.last_kernel_name()}", flush=True)
        nzo = orc.noise_from_key(keys[b], P, H)
        uo2, xo2, io2, _ = O.solve(x0[b], xref[b], nzo, u[b], float(cfg.ls_init_stepsize if cfg.ls_maxls else cfg.stepsize))
        nb += bits_differ(nk[b], nzo) + bits_differ(uk[b], uo2) + bits_differ(xk[b], xo2) + bits_differ(ik[b], io2)
    if nb: print(f"MISMATCH case {seed0+it} [{lname}]: m={m} H={H} P={P} B={B} words={nb} cfg={kw}", flush=True); bad += 1
    if it % 50 == 49: print(f"  ... {it + 1} configurations, {bad} with mismatches, {time.time()-t0:.0f} s", flush=True)
    S.close()
print(f"soak: {n} configurations ({nonfinite} with non-finite trajectories), {bad} with mismatches, {time.time()-t0:.1f} s; layouts {used}")
sys.exit(1 if bad else 0)
