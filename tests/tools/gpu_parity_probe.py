"""Quick GPU-vs-oracle parity probe (development aid; the real tests live in tests/)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from sde4mbrl_px4_amd import MPCConfig, synthetic_iris, synthetic_hexa
from sde4mbrl_px4_amd import workload as W
from sde4mbrl_px4_amd.solver import SdeMpcSolver
import orc

def bits(a, b):
    a = np.ascontiguousarray(a, np.float32).view(np.uint32); b = np.ascontiguousarray(b, np.float32).view(np.uint32)
    return int((a != b).sum())

def run(H, P, m, B, mi, slew_constr=False):
    model = synthetic_iris() if m == 4 else synthetic_hexa()
    kw = dict(horizon=H, num_short_dt=H, num_particles=P, u_slew_coeff=1.0, max_iter=mi, max_no_improvement_iter=mi)
    if m == 6:
        kw.update(input_id=list(range(6)), input_bound=[[1e-4, 1.0]] * 6, uref=[0.42] * 6)
    if slew_constr:
        kw.update(u_slew_constr=[[-18, 0.07], [-26, 0.32], [-29, 0.2], [-10, 0.25]][:m] + [[-10, 0.1]] * (m - 4), u_slew_constr_coeff=10.0)
    cfg = MPCConfig(**kw)
    O = orc.Oracle(cfg, model)
    S = SdeMpcSolver(cfg, model, max_batch=B)
    x0 = W.random_initial_states(B, 0)
    xref = np.stack([W.reference_window(0.1 * b, cfg.time_steps) for b in range(B)])
    noise = W.make_noise(B, P, H, 0)
    rng = np.random.default_rng(1)
    u = (np.array(cfg.uref, np.float32) + 0.1 * rng.standard_normal((B, H, m))).astype(np.float32)
    u = np.clip(u, 1e-4, 1.0)
    cost, traj, mean = S.rollout(x0, u, xref, noise, want_traj=True, want_mean=True)
    oc = []; nb_t = nb_m = 0
    for b in range(B):
        c, t, mm = O.rollout(x0[b], u[b], xref[b], noise[b], True, True)
        oc.append(c); nb_t += bits(traj[b], t); nb_m += bits(mean[b], mm)
    print(f"[H{H} P{P} m{m} B{B}] rollout: cost mismatched bits {bits(cost, np.array(oc, np.float32))}/{B}, traj {nb_t}, xmean {nb_m}; cost0 gpu {cost[0]} orc {oc[0]}")
    cg, g = S.grad(x0, u, xref, noise)
    nb_g = 0; og = []
    for b in range(B):
        c, gg = O.grad(x0[b], u[b], xref[b], noise[b]); og.append(c); nb_g += bits(g[b], gg.astype(np.float32))
        if b == 0: rel = np.abs(g[b] - gg).max() / np.abs(gg).max()
    print(f"   grad: cost bits {bits(cg, np.array(og, np.float32))}, grad mismatched {nb_g}/{g.size}, rel err inst0 {rel:.3e}")
    u0 = np.tile(np.array(cfg.uref, np.float32), (B, H, 1))
    t0 = time.time(); uopt, xevol, info = S.solve(x0, xref, noise, u0, np.full(B, cfg.ls_init_stepsize, np.float32)); tg = time.time() - t0
    nb_u = nb_x = nb_i = 0
    t0 = time.time()
    for b in range(B):
        uo, xe, inf, _ = O.solve(x0[b], xref[b], noise[b], u0[b], cfg.ls_init_stepsize)
        nb_u += bits(uopt[b], uo); nb_x += bits(xevol[b], xe); nb_i += bits(info[b], inf)
        if b == 0: print("   info gpu", info[0], "\n   info orc", inf)
    tc = time.time() - t0
    print(f"   solve: uopt mismatched {nb_u}/{uopt.size}, xevol {nb_x}/{xevol.size}, info {nb_i}/{info.size}; gpu {tg:.3f}s cpu-oracle {tc:.3f}s")
    S.close()
    return nb_t + nb_m + nb_g + nb_u + nb_x + nb_i

if __name__ == "__main__":
    bad = 0
    bad += run(8, 32, 4, 2, 5)
    bad += run(20, 32, 4, 3, 12, slew_constr=True)
    bad += run(10, 1, 4, 2, 6)
    bad += run(12, 70, 6, 2, 6)
    bad += run(50, 128, 4, 2, 10)
    bad += run(16, 300, 4, 1, 4)
    print("TOTAL MISMATCHED WORDS:", bad)
    sys.exit(1 if bad else 0)
