"""f16-MLP mode: GPU vs oracle(f16 emulation) closeness; f32 mode must be unaffected."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from sde4mbrl_px4_amd import MPCConfig, synthetic_iris, workload as W
from sde4mbrl_px4_amd.solver import SdeMpcSolver
import orc
for (H, P, mi) in [(12, 40, 8), (50, 128, 10)]:
    cfg = MPCConfig(horizon=H, num_short_dt=H, num_particles=P, u_slew_coeff=1.0, max_iter=mi, max_no_improvement_iter=mi, mlp_dtype="f16")
    model = synthetic_iris(); B = 3
    x0 = W.random_initial_states(B, 0); xref = np.stack([W.reference_window(0.1 * b, cfg.time_steps) for b in range(B)]); noise = W.make_noise(B, P, H, 0)
    u = np.clip(0.71 + 0.1 * np.random.default_rng(1).standard_normal((B, H, 4)), 1e-4, 1).astype(np.float32)
    S = SdeMpcSolver(cfg, model, max_batch=B); O = orc.Oracle(cfg, model); O32 = orc.Oracle(cfg.replace(mlp_dtype="f32"), model)
    cost, traj, xm = S.rollout(x0, u, xref, noise, True, True); gc, g = S.grad(x0, u, xref, noise)
    cost2, _, _ = S.rollout(x0, u, xref, noise, False, False)
    u0 = np.tile(np.float32(0.71), (B, H, 4)); uopt, xevol, info = S.solve(x0, xref, noise, u0, np.full(B, 0.01, np.float32))
    for b in range(B):
        c, t, m_ = O.rollout(x0[b], u[b], xref[b], noise[b], True, True); c32 = O32.rollout(x0[b], u[b], xref[b], noise[b])[0]
        cg, gg = O.grad(x0[b], u[b], xref[b], noise[b])
        uo, xe, inf, _ = O.solve(x0[b], xref[b], noise[b], u0[b], 0.01)
        print(f"H{H} P{P} b{b}: cost gpu {cost[b]:.6f} orc16 {c:.6f} (rel {abs(cost[b]-c)/c:.2e}; f32-oracle rel {abs(cost[b]-c32)/c32:.2e}) traj maxabs {np.abs(traj[b]-t).max():.2e} "
              f"grad rel {np.abs(g[b]-gg).max()/np.abs(gg).max():.2e} | solve: N_it {info[b,2]:.0f}/{inf[2]:.0f} opt {info[b,6]:.5f}/{inf[6]:.5f} uopt maxabs {np.abs(uopt[b]-uo).max():.2e}")
    print("   deterministic:", np.array_equal(cost, cost2), " gradcost==rollout cost:", np.array_equal(gc, cost))
    S.close()
