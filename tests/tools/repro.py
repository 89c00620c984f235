import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from sde4mbrl_px4_amd import MPCConfig, synthetic_multirotor, workload as W
from sde4mbrl_px4_amd.solver import SdeMpcSolver
import orc
from cases import bits_differ
def run(m, H, P, B, it, **extra):
    kw = dict(horizon=H, num_short_dt=6, short_step_dt=0.05, long_step_dt=0.1, num_particles=P, input_id=list(range(m)), input_bound=[[1e-4, 1.0]] * m, uref=[0.55] * m,
              u_slew_coeff=1.0, max_iter=1, max_no_improvement_iter=6, ls_maxls=5)
    kw.update(extra)
    cfg = MPCConfig(**kw); model = synthetic_multirotor(m, seed=it)
    x0 = W.random_initial_states(B, 1000 + it); xref = np.stack([W.reference_window(0.2 * b, cfg.time_steps) for b in range(B)]); noise = W.make_noise(B, P, H, it)
    rng = np.random.default_rng(it); u = np.clip(0.55 + 0.15 * rng.standard_normal((B, H, m)), 1e-4, 1).astype(np.float32)
    S = SdeMpcSolver(cfg, model, max_batch=B); O = orc.Oracle(cfg, model)
    cost, traj, xm = S.rollout(x0, u, xref, noise, True, True); gc, g = S.grad(x0, u, xref, noise)
    uopt, xe, info = S.solve(x0, xref, noise, u, np.full(B, 0.01, np.float32))
    out = []
    for b in range(B):
        c, t, mm = O.rollout(x0[b], u[b], xref[b], noise[b], True, True); c2, g2 = O.grad(x0[b], u[b], xref[b], noise[b])
        uo, xo, io, _ = O.solve(x0[b], xref[b], noise[b], u[b], 0.01)
        gd = (np.ascontiguousarray(g[b]).view(np.uint32) != g2.astype(np.float32).view(np.uint32)) & ~(np.isnan(g[b]) & np.isnan(g2))
        out.append(dict(cost=bits_differ(cost[b], c), traj=bits_differ(traj[b], t), xm=bits_differ(xm[b], mm), gcost=bits_differ(gc[b], c2),
                        grad=int(gd.sum()), grad_rows=np.nonzero(gd.any(axis=1))[0][:8].tolist(), uopt=bits_differ(uopt[b], uo), xe=bits_differ(xe[b], xo), info=bits_differ(info[b], io)))
    print(f"m={m} H={H} P={P} B={B}:", out)
    S.close()
for (m, H, P) in [(1, 55, 100), (1, 55, 32), (1, 34, 100), (1, 70, 65), (2, 55, 100), (1, 60, 64), (1, 52, 100), (1, 50, 100), (4, 70, 100)]:
    run(m, H, P, 2, 495)
