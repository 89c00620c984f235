"""Prints both sides of the one soak configuration whose solve hits a non-finite gradient (m=1, H=55)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from sde4mbrl_px4_amd import MPCConfig, synthetic_multirotor, workload as W
from sde4mbrl_px4_amd.solver import SdeMpcSolver
import orc
m, H, P, B, it = 1, 55, 32, 2, 495
cfg = MPCConfig(horizon=H, num_short_dt=6, short_step_dt=0.05, long_step_dt=0.1, num_particles=P, input_id=[0], input_bound=[[1e-4, 1.0]], uref=[0.55], u_slew_coeff=1.0,
                max_iter=1, max_no_improvement_iter=6, ls_maxls=5)
model = synthetic_multirotor(m, seed=it)
x0 = W.random_initial_states(B, 1000 + it); xref = np.stack([W.reference_window(0.2 * b, cfg.time_steps) for b in range(B)]); noise = W.make_noise(B, P, H, it)
rng = np.random.default_rng(it); u = np.clip(0.55 + 0.15 * rng.standard_normal((B, H, m)), 1e-4, 1).astype(np.float32)
S = SdeMpcSolver(cfg, model, max_batch=B); O = orc.Oracle(cfg, model)
gc, g = S.grad(x0, u, xref, noise); uopt, xe, info = S.solve(x0, xref, noise, u, np.full(B, 0.01, np.float32))
b = 1
c2, g2 = O.grad(x0[b], u[b], xref[b], noise[b]); uo, xo, io, _ = O.solve(x0[b], xref[b], noise[b], u[b], 0.01)
np.set_printoptions(linewidth=200, precision=5)
print("cost gpu/orc", gc[b], c2); print("grad gpu", g[b].ravel()[:12]); print("grad orc", np.asarray(g2).ravel()[:12])
print("uopt gpu", uopt[b].ravel()[:12]); print("uopt orc", uo.ravel()[:12]); print("u in   ", u[b].ravel()[:12])
print("info gpu", info[b]); print("info orc", io)
