"""GPU gradient against the oracle, entry by entry, for the matrix-pipe contraction modes in math_mode fast (how the wrong F16 template argument of round 5 was found:
the ratio pattern said which part of the adjoint was missing).  usage: python tests/tools/grad_diag.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, orc
from sde4mbrl_px4_amd import MPCConfig, synthetic_iris
from sde4mbrl_px4_amd.solver import SdeMpcSolver
import test_gpu_parity as T
for mlp in ("f32x3", "f16"):
    for P in (70, 32):
        cfg = MPCConfig(horizon=6, num_short_dt=3, long_step_dt=0.1, num_particles=P, u_slew_coeff=1.0, max_iter=2, max_no_improvement_iter=2, mlp_dtype=mlp, math_mode="fast")
        model = synthetic_iris()
        x0, xref, noise, u = T._problem(cfg, 2, 11)
        S = SdeMpcSolver(cfg, model, max_batch=2, options={"coop": 0})
        gc, g = S.grad(x0, u, xref, noise)
        c, go = orc.Oracle(cfg, model).grad(x0[0], u[0], xref[0], noise[0])
        print(mlp, P, S.last_kernel_name()[-60:], "cost equal", gc[0] == np.float32(c))
        print("  gpu ", g[0].ravel()[:8]); print("  orc ", go.astype(np.float32).ravel()[:8])
        print("  ratio", (g[0] / go).ravel()[:12])
        S.close()
