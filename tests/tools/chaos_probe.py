"""How far do full 200-iteration solves drift under ~1e-7 arithmetic perturbations?
GPU f16-MLP mode vs the oracle's emulation of it (identical except for the MFMA summation order)."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from sde4mbrl_px4_amd import load_mpc_config, synthetic_iris, workload as W
from sde4mbrl_px4_amd.solver import SdeMpcSolver
import orc
cfg = load_mpc_config(os.path.join(ROOT, "configs", "c2_iris_traj_h50_p128.yaml")).replace(mlp_dtype="f16")
B = 16
model = synthetic_iris()
x0 = W.random_initial_states(B, 0); xref = np.stack([W.reference_window(0.05 * b, cfg.time_steps) for b in range(B)]); noise = W.make_noise(B, 128, 50, 0)
u0 = np.tile(np.float32(0.71), (B, 50, 4)); s0 = np.full(B, 0.01, np.float32)
S = SdeMpcSolver(cfg, model, max_batch=B)
uopt, xevol, info = S.solve(x0, xref, noise, u0, s0)
O = [orc.Oracle(cfg, model) for _ in range(B)]
res = [None] * B
def w(i): res[i] = O[i].solve(x0[i], xref[i], noise[i], u0[i], 0.01, trace_cap=200)
t = time.time(); th = [threading.Thread(target=w, args=(i,)) for i in range(B)]; [x.start() for x in th]; [x.join() for x in th]
print("oracle time %.1f s" % (time.time() - t))
for b in range(B):
    uo, xe, inf, tr = res[b]
    du = np.abs(uopt[b] - uo).max(); rel = np.abs(uopt[b] - uo).max() / np.abs(uo).max()
    print(f"inst {b}: N_ls gpu {info[b,7]:.0f} orc {inf[7]:.0f}  opt_cost gpu {info[b,6]:.6f} orc {inf[6]:.6f} (rel {abs(info[b,6]-inf[6])/inf[6]:.1e})  max|du| {du:.2e}  xevol max abs {np.abs(xevol[b]-xe).max():.2e}")
