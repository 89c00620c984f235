"""How far do the tolerance-parity modes (math_mode: fast, mlp_dtype: f16; SPEC.md 9-10) drift from the bit-reproducible f32 path over a
FULL solve (C2: 200 APG iterations, ~375 line-search decisions)?  usage: python tools/mode_drift.py [--batch 64] [--config ...]
Prints, per mode, the deviation of the optimal controls / predicted trajectory / optimal cost from the exact mode's on identical inputs."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from sde4mbrl_px4_amd import load_mpc_config, synthetic_iris, synthetic_hexa, prng
from sde4mbrl_px4_amd import workload as W
from sde4mbrl_px4_amd.solver import SdeMpcSolver

ap = argparse.ArgumentParser()
ap.add_argument("--config", default=os.path.join(ROOT, "configs", "c2_iris_traj_h50_p128.yaml"))
ap.add_argument("--batch", type=int, default=64)
a = ap.parse_args()
base = load_mpc_config(a.config)
B, H, m = a.batch, base.horizon, base.num_motors
model = synthetic_iris() if m == 4 else synthetic_hexa()
x0 = W.random_initial_states(B, 0)
xref = np.stack([W.reference_window(0.05 * (b % 160), base.time_steps) for b in range(B)])
keys = prng.split(prng.PRNGKey(10), B)
res = {}
for name, kw in (("exact", {}), ("fast", dict(math_mode="fast")), ("f16", dict(mlp_dtype="f16")), ("fast+f16", dict(math_mode="fast", mlp_dtype="f16"))):
    cfg = base.replace(**kw)
    S = SdeMpcSolver(cfg, model, max_batch=B)
    yk, i0 = S.reset()
    res[name] = S.solve_keys(x0, xref, keys, np.tile(yk[None], (B, 1, 1)), np.full(B, i0["stepsize"], np.float32))
    S.close()
ue, xe, ie = res["exact"]
print(f"{os.path.basename(a.config)}: B={B}, N_it {ie[:, 2].mean():.0f}, N_ls {ie[:, 7].mean():.0f}; controls in [{ue.min():.3f}, {ue.max():.3f}]")
for name in ("fast", "f16", "fast+f16"):
    u, x, i = res[name]
    du = np.abs(u - ue).reshape(B, -1).max(1)
    dc = np.abs(i[:, 6] - ie[:, 6]) / np.abs(ie[:, 6])
    dx = np.abs(x - xe).reshape(B, -1).max(1)
    within = np.mean(np.all(np.abs(u - ue) <= 1e-4 + 1e-4 * np.abs(ue), axis=(1, 2)))
    print(f"{name:9s}: max|du| median {np.median(du):.2e} worst {du.max():.2e}; first control max|du0| {np.abs(u[:, 0] - ue[:, 0]).max():.2e}; "
          f"max|dxevol| median {np.median(dx):.2e} worst {dx.max():.2e}; opt_cost rel median {np.median(dc):.2e} worst {dc.max():.2e}; "
          f"N_ls differs in {np.mean(i[:, 7] != ie[:, 7]) * 100:.0f} % of instances; within 1e-4 (abs+rel) on all controls: {within * 100:.0f} %")
