"""Full-length solves on the GPU in every arithmetic of the library — mlp_dtype (f32 | f32x3 | f16) x math_mode (exact | fast); each is bit-identical to its
own CPU oracle — against each other and against the FLOAT64 build of the oracle (benchlib/referee.py): how far is each from the real-number solution, over a
FULL solve (C2: 200 APG iterations, ~375 line-search decisions) and per gradient?  usage: python tools/mode_drift.py [--batch 64] [--config ...] [--no-f64]
Prints, per arithmetic, the deviation of the optimal controls from the f32/exact solve's and from the float64 solve's on identical inputs (north star: 1e-4)."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from benchlib import referee as R
from benchlib.verify import Verifier, effective_cores
from sde4mbrl_px4_amd import load_mpc_config, synthetic_iris, synthetic_hexa, prng
from sde4mbrl_px4_amd import workload as W
from sde4mbrl_px4_amd.solver import SdeMpcSolver

ap = argparse.ArgumentParser()
ap.add_argument("--config", default=os.path.join(ROOT, "configs", "c2_iris_traj_h50_p128.yaml"))
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--no-f64", action="store_true", help="skip the float64 referee (CPU: about 10 s per C2 instance and thread)")
a = ap.parse_args()
base = load_mpc_config(a.config)
B, H, m = a.batch, base.horizon, base.num_motors
model = synthetic_iris() if m == 4 else synthetic_hexa()
x0 = W.random_initial_states(B, 0)
xref = np.stack([W.reference_window(0.05 * (b % 160), base.time_steps) for b in range(B)])
keys = prng.split(prng.PRNGKey(10), B)
ug = None
res, grads = {}, {}
for mlp in ("f32", "f32x3", "f16"):
    for mm in ("exact", "fast"):
        cfg = base.replace(mlp_dtype=mlp, math_mode=mm)
        S = SdeMpcSolver(cfg, model, max_batch=B)
        yk, i0 = S.reset()
        u0 = np.tile(yk[None], (B, 1, 1))
        if ug is None:
            ug = np.clip(u0 + 0.1 * np.random.default_rng(7).standard_normal(u0.shape), 1e-4, 1).astype(np.float32)
        res[R.name(mlp, mm)] = S.solve_keys(x0, xref, keys, u0, np.full(B, i0["stepsize"], np.float32))
        grads[R.name(mlp, mm)] = S.grad(x0, ug, xref, S.noise_from_keys(keys))
        S.close()
ue, xe, ie = res["f32/exact"]
print(f"{os.path.basename(a.config)}: B={B}, N_it {ie[:, 2].mean():.0f}, N_ls {ie[:, 7].mean():.0f}; controls in [{ue.min():.3f}, {ue.max():.3f}]")
ref = None
if not a.no_f64:
    V = Verifier(max(1, effective_cores() - 1))
    V.add_referee(base, model.to_blob(), range(B), x0, xref, keys, u0, ug, float(i0["stepsize"]))
    V.start(); V.join()
    assert not V.errors, V.errors[:3]
    ref = V.referee
for name_, (u, x, i) in res.items():
    du = np.abs(u - ue).reshape(B, -1).max(1)
    within = np.mean(np.all(np.abs(u - ue) <= 1e-4 + 1e-4 * np.abs(ue), axis=(1, 2)))
    line = (f"{name_:12s}: vs f32/exact max|du| median {np.median(du):.2e} worst {du.max():.2e}, within 1e-4 {within * 100:5.1f} %, "
            f"N_ls differs in {np.mean(i[:, 7] != ie[:, 7]) * 100:3.0f} %")
    if ref is not None:
        s = [R.solve_error(u[b], ref[b][2]) for b in range(B)]
        g = [R.gradient_error(grads[name_][1][b], ref[b][0], grads[name_][0][b], ref[b][1]) for b in range(B)]
        d64 = np.array([r["max_abs_du"] for r in s])
        rms = np.array([r["rms_rel"] for r in g])
        line += (f" | vs FLOAT64: within 1e-4 {np.mean([r['within'] for r in s]) * 100:5.1f} %, max|du| median {np.median(d64):.2e} worst {d64.max():.2e}; "
                 f"gradient rms {np.sqrt(np.mean(rms * rms)):.2e} worst entry {max(r['max_rel'] for r in g):.2e} (of its largest entry)")
    print(line)
