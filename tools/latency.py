"""Single-instance solve latency (what one control tick of the reference's mpc_process sees), per config.
usage: python tools/latency.py [--reps 7]; environment defaults of the handle options apply (include/sdempc.h: SDEMPC_PK=0/1 forces the
scalar / packed tanh instantiation, SDEMPC_COOP=0 the one-workgroup-per-instance layout, SDEMPC_COOP_FENCE=1 the fenced grid barrier)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from sde4mbrl_px4_amd import load_mpc_config, synthetic_iris, synthetic_hexa, prng
from sde4mbrl_px4_amd import workload as W
from sde4mbrl_px4_amd.solver import SdeMpcSolver
ap = argparse.ArgumentParser(); ap.add_argument("--reps", type=int, default=7); ap.add_argument("--batches", default="1"); ap.add_argument("--math-mode", default="exact", choices=["exact", "fast"]); a = ap.parse_args()
torch.cuda.init()
for name in ("iris_traj_shipped_h20_p1", "c1_iris_posctrl_h20_p32", "c2_iris_traj_h50_p128", "c3_hexa_traj_h50_p256"):
    cfg = load_mpc_config(os.path.join(ROOT, "configs", name + ".yaml")).replace(math_mode=a.math_mode)
    H, P, m = cfg.horizon, cfg.num_particles, cfg.num_motors
    for B in [int(b) for b in a.batches.split(",")]:
        S = SdeMpcSolver(cfg, synthetic_iris() if m == 4 else synthetic_hexa(), max_batch=B)
        x0 = W.random_initial_states(B, 0)
        xref = np.stack([W.reference_window(0.05 * b, cfg.time_steps) if cfg.trajectory_path else W.constant_reference(W.HOVER, H) for b in range(B)])
        keys = prng.split(prng.PRNGKey(10), B)
        yk, info0 = S.reset()
        u0 = np.tile(yk[None], (B, 1, 1)); s0 = np.full(B, info0["stepsize"], np.float32)
        lat = []
        for r in range(a.reps):
            t = time.perf_counter(); uopt, xevol, info = S.solve_keys(x0, xref, keys, u0, s0); lat.append((time.perf_counter() - t) * 1e3)
        print(f"{name:28s} B={B:4d} H={H} P={P}: host-API latency p50 {np.median(lat[1:]):8.2f} ms (kernel {S.last_kernel_ms():8.2f} ms)  N_it {info[:,2].mean():.0f} N_ls {info[:,7].mean():.0f}", flush=True)
        S.close()
