"""PCIe-inclusive rate of the host-pointer entry point sdempc_solve_batch (never bench.py's `value`): host arrays in, host arrays out,
against the kernel time of the same call (HIP events). usage: host_api_rate.py [--batch 1024] [--config configs/c2...yaml]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401  (HIP runtime initialised by torch first)
torch.cuda.init()
from sde4mbrl_px4_amd import load_mpc_config, synthetic_iris, workload as W
from sde4mbrl_px4_amd.solver import SdeMpcSolver
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--config", default=os.path.join(ROOT, "configs", "c2_iris_traj_h50_p128.yaml"))
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--mlp-dtype", default="f32x3", choices=["f32", "f16", "f32x3"])
ap.add_argument("--math-mode", default="fast", choices=["exact", "fast"])
a = ap.parse_args()
cfg = load_mpc_config(a.config).replace(mlp_dtype=a.mlp_dtype, math_mode=a.math_mode); model = synthetic_iris(); B = a.batch
x0 = W.random_initial_states(B, 0); xref = np.stack([W.reference_window(0.05 * b, cfg.time_steps) for b in range(B)]); noise = W.make_noise(B, cfg.num_particles, cfg.horizon, 1)
u0 = np.tile(np.asarray(cfg.uref, np.float32), (B, cfg.horizon, 1)); s0 = np.full(B, cfg.ls_init_stepsize, np.float32)
S = SdeMpcSolver(cfg, model, max_batch=B)
S.solve(x0[:8], xref[:8], noise[:8], u0[:8], s0[:8])
for r in range(a.reps):
    t0 = time.perf_counter(); S.solve(x0, xref, noise, u0, s0); wall = time.perf_counter() - t0
    k = S.last_kernel_ms() * 1e-3
    print(f"B={B}: host-pointer call {wall*1e3:.1f} ms = {B/wall:.0f} solves/s; kernel alone {k*1e3:.1f} ms = {B/k:.0f} solves/s; overhead {100*(wall-k)/wall:.1f} %")
# the same solves from 8-byte keys: the noise is drawn on the device (sdempc_solve_batch_keys), only states, references and controls cross PCIe
keys = np.random.default_rng(3).integers(0, 2 ** 32, size=(B, 2), dtype=np.uint32)
S.solve_keys(x0[:8], xref[:8], keys[:8], u0[:8], s0[:8])
for r in range(a.reps):
    t0 = time.perf_counter(); S.solve_keys(x0, xref, keys, u0, s0); wall = time.perf_counter() - t0
    k = S.last_kernel_ms() * 1e-3
    print(f"B={B}: key-driven host-pointer call {wall*1e3:.1f} ms = {B/wall:.0f} solves/s; solve kernel alone {k*1e3:.1f} ms = {B/k:.0f} solves/s; overhead {100*(wall-k)/wall:.1f} %")
S.close()
