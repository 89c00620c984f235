"""Where a duo solve launch spends its wave time (diagnostic build only: tools/build_variant.sh clk "-DSDEMPC_VAR_PHASE_CLK=1", run with
SDEMPC_LIB=build/libsdempc_clk.so): forward and adjoint sweeps of the gradient evaluations from the packed counter work[3]
(s_memrealtime, 100 MHz, >> 10), the cost rollouts as the remainder of waves x kernel time."""
import argparse, ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from sde4mbrl_px4_amd import load_mpc_config, synthetic_hexa, synthetic_iris, prng
from sde4mbrl_px4_amd import workload as W
from sde4mbrl_px4_amd.solver import SdeMpcSolver

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="configs/c2_iris_traj_h50_p128.yaml")
ap.add_argument("--batch", type=int, default=3072)
a = ap.parse_args()
torch.cuda.init()
cfg = load_mpc_config(a.config)
m, B, H, P = cfg.num_motors, a.batch, cfg.horizon, cfg.num_particles
S = SdeMpcSolver(cfg, (synthetic_iris() if m == 4 else synthetic_hexa()).to_blob(), max_batch=B, device=0)
x0 = W.random_initial_states(B, 0)
xref = np.stack([W.reference_window(0.05 * (b % 160), cfg.time_steps) for b in range(B)])
keys = prng.split(prng.PRNGKey(10), B)
yk, i0 = S.reset()
u0 = np.tile(yk[None], (B, 1, 1)); s0 = np.full(B, i0["stepsize"], np.float32)
S.solve_keys(x0, xref, keys, u0, s0)
out = (ctypes.c_uint64 * 4)()
S.lib.sdempc_work_counters(S._h, out, 1)
uopt, xevol, info = S.solve_keys(x0, xref, keys, u0, s0)
S.lib.sdempc_work_counters(S._h, out, 0)
w = [int(v) for v in out]
hi = lambda v: (v >> 32) * 10.24
lo = lambda v: (v & 0xFFFFFFFF) * 10.24
t_solve, t_roll, ngrad, t_fwd, t_adj, t_sync = hi(w[0]), lo(w[0]), w[1], hi(w[2]), lo(w[2]), hi(w[3])
nwaves = w[3] & 0xFFFFFFFF
ms = S.last_kernel_ms()
G = (P + 31) // 32; pairs = (G + 1) // 2
nfwd = float((info[:, 7] + 2).sum())                 # cost rollouts: line-search trials + initial + final
print(f"{S.last_kernel_name()}\nkernel {ms:.1f} ms; {B} solves on {nwaves} wave-solves; {ngrad / B:.1f} gradient evaluations and {nfwd / B:.1f} cost rollouts per solve")
for name, v in [("cost rollouts (whole call)", t_roll), ("gradient evaluations: forward sweeps", t_fwd), ("gradient evaluations: adjoint sweeps", t_adj),
                ("everything else (optimiser, reductions, gradient assembly)", t_solve - t_roll - t_fwd - t_adj), ("team barriers (inside all of the above)", t_sync)]:
    print(f"{name:60s} {v / t_solve:6.1%}")
print(f"per wave-step: cost rollout {t_roll / (nfwd * pairs * H):.2f} us (whole call / steps), forward sweep {t_fwd / (ngrad * pairs * H):.2f} us, adjoint sweep {t_adj / (ngrad * pairs * H):.2f} us")
# where and when each instance was solved (diagnostic telemetry words)
hw = info[:, 0].view(np.uint32); xcc = info[:, 1].view(np.uint32) & 15
start = info[:, 3].astype(np.float64); start -= start.min(); dur = info[:, 4].astype(np.float64); nls = info[:, 7]
cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; simd = (hw >> 4) & 3
print(f"per-instance solve duration: mean {dur.mean():.1f} ms, min {dur.min():.1f}, p5 {np.percentile(dur, 5):.1f}, p50 {np.median(dur):.1f}, p95 {np.percentile(dur, 95):.1f}, max {dur.max():.1f}; "
      f"last end {np.max(start + dur):.1f} ms; corr(duration, N_ls) = {np.corrcoef(dur, nls)[0, 1]:.2f}")
first = start < 1.0
print(f"first-round instances ({first.sum()}): mean {dur[first].mean():.1f} ms; later ones: mean {dur[~first].mean():.1f} ms")
for name, key in [("XCC", xcc), ("SE", se), ("SH", sh), ("CU", cu), ("SIMD of the team's first wave", simd)]:
    print(name + ": " + "  ".join(f"{k}: {dur[key == k].mean():.0f} ms x{(key == k).sum()}" for k in np.unique(key)))
print("duration deciles (ms):", " ".join(f"{np.percentile(dur, q):.0f}" for q in range(0, 101, 10)), f"  N_ls min/median/max {nls.min():.0f}/{np.median(nls):.0f}/{nls.max():.0f}")
order = np.argsort(start)
nseg = 8
print("mean duration by start time (eighths of the launch):", " ".join(f"{dur[order[i * B // nseg:(i + 1) * B // nseg]].mean():.0f}" for i in range(nseg)))
wave = hw & 15
print("wave slot of the team's first wave: " + "  ".join(f"{k}: {dur[wave == k].mean():.0f} ms x{(wave == k).sum()}" for k in np.unique(wave)))
key = (xcc.astype(np.int64) << 16) | (se.astype(np.int64) << 8) | cu
cm = np.array([dur[key == k].mean() for k in np.unique(key)])
print(f"per-CU mean duration: min {cm.min():.0f}, p50 {np.median(cm):.0f}, max {cm.max():.0f} ms over {cm.size} CUs")
