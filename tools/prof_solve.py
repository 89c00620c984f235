"""Run the solve kernel a few times with device-resident inputs (profiling / timing aid)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from sde4mbrl_px4_amd import load_mpc_config, synthetic_iris, synthetic_hexa
from sde4mbrl_px4_amd import workload as W
from sde4mbrl_px4_amd.solver import SdeMpcSolver
ap = argparse.ArgumentParser()
ap.add_argument("--config", default=os.path.join(ROOT, "configs", "c2_iris_traj_h50_p128.yaml"))
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--max-iter", type=int, default=0)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--mode", default="solve", choices=["solve", "rollout", "grad"])
ap.add_argument("--math-mode", default="exact", choices=["exact", "fast"])
ap.add_argument("--mlp-dtype", default="f32", choices=["f32", "f16", "f32x3"])
ap.add_argument("--fixed-work", action="store_true", help="no line search, no stopping rule: max_iter gradient evaluations and rollouts whatever the values")
a = ap.parse_args()
cfg = load_mpc_config(a.config).replace(math_mode=a.math_mode, mlp_dtype=a.mlp_dtype)
if a.max_iter: cfg = cfg.replace(max_iter=a.max_iter, max_no_improvement_iter=a.max_iter)
if a.fixed_work: cfg = cfg.replace(ls_maxls=0, stepsize=1e-4, rtol=0.0, atol=0.0, max_no_improvement_iter=10 ** 6)
H, P, m, B = cfg.horizon, cfg.num_particles, cfg.num_motors, a.batch
dev = torch.device("cuda", 0)
S = SdeMpcSolver(cfg, synthetic_iris() if m == 4 else synthetic_hexa(), max_batch=B)
x0 = torch.from_numpy(W.random_initial_states(B, 0)).to(dev)
xref = torch.from_numpy(np.stack([W.reference_window(0.05 * (b % 160), cfg.time_steps) for b in range(B)])).to(dev)
noise = torch.from_numpy(np.random.default_rng(1).standard_normal(S.lib.sdempc_noise_dev_floats(S._h, B), dtype=np.float32)).to(dev)
yk, info0 = S.reset()
u0 = torch.from_numpy(np.tile(yk[None], (B, 1, 1))).to(dev)
st = torch.full((B,), float(info0["stepsize"]), device=dev)
uopt = torch.empty((B, H, m), device=dev); xevol = torch.empty((B, H + 1, 13), device=dev); info = torch.empty((B, 8), device=dev)
cost = torch.empty((B,), device=dev); grad = torch.empty((B, H, m), device=dev)
stream = torch.cuda.current_stream().cuda_stream
for r in range(a.reps):
    if a.mode == "solve":
        S.solve_dev(B, x0.data_ptr(), xref.data_ptr(), noise.data_ptr(), u0.data_ptr(), st.data_ptr(), uopt.data_ptr(), xevol.data_ptr(), info.data_ptr(), stream)
    elif a.mode == "rollout":
        S.rollout_dev(B, x0.data_ptr(), u0.data_ptr(), xref.data_ptr(), noise.data_ptr(), cost.data_ptr(), None, False, stream)
    else:
        S.grad_dev(B, x0.data_ptr(), u0.data_ptr(), xref.data_ptr(), noise.data_ptr(), cost.data_ptr(), grad.data_ptr(), stream)
    ms = S.last_kernel_ms()
    torch.cuda.synchronize()
    extra = ""
    if a.mode == "solve":
        ih = info.cpu().numpy(); extra = f" N_it {ih[:,2].mean():.1f} N_ls {ih[:,7].mean():.1f} -> {B/ms*1e3:.1f} solves/s"
        wc = S.work_counters(reset=True)
        extra += f"  work: {wc[1]/B:.1f} gradients + {wc[2]/B:.1f} rollouts per solve -> {(2*wc[1]+wc[2])*H*P/ms*1e3/1e9:.2f} G particle-steps/s"
        if wc[0] != B: extra += f"  [work counters: {wc[0]} solves for a batch of {B}]"
    else:
        extra = f" -> {B*H*P/ms*1e3/1e9:.3f} G particle-steps/s"
    print(f"{a.mode} {a.math_mode}/{a.mlp_dtype} B={B} H={H} P={P} rep {r}: {ms:.3f} ms{extra}")
S.close()
