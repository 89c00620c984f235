"""Where a single solve in the speculative layout spends its time, by phase of the state machine and by section (work item of the phase /
grid barrier / optimiser code behind it). Diagnostic build only: ONLY_MAIN=1 tools/build_variant.sh sclk "-DSDEMPC_VAR_SPEC_CLK=1", run with
SDEMPC_LIB=build/libsdempc_sclk.so. One workgroup of the group that evaluates the gradient at xk reports (s_memrealtime, 10 ns ticks)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from sde4mbrl_px4_amd import load_mpc_config, synthetic_iris, synthetic_hexa, prng
from sde4mbrl_px4_amd import workload as W
from sde4mbrl_px4_amd.solver import SdeMpcSolver
torch.cuda.init()
name = sys.argv[1] if len(sys.argv) > 1 else "c2_iris_traj_h50_p128"
cfg = load_mpc_config(os.path.join(ROOT, "configs", name + ".yaml"))
if len(sys.argv) > 2: cfg = cfg.replace(math_mode=sys.argv[2])       # (a clock build of the fast-mode translation unit: -DSDEMPC_FAST=1 -DSDEMPC_VAR_SPEC_CLK=1)
H, m = cfg.horizon, cfg.num_motors
S = SdeMpcSolver(cfg, synthetic_iris() if m == 4 else synthetic_hexa(), max_batch=1)
x0 = W.random_initial_states(1, 0)
xref = np.stack([W.reference_window(0.0, cfg.time_steps) if cfg.trajectory_path else W.constant_reference(W.HOVER, H)])
keys = prng.split(prng.PRNGKey(10), 1)
yk, info0 = S.reset()
u0 = yk[None]; s0 = np.full(1, info0["stepsize"], np.float32)
for r in range(3):
    t = time.perf_counter(); uopt, xevol, info = S.solve_keys(x0, xref, keys, u0, s0); ms = (time.perf_counter() - t) * 1e3
print(f"{name}: {S.last_kernel_name()}  host latency {ms:.2f} ms, kernel {S.last_kernel_ms():.2f} ms, N_it {info[0, 2]:.0f} N_ls {info[0, 7]:.0f}")
c = xevol.reshape(-1)[64:64 + 24].astype(np.float64).reshape(6, 4)
names = ["INIT", "GRAD (sequential gradient)", "PAR (trials + candidate gradients)", "SEQ (further trials)", "RED (distributed reduction)", "FINAL"]
tot = c[:, :3].sum() * 1e-5
print(f"  sum of all sections {tot:.2f} ms")
for ph in range(6):
    w, b, o, n = c[ph]
    if n:
        print(f"  {names[ph]:38s} x {n:4.0f}: work {w * 1e-5:7.2f} ms ({w / n * 1e-2:7.1f} us each)  barrier {b * 1e-5:6.2f} ms ({b / n * 1e-2:5.1f} us)  optimiser {o * 1e-5:6.2f} ms ({o / n * 1e-2:5.1f} us)")
hk = xevol.reshape(-1)[64 + 24:64 + 29].astype(np.float64)
nred = c[4, 3]
if nred:
    print("  a polled head (behind a reduction phase), us each: " + ", ".join(f"{n} {v / nred * 1e-2:.2f}" for n, v in zip(
        ["control cost of yk", "wait for the totals + gradient + trial points", "barrier + arrive", "ten reductions", "candidate points + barrier"], hk)))
hw = xevol.reshape(-1)[560:576].astype(np.float64).reshape(4, 4)
print("  gradient groups (y2, xk, y1, y3): work when their gradient was the one used / when not, us: " + "; ".join(f"{r[0]:.1f} (x{r[2]:.0f}) / {r[1]:.1f} (x{r[3]:.0f})" for r in hw))
w = xevol.reshape(-1)[100:100 + 2 * 224].astype(np.float64).reshape(-1, 2)
w = w[w[:, 0] > 0]
xcc, grp = (w[:, 1] // 100).astype(int), (w[:, 1] % 100).astype(int)
print("  work time of a parallel phase by workgroup (us): all %.1f .. %.1f" % (w[:, 0].min(), w[:, 0].max()))
for g in sorted(set(grp)):
    print(f"    group {g}: mean {w[grp == g, 0].mean():6.1f}  min {w[grp == g, 0].min():6.1f}  max {w[grp == g, 0].max():6.1f}   ({(grp == g).sum()} workgroups)")
for x in sorted(set(xcc)):
    sel = (xcc == x) & (grp >= 2) & (grp != 5)
    if sel.any():
        print(f"    XCC {x}, gradient groups only: mean {w[sel, 0].mean():6.1f}  max {w[sel, 0].max():6.1f}   ({sel.sum()} workgroups)")
S.close()
