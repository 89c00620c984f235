"""What happens to single-instance (cooperative-layout) solves when ANOTHER process keeps the GPU busy with throughput launches?
usage: python tools/shared_gpu_probe.py            (parent = latency client; child = throughput load, 3 x 3072 C2 solves)
Expected: no wrong result ever; latencies stretch while the other process's grid occupies the CUs; a barrier timeout (if any) is
answered by the tile-layout fallback (SdeMpcSolver.layout_fallbacks)."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from sde4mbrl_px4_amd import load_mpc_config, synthetic_iris, prng
from sde4mbrl_px4_amd import workload as W
from sde4mbrl_px4_amd.solver import SdeMpcSolver

cfg = load_mpc_config(os.path.join(ROOT, "configs", "c2_iris_traj_h50_p128.yaml"))

def problem(B, seed):
    x0 = W.random_initial_states(B, seed)
    xref = np.stack([W.reference_window(0.05 * (b % 160), cfg.time_steps) for b in range(B)])
    return x0, xref, prng.split(prng.PRNGKey(10 + seed), B)

if len(sys.argv) > 1 and sys.argv[1] == "load":
    B = 3072
    S = SdeMpcSolver(cfg, synthetic_iris(), max_batch=B)
    x0, xref, keys = problem(B, 1)
    yk, i0 = S.reset()
    u0 = np.tile(yk[None], (B, 1, 1)); s0 = np.full(B, i0["stepsize"], np.float32)
    print("load: ready", flush=True)
    sys.stdin.readline()
    for r in range(3):
        t = time.time(); S.solve_keys(x0, xref, keys, u0, s0); print(f"load: batch {r} {time.time() - t:.2f} s", flush=True)
    S.close(); sys.exit(0)

S = SdeMpcSolver(cfg, synthetic_iris(), max_batch=1)
x0, xref, keys = problem(1, 0)
yk, i0 = S.reset()
u0 = yk[None]; s0 = np.array([i0["stepsize"]], np.float32)
ref = S.solve_keys(x0, xref, keys, u0, s0)
quiet = []
for _ in range(10):
    t = time.time(); S.solve_keys(x0, xref, keys, u0, s0); quiet.append((time.time() - t) * 1e3)
child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "load"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
print(child.stdout.readline().strip(), flush=True)
child.stdin.write("go\n"); child.stdin.flush()
lat, bad = [], 0
t_end = time.time() + 6.0
while time.time() < t_end:
    t = time.time(); r = S.solve_keys(x0, xref, keys, u0, s0); lat.append((time.time() - t) * 1e3)
    bad += int(any(not np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(r, ref)))
out = child.communicate(timeout=120)[0]
print(out.strip())
lat = np.array(lat)
print(f"alone: p50 {np.median(quiet):.1f} ms; shared GPU: {len(lat)} solves, p50 {np.median(lat):.1f} ms, p95 {np.percentile(lat, 95):.1f} ms, max {lat.max():.1f} ms; "
      f"results differing from the uncontended one: {bad}; layout fallbacks: {S.layout_fallbacks()}")
S.close()
