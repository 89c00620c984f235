"""Oracle (CPU baseline) thread scaling on the current host."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, orc
from sde4mbrl_px4_amd import load_mpc_config, synthetic_iris, workload as W
cfg = load_mpc_config(os.path.join(ROOT, "configs", "c2_iris_traj_h50_p128.yaml")).replace(max_iter=10, max_no_improvement_iter=10)
m = synthetic_iris()
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try: print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e: print("no cgroup cpu.max", e)
u0 = np.tile(np.float32(0.71), (50, 4))
for n in (1, 4, 16, 32, 64, 128):
    O = [orc.Oracle(cfg, m) for _ in range(n)]
    x0 = W.random_initial_states(n, 0); xref = W.reference_window(0, cfg.time_steps); noise = W.make_noise(n, 128, 50, 0)
    def w(i): O[i].solve(x0[i], xref, noise[i], u0, 0.01)
    t = time.time(); th = [threading.Thread(target=w, args=(i,)) for i in range(n)]; [x.start() for x in th]; [x.join() for x in th]; dt = time.time() - t
    print(f"{n:4d} threads: {dt:.2f} s -> {n/dt:.2f} solves(10 it)/s")
