"""Per-tick latency through the Python entry points the reference node calls (m_mpc of load_mpc_from_cfgfile; sde_control.py:400-420):
key split, reference window, host-pointer solve, warm start — against the kernel time alone.  usage: python tools/tick_latency.py [--ticks 60]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from sde4mbrl_px4_amd import jax_shim
from sde4mbrl_px4_amd import workload as W
from sde4mbrl_px4_amd.sde_mpc_design import load_mpc_problem

ap = argparse.ArgumentParser(); ap.add_argument("--ticks", type=int, default=60); a = ap.parse_args()
for name in ("iris_traj_shipped_h20_p1", "c1_iris_posctrl_h20_p32", "c2_iris_traj_h50_p128"):
    prob = load_mpc_problem(os.path.join(ROOT, "configs", name + ".yaml"))
    x = W.random_initial_states(1, 4)[0]
    xdes = W.HOVER.copy()
    rng = jax_shim.random.PRNGKey(10)
    st = prob.m_reset(x=x, rng=rng, xdes=xdes)
    lat, ker = [], []
    for k in range(a.ticks):
        t = time.perf_counter()
        uopt, st, rng, xevol = prob.m_mpc(x, rng, st, curr_t=0.05 * k, xdes=xdes)
        uopt.block_until_ready()
        lat.append((time.perf_counter() - t) * 1e3); ker.append(prob.solver().last_kernel_ms())
        x = np.asarray(xevol)[1].astype(np.float32)          # follow the predicted trajectory (warm-started ticks, as in closed loop)
    lat, ker = np.array(lat[5:]), np.array(ker[5:])
    print(f"{name:28s}: m_mpc per tick p50 {np.median(lat):7.2f} ms p95 {np.percentile(lat, 95):7.2f} ms; kernel p50 {np.median(ker):7.2f} ms; "
          f"Python + staging overhead p50 {np.median(lat - ker):5.2f} ms; iterations per tick {float(st.num_steps):.0f}", flush=True)
