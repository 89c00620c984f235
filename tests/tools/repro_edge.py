"""Re-runs ONE edge case of tests/test_gpu_parity.py (EDGE) in one execution layout on the GPU and prints where the outputs differ from the oracle.
usage: python tests/tools/repro_edge.py <edge case name> auto|coop|tile"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import orc
from cases import bits_differ
from sde4mbrl_px4_amd import MPCConfig, synthetic_iris
from sde4mbrl_px4_amd import workload as W
from sde4mbrl_px4_amd.solver import SdeMpcSolver
import test_gpu_parity as T
name, layout = sys.argv[1], sys.argv[2]
opts = {"auto": {}, "coop": {"spec": 0}, "tile": {"lane": 0, "coop": 0}}[layout]
kw = dict(u_slew_coeff=1.0, max_iter=8, max_no_improvement_iter=8); kw.update(T.EDGE[name])
cfg = MPCConfig(**kw); model = synthetic_iris(); B = 3
x0, xref, noise, u = T._problem(cfg, B, seed=21)
S = SdeMpcSolver(cfg, model, max_batch=B, options=opts); O = orc.Oracle(cfg, model)
u0 = np.tile(np.asarray(cfg.uref, np.float32), (B, cfg.horizon, 1)); s0 = np.full(B, cfg.ls_init_stepsize, np.float32)
print("solving", name, layout, flush=True)
uopt, xevol, info = S.solve(x0, xref, noise, u0, s0)
print("kernel", S.last_kernel_name(), flush=True)
bad = 0
for b in range(B):
    uo, xe, inf, _ = O.solve(x0[b], xref[b], noise[b], u0[b], float(s0[b]))
    d = (bits_differ(uopt[b], uo), bits_differ(xevol[b], xe), bits_differ(info[b], inf))
    bad += sum(d)
    if sum(d):
        print("  instance", b, "uopt/xevol/info words differing:", d, "info gpu", info[b], "oracle", inf)
        w = np.argwhere(xevol[b].view(np.uint32) != np.asarray(xe, np.float32).view(np.uint32))
        print("   first xevol diffs (t, i):", w[:6].tolist(), " max abs", float(np.abs(xevol[b] - xe).max()), " uopt max abs", float(np.abs(uopt[b] - uo).max()))
print(name, layout, "differing words:", bad, flush=True)
