"""At which step does the partial expected cost of a FAILING line-search trial cross the Armijo threshold? (bench workload, CPU oracle)"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import orc
from sde4mbrl_px4_amd import load_mpc_config, synthetic_iris, prng
from sde4mbrl_px4_amd import workload as W
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import sde_mpc_numpy as R2
cfg = load_mpc_config(os.path.join(ROOT, 'configs', 'c2_iris_traj_h50_p128.yaml'))
model = synthetic_iris(); O = orc.Oracle(cfg, model, fast=True); N = R2.Restatement(cfg, model)
H, P = cfg.horizon, cfg.num_particles
F = np.float32
def stage_partial(traj, xref):
    # per particle per step stage cost (without the res_mult eta^2 term: a lower bound), numpy float64
    x = traj[:, 1:, :].astype(np.float64); r = xref[1:].astype(np.float64)[None]
    l = (np.array(cfg.perr) * (x[..., 0:3] - r[..., 0:3]) ** 2).sum(-1) + (np.array(cfg.verr) * (x[..., 3:6] - r[..., 3:6]) ** 2).sum(-1) + (np.array(cfg.werr) * (x[..., 10:13] - r[..., 10:13]) ** 2).sum(-1)
    qw, qx, qy, qz = [x[..., 6 + i] for i in range(4)]; rw, rx, ry, rz = [r[..., 6 + i] for i in range(4)]
    ex = rw * qx - rx * qw - ry * qz + rz * qy; ey = rw * qy + rx * qz - ry * qw - rz * qx; ez = rw * qz - rx * qy + ry * qx - rz * qw
    l = l + cfg.qerr[0] * ex ** 2 + cfg.qerr[1] * ey ** 2 + cfg.qerr[2] * ez ** 2
    return np.cumsum(l.mean(0) * N.disc[:H].astype(np.float64))      # [H] partial expected state cost after step t
cross = []; ntr = 0; nfail = 0; kinds = []
for inst in range(int(sys.argv[1])):
    x0 = W.random_initial_states(8, 0)[inst]; xref = W.reference_window(0.05 * inst, cfg.time_steps)
    key = prng.split(prng.PRNGKey(10), 8)[inst]; noise = orc.noise_from_key(key, P, H)
    u0 = np.tile(np.asarray(cfg.uref, F), (H, 1))
    def cost_fn(xn):
        return F(O.rollout(x0, xn, xref, noise)[0])
    log = []
    def cost_probe(xn):
        c, traj, _ = O.rollout(x0, xn, xref, noise, True, False)
        log.append((F(c), stage_partial(traj, xref), float(N.control_cost(xn))))
        return F(c)
    # replicate the line-search of R2.solve with probes: wrap cost_fn so that every trial is logged, then post-process with thresholds
    trials = []
    import types
    orig_dot = N.dot256
    state = {}
    def grad_fn(yk):
        c, g = O.grad(x0, yk, xref, noise); state['c_y'] = F(c); return F(c), g.astype(F)
    # monkeypatch: capture (thr) by re-deriving it inside a custom solve loop copy
    C = cfg; m = 4
    lo, hi = np.asarray([b[0] for b in C.input_bound], F), np.asarray([b[1] for b in C.input_bound], F)
    proj = lambda v: np.stack([R2.clamp(v[:, j], lo[j], hi[j]) for j in range(m)], axis=1)
    beta = [F(C.beta_init)] + [F(F(i + 1) / F(i + 4)) for i in range(1, C.max_iter + 2)]
    xk = proj(u0); yk = xk.copy(); c_x = cost_fn(xk); s = F(0.01); kr = noimp = 0; plain = True
    for k in range(C.max_iter):
        c_y, g = grad_fn(yk); gsq = N.dot256(g, g)
        if k > 0: s = F(s * F(C.ls_increase_factor))
        if s > F(C.ls_max_stepsize): s = F(C.ls_max_stepsize)
        for jl in range(C.ls_maxls):
            xn = proj(R2.fma(-s, g, yk)); d1 = xn - yk
            log.clear(); c_n = cost_probe(xn); gd = N.dot256(g, d1); thr = R2.fma(F(C.ls_coef), gd, c_y)
            ntr += 1
            ok = c_n <= thr
            if not ok:
                nfail += 1
                part = log[0][1] / 1.0 + log[0][2]          # partial state cost + full control cost
                idx = np.argmax(part > float(thr)) if np.any(part > float(thr)) else H      # first step whose partial cost exceeds the threshold
                cross.append(idx + 1 if idx < H else H + 1); kinds.append(jl == C.ls_maxls - 1)
            if ok: break
            if jl < C.ls_maxls - 1: s = F(s * F(C.ls_decrease_factor))
        if c_n < c_x:
            if N.dot256(yk - xn, xn - xk) > 0: yk, kr, plain = xn.copy(), 0, True
            else: yk, kr, plain = proj(R2.fma(beta[kr], xn - xk, xn)), kr + 1, False
            xk, c_x, noimp = xn.copy(), c_n, 0
        else:
            yk, kr, plain, noimp = xk.copy(), 0, True, noimp + 1
    print(f"instance {inst}: trials so far {ntr}, failing {nfail}, mean crossing step {np.mean(cross):.1f} of {H}", flush=True)
cross = np.array(cross); kinds = np.array(kinds)
print("failing trials per solve", nfail / int(sys.argv[1]), "of", ntr / int(sys.argv[1]), "trials")
print("crossing step: mean %.1f median %.0f; never crossed before the end (H+1): %.1f %%" % (cross.mean(), np.median(cross), 100 * np.mean(cross > H)))
print("histogram (deciles of H):", np.histogram(cross, bins=[0, 5, 10, 15, 20, 25, 30, 35, 40, 45, 50, 52])[0])
print("last-trial failures (cost IS used):", int(kinds.sum()))
print("rollout steps saved per solve (non-last failing trials): %.0f of %.0f forward-rollout steps" % ((H - np.minimum(cross[~kinds], H)).sum() / int(sys.argv[1]), 377 * H))
