#!/usr/bin/env python3
"""Generates tests/golden/*.npz with the CPU oracle (oracle/sde_mpc_oracle.c).

SELF-GENERATED vectors, NOT reference output: the reference's implementation of this path lives in
the external JAX package sde4mbrl, which is neither vendored in wuwushrek/sde4mbrl_px4 nor
installable here, and the reference holds no tests or fixtures. These files pin the
oracle itself against drift and give the GPU tests fixed inputs/outputs; they do not pin parity with
the original JAX path (parity unpinned).

Usage: python tests/golden/make_golden.py [name ...]   (rewrites the .npz files next to this script; all cases, or the named ones)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import orc  # noqa: E402
from cases import golden_cases  # noqa: E402
from sde4mbrl_px4_amd import workload as W  # noqa: E402


def case(name, cfg, model, seed, curr_t=0.0, pos=False, u_pert=0.1, trace=True):
    H, P, m = cfg.horizon, cfg.num_particles, cfg.num_motors
    O = orc.Oracle(cfg, model)
    x0 = W.random_initial_states(1, seed)[0]
    xref = W.constant_reference(W.HOVER, H) if pos else W.reference_window(curr_t, cfg.time_steps)
    noise = W.make_noise(1, P, H, seed)[0]
    rng = np.random.default_rng(seed + 99)
    u = np.clip(np.asarray(cfg.uref, np.float32) + u_pert * rng.standard_normal((H, m)), 1e-4, 1.0).astype(np.float32)
    cost, traj, xmean = O.rollout(x0, u, xref, noise, True, True)
    gcost, grad = O.grad(x0, u, xref, noise)
    xn, eta = O.step(x0, u[0], noise[0, 0], 0)
    u0 = np.tile(np.asarray(cfg.uref, np.float32), (H, 1))
    uopt, xevol, info, tr = O.solve(x0, xref, noise, u0, cfg.ls_init_stepsize, trace_cap=cfg.max_iter if trace else 0)
    # A8 post-processing (sde_control.py:428-432)
    thrust = uopt.sum(axis=1) / uopt.shape[1]
    wopt = np.stack([thrust, xevol[1:, 10], xevol[1:, 11], xevol[1:, 12]]).T.astype(np.float64)
    out = dict(x0=x0, xref=xref, noise=noise, u=u, cost=np.float32(cost), traj_last=traj[:, -1, :], traj_p0=traj[0], xmean=xmean,
               grad_cost=np.float32(gcost), grad=grad.astype(np.float32), step_xn=xn, step_eta=np.float32(eta),
               u_init=u0, stepsize_in=np.float32(cfg.ls_init_stepsize), uopt=uopt, xevol=xevol, info=info,
               trace=tr if trace else np.zeros((0, 4), np.float32), wopt=wopt)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: cost {cost:.6f} N_it {int(info[2])} N_ls {int(info[7])} init {info[5]:.4f} opt {info[6]:.4f}")


def main():
    for name, (cfg, model, seed, curr_t, pos) in golden_cases().items():
        if len(sys.argv) < 2 or name in sys.argv[1:]:
            case(name, cfg, model, seed, curr_t=curr_t, pos=pos)


if __name__ == "__main__":
    main()
