"""Golden-case definitions shared by tests/golden/make_golden.py and the tests."""
import os

import numpy as np

from sde4mbrl_px4_amd import MPCConfig, load_mpc_config, synthetic_hexa, synthetic_iris, synthetic_multirotor
from sde4mbrl_px4_amd import workload as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CDIR = os.path.join(ROOT, "configs")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def golden_cases():
    """name -> (MPCConfig, model, seed, curr_t, pos_mode)"""
    c1 = load_mpc_config(os.path.join(CDIR, "c1_iris_posctrl_h20_p32.yaml"))
    c2 = load_mpc_config(os.path.join(CDIR, "c2_iris_traj_h50_p128.yaml"))
    c3 = load_mpc_config(os.path.join(CDIR, "c3_hexa_traj_h50_p256.yaml"))
    sh = load_mpc_config(os.path.join(CDIR, "iris_traj_shipped_h20_p1.yaml"))
    return {
        "c1_posctrl_h20_p32": (c1.replace(max_iter=40, max_no_improvement_iter=40), synthetic_iris(), 3, 0.0, True),
        "c2_traj_h12_p40_small": (c2.replace(horizon=12, num_short_dt=12, num_particles=40, max_iter=25, max_no_improvement_iter=25),
                                  synthetic_iris(), 5, 0.7, False),
        "c3_hexa_h10_p33_small": (c3.replace(horizon=10, num_short_dt=6, long_step_dt=0.1, num_particles=33, max_iter=15,
                                             max_no_improvement_iter=15, discount=0.97), synthetic_hexa(), 7, 1.3, False),
        "iris_shipped_h20_p1": (sh.replace(max_iter=30, max_no_improvement_iter=30), synthetic_iris(), 11, 0.2, False),
        # the matrix-pipe contraction modes (SPEC.md §9, §9b): the oracle evaluates them through its model of the instruction (§9a)
        "c2_traj_h12_p40_f32x3": (c2.replace(horizon=12, num_short_dt=12, num_particles=40, max_iter=25, max_no_improvement_iter=25, mlp_dtype="f32x3"),
                                  synthetic_iris(), 5, 0.7, False),
        "c3_hexa_h10_p33_f16": (c3.replace(horizon=10, num_short_dt=6, long_step_dt=0.1, num_particles=33, max_iter=15,
                                           max_no_improvement_iter=15, discount=0.97, mlp_dtype="f16"), synthetic_hexa(), 7, 1.3, False),
    }


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))


def bits_differ(a, b):
    """Number of f32 words whose bit patterns differ. NaNs are compared as a class (a NaN must sit at the
    same position on both sides, but its sign/payload is not specified: x86 SSE produces 0xFFC00000, gfx950
    0x7FC00000 for the same invalid operation — SPEC.md §3); everything else, including the sign of zero and
    infinities, is compared bit for bit."""
    fa = np.ascontiguousarray(a, np.float32)
    fb = np.ascontiguousarray(b, np.float32)
    assert fa.shape == fb.shape, (fa.shape, fb.shape)
    both_nan = np.isnan(fa) & np.isnan(fb)
    return int(((fa.view(np.uint32) != fb.view(np.uint32)) & ~both_nan).sum())


def diverging_single_rotor_case(P=32):
    """A single-rotor vehicle over a 55-step horizon: it tumbles and the explicit-Euler state overflows f32 around t = 53 — cost NaN
    (instance 0) / +inf (instance 1) and a NaN gradient (found by tests/tools/soak.py, case 5495); instance 2 repeats instance 0 at low
    throttle, which stays finite. Exercises SPEC.md §3.7 and the §8 non-finite guard in a mixed batch."""
    B = 2
    m, H, it = 1, 55, 495
    cfg = MPCConfig(horizon=H, num_short_dt=6, short_step_dt=0.05, long_step_dt=0.1, num_particles=P, input_id=[0], input_bound=[[1e-4, 1.0]],
                    uref=[0.55], u_slew_coeff=1.0, max_iter=3, max_no_improvement_iter=6, ls_maxls=5)
    model = synthetic_multirotor(m, seed=it)
    x0 = W.random_initial_states(B, 1000 + it)
    xref = np.stack([W.reference_window(0.2 * b, cfg.time_steps) for b in range(B)])
    noise = W.make_noise(B, P, H, it)
    u = np.clip(0.55 + 0.15 * np.random.default_rng(it).standard_normal((B, H, m)), 1e-4, 1).astype(np.float32)
    x0, xref, noise = (np.concatenate([a, a[:1]]) for a in (x0, xref, noise))
    u = np.concatenate([u, np.full_like(u[:1], 0.02)])
    return cfg, model, x0, xref, noise, u
