"""Synthetic MPC workloads: initial states, reference trajectories and noise.

The reference's learned weights and trajectory CSVs live in the external sde4mbrl repository
(launch/iris_sitl_traj_mpc.yaml:3,6) and are unavailable, so benchmarks and parity tests use:
hover state (sde_control.py:747) + Gaussian perturbations, an analytic lemniscate at 1 m scale, and
N(0,1) float32 noise from numpy's PCG64.
"""
from __future__ import annotations

import numpy as np

HOVER = np.array([0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0], dtype=np.float32)  # sde_control.py:747


def random_initial_states(B: int, seed0: int = 0) -> np.ndarray:
    """Instance b uses seed seed0+b: sigma_pos 0.5 m, sigma_vel 0.5 m/s, <=15 deg attitude, sigma_w 0.3 rad/s."""
    out = np.zeros((B, 13), dtype=np.float32)
    for b in range(B):
        rng = np.random.default_rng(seed0 + b)
        x = HOVER.copy()
        x[0:3] = 0.5 * rng.standard_normal(3)
        x[3:6] = 0.5 * rng.standard_normal(3)
        axis = rng.standard_normal(3)
        axis /= np.linalg.norm(axis)
        ang = np.deg2rad(15.0) * rng.uniform(0.0, 1.0)
        q = np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * axis])
        x[6:10] = (q / np.linalg.norm(q)).astype(np.float32)
        x[10:13] = 0.3 * rng.standard_normal(3)
        out[b] = x
    return out


def lemniscate_state(t, scale: float = 1.0, period: float = 8.0, z0: float = 1.0) -> np.ndarray:
    """13-state on a figure-eight (stand-in for fast2_lemn.csv): identity attitude, zero body rates."""
    t = np.asarray(t, dtype=np.float64)
    w = 2.0 * np.pi / period
    x = np.zeros(t.shape + (13,), dtype=np.float64)
    x[..., 0] = scale * np.sin(w * t)
    x[..., 1] = scale * np.sin(w * t) * np.cos(w * t)
    x[..., 2] = z0
    x[..., 3] = scale * w * np.cos(w * t)
    x[..., 4] = scale * w * np.cos(2 * w * t)
    x[..., 6] = 1.0
    return x.astype(np.float32)


def reference_window(curr_t: float, time_steps: np.ndarray, state_fn=lemniscate_state) -> np.ndarray:
    """xref[H+1,13] at curr_t + cumulative dt (what the traj solver tracks, sde_control.py:412)."""
    ts = curr_t + np.concatenate([[0.0], np.cumsum(np.asarray(time_steps, dtype=np.float64))])
    return state_fn(ts)


def constant_reference(xdes: np.ndarray, H: int) -> np.ndarray:
    """xref for the position solver: constant target (sde_control.py:400,405,416)."""
    return np.repeat(np.asarray(xdes, dtype=np.float32)[None, :], H + 1, axis=0)


def make_noise(B: int, P: int, H: int, seed: int = 0) -> np.ndarray:
    """xi ~ N(0,1) float32 [B,P,H,6], instance b seeded seed+b (pre-generated noise, parity runs)."""
    out = np.empty((B, P, H, 6), dtype=np.float32)
    for b in range(B):
        out[b] = np.random.default_rng(1_000_003 * (seed + b) + 17).standard_normal((P, H, 6), dtype=np.float32)
    return out
