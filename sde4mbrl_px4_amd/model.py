"""Rotor neural-SDE model parameters and their flat binary blob (SPEC.md §2).

The reference loads learned weights from a pickle named by `learned_model_params`
(launch/iris_sitl_traj_mpc.yaml:3) through the external package sde4mbrl; that format is not in the
reference. This module defines this build's own model container: physics prior
(mass, inertia, rotor geometry, thrust/moment polynomials) + residual drift MLP (6+m -> 32 -> 32 -> 6)
+ density MLP (6 -> 32 -> 1) scaling the diffusion on v and omega.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from ._abi import BLOB_FLOATS, BLOB_HEADER_INTS, BLOB_MAGIC, HID, MAX_MOTORS, NNOISE


@dataclass
class RotorSDEModel:
    num_motors: int
    mass: float
    grav: float
    inertia: np.ndarray            # [3] diagonal
    thrust_poly: np.ndarray        # [ct2, ct1, ct0]  thrust_j = ct2 u^2 + ct1 u + ct0   [N]
    moment_poly: np.ndarray        # [cm2, cm1]       yaw moment_j = dir_j (cm2 u^2 + cm1 u) [Nm]
    rotor_x: np.ndarray            # [m] body-frame x of each rotor (FLU body, z up)
    rotor_y: np.ndarray            # [m]
    rotor_dir: np.ndarray          # [m] +1 / -1 yaw-torque sign
    res_force_scale: np.ndarray    # [3]
    res_torque_scale: np.ndarray   # [3]
    sigma: np.ndarray              # [6] diffusion amplitude on (v, omega)
    W1z: np.ndarray                # [64][6]  rows 0..31 drift net, 32..63 density net; inputs (v_body, omega)
    b1: np.ndarray                 # [64]
    W1u: np.ndarray                # [32][m]  drift net input weights for the controls
    W2: np.ndarray                 # [32][32]
    b2: np.ndarray                 # [32]
    W3: np.ndarray                 # [6][32]  -> (F_res xyz, tau_res xyz)
    b3: np.ndarray                 # [6]
    w3n: np.ndarray                # [32]     density net output weights
    b3n: float = 0.0

    _ARRAY_FIELDS = ("inertia", "thrust_poly", "moment_poly", "rotor_x", "rotor_y", "rotor_dir", "res_force_scale", "res_torque_scale",
                     "sigma", "W1z", "b1", "W1u", "W2", "b2", "W3", "b3", "w3n")

    def save_npz(self, path: str) -> None:
        """Portable container for user-supplied weights (the reference's *.pkl format lives in the external repo)."""
        np.savez(path, num_motors=self.num_motors, mass=self.mass, grav=self.grav, b3n=self.b3n,
                 **{k: np.asarray(getattr(self, k), np.float32) for k in self._ARRAY_FIELDS})

    @classmethod
    def load_npz(cls, path: str) -> "RotorSDEModel":
        d = np.load(path)
        m = int(d["num_motors"])
        shapes = {"inertia": (3,), "thrust_poly": (3,), "moment_poly": (2,), "rotor_x": (m,), "rotor_y": (m,), "rotor_dir": (m,),
                  "res_force_scale": (3,), "res_torque_scale": (3,), "sigma": (6,), "W1z": (64, 6), "b1": (64,), "W1u": (32, m),
                  "W2": (32, 32), "b2": (32,), "W3": (6, 32), "b3": (6,), "w3n": (32,)}
        kw = {}
        for k, shp in shapes.items():
            a = np.asarray(d[k], np.float32)
            if a.shape != shp:
                raise ValueError(f"{path}: array {k} has shape {a.shape}, expected {shp}")
            kw[k] = a
        return cls(num_motors=m, mass=float(d["mass"]), grav=float(d["grav"]), b3n=float(d["b3n"]), **kw)

    def to_blob(self) -> bytes:
        m = self.num_motors
        assert 1 <= m <= MAX_MOTORS
        hd = np.zeros(BLOB_HEADER_INTS, dtype=np.int32)
        hd[:7] = [BLOB_MAGIC, 1, m, HID, NNOISE, NNOISE, 0]
        f = np.zeros(BLOB_FLOATS, dtype=np.float32)
        J = np.asarray(self.inertia, dtype=np.float32)
        f[0] = np.float32(1.0) / np.float32(self.mass)
        f[1] = self.grav
        f[2:5] = J
        f[5:8] = np.float32(1.0) / J
        f[8:11] = self.thrust_poly
        f[11:13] = self.moment_poly
        o = 16
        f[o:o + m] = self.rotor_x
        f[o + 8:o + 8 + m] = self.rotor_y
        f[o + 16:o + 16 + m] = self.rotor_dir
        o += 24
        f[o:o + 3] = self.res_force_scale
        f[o + 3:o + 6] = self.res_torque_scale
        o += 8
        f[o:o + 6] = self.sigma
        o += 8
        f[o:o + 384] = np.asarray(self.W1z, dtype=np.float32).reshape(-1)
        o += 384
        f[o:o + 64] = self.b1
        o += 64
        w1u = np.zeros((HID, MAX_MOTORS), dtype=np.float32)
        w1u[:, :m] = self.W1u
        f[o:o + 256] = w1u.reshape(-1)
        o += 256
        f[o:o + 1024] = np.asarray(self.W2, dtype=np.float32).reshape(-1)
        o += 1024
        f[o:o + 32] = self.b2
        o += 32
        f[o:o + 192] = np.asarray(self.W3, dtype=np.float32).reshape(-1)
        o += 256
        f[o:o + 6] = self.b3
        o += 8
        f[o:o + 32] = self.w3n
        o += 32
        f[o] = self.b3n
        o += 8
        assert o == BLOB_FLOATS
        return hd.tobytes() + f.tobytes()


def _mlp_weights(rng, m):
    def lin(n_out, n_in, gain=1.0):
        return (rng.standard_normal((n_out, n_in)) * gain / np.sqrt(n_in)).astype(np.float32)
    W1 = lin(64, 6 + m)
    W1z = np.ascontiguousarray(W1[:, :6])
    W1u = np.ascontiguousarray(W1[:32, 6:])
    b1 = (0.1 * rng.standard_normal(64)).astype(np.float32)
    W2 = lin(32, 32)
    b2 = (0.1 * rng.standard_normal(32)).astype(np.float32)
    W3 = lin(6, 32)
    b3 = np.zeros(6, dtype=np.float32)
    w3n = lin(1, 32)[0]
    return W1z, b1, W1u, W2, b2, W3, b3, w3n


def synthetic_iris(seed: int = 10) -> RotorSDEModel:
    """Synthetic Iris quadrotor (the real iris_sitl_sde.pkl lives in the external repo).

    Seed 10 mirrors launch/iris_sdectrl.launch:8. Hover thrust is reached at u = 0.71
    (cost_params.uref, launch/iris_sitl_traj_mpc.yaml:33)."""
    rng = np.random.default_rng(seed)
    W1z, b1, W1u, W2, b2, W3, b3, w3n = _mlp_weights(rng, 4)
    return RotorSDEModel(
        num_motors=4, mass=1.5, grav=9.81, inertia=np.array([0.029, 0.029, 0.055], np.float32),
        thrust_poly=np.array([5.5, 1.28, 0.0], np.float32), moment_poly=np.array([0.088, 0.02048], np.float32),
        rotor_x=np.array([0.13, -0.13, 0.13, -0.13], np.float32),
        rotor_y=np.array([-0.22, 0.20, 0.22, -0.20], np.float32),
        rotor_dir=np.array([1.0, 1.0, -1.0, -1.0], np.float32),
        res_force_scale=np.array([0.3, 0.3, 0.5], np.float32),
        res_torque_scale=np.array([0.004, 0.004, 0.002], np.float32),
        sigma=np.array([0.15, 0.15, 0.15, 0.3, 0.3, 0.3], np.float32),
        W1z=W1z, b1=b1, W1u=W1u, W2=W2, b2=b2, W3=W3, b3=b3, w3n=w3n, b3n=0.0)


def synthetic_hexa(seed: int = 10) -> RotorSDEModel:
    """Synthetic hexarotor: hover at u = 0.42 (launch/hexa_sitl_traj_mpc.yaml:14)."""
    rng = np.random.default_rng(seed + 1000)
    W1z, b1, W1u, W2, b2, W3, b3, w3n = _mlp_weights(rng, 6)
    ang = np.deg2rad(np.array([90.0, 270.0, 330.0, 150.0, 30.0, 210.0]))  # PX4 hexa-x ordering, angle from +x
    arm = 0.275
    return RotorSDEModel(
        num_motors=6, mass=2.0, grav=9.81, inertia=np.array([0.045, 0.045, 0.08], np.float32),
        thrust_poly=np.array([12.0, 2.745, 0.0], np.float32), moment_poly=np.array([0.192, 0.04392], np.float32),
        rotor_x=(arm * np.cos(ang)).astype(np.float32), rotor_y=(arm * np.sin(ang)).astype(np.float32),
        rotor_dir=np.array([-1.0, 1.0, -1.0, 1.0, 1.0, -1.0], np.float32),
        res_force_scale=np.array([0.3, 0.3, 0.5], np.float32),
        res_torque_scale=np.array([0.006, 0.006, 0.003], np.float32),
        sigma=np.array([0.15, 0.15, 0.15, 0.3, 0.3, 0.3], np.float32),
        W1z=W1z, b1=b1, W1u=W1u, W2=W2, b2=b2, W3=W3, b3=b3, w3n=w3n, b3n=0.0)


def synthetic_multirotor(num_motors: int, seed: int = 10, hover_u: float = 0.6) -> RotorSDEModel:
    """Generic synthetic vehicle with `num_motors` rotors evenly spaced on a circle (1..8 motors): exercises the
    motor-count-generic kernel instantiation. Hover thrust is reached at u = hover_u."""
    m = int(num_motors)
    assert 1 <= m <= MAX_MOTORS
    rng = np.random.default_rng(seed + 77 * m)
    W1z, b1, W1u, W2, b2, W3, b3, w3n = _mlp_weights(rng, m)
    ang = 2.0 * np.pi * (np.arange(m) + 0.5) / m
    mass, grav, arm = 1.2 + 0.2 * m, 9.81, 0.25
    t_hover = mass * grav / m
    ct2 = 0.8 * t_hover / hover_u ** 2
    ct1 = 0.2 * t_hover / hover_u
    return RotorSDEModel(
        num_motors=m, mass=mass, grav=grav, inertia=np.array([0.03, 0.03, 0.05], np.float32) * (1 + 0.1 * m),
        thrust_poly=np.array([ct2, ct1, 0.0], np.float32), moment_poly=np.array([0.016 * ct2, 0.016 * ct1], np.float32),
        rotor_x=(arm * np.cos(ang)).astype(np.float32), rotor_y=(arm * np.sin(ang)).astype(np.float32),
        rotor_dir=np.where(np.arange(m) % 2 == 0, 1.0, -1.0).astype(np.float32),
        res_force_scale=np.array([0.3, 0.3, 0.5], np.float32), res_torque_scale=np.array([0.004, 0.004, 0.002], np.float32),
        sigma=np.array([0.15, 0.15, 0.15, 0.3, 0.3, 0.3], np.float32),
        W1z=W1z, b1=b1, W1u=W1u, W2=W2, b2=b2, W3=W3, b3=b3, w3n=w3n, b3n=0.0)
