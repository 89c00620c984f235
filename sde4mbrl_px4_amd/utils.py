"""Frame conversion and reference-trajectory helpers on the MPC boundary.

Mirrors `sde4mbrlExamples.rotor_uav.utils.enu2ned` (import site sde_control.py:13, use :400) and the
`state_from_traj` callable returned by load_mpc_from_cfgfile (sde_control.py:164,206,694). The bodies
of both live in the external sde4mbrl package (not in the reference); the CSV column contract and
the linear interpolation rule are the ones visible in the reference's own consumer of the same files,
sde4mbrl_px4/geometric_controller/geometric_controller.cpp:463 (columns) and :251-269 (interpolation).
"""
from __future__ import annotations

import csv
import os

import numpy as np

_SQ = np.float32(np.sqrt(0.5))


def _qmul(a, b, xp):
    """Hamilton product on the last axis (components are [..., 4]); float32 operations in the order written (no fma in numpy)."""
    aw, ax, ay, az = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bw, bx, by, bz = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return xp.stack([aw * bw - ax * bx - ay * by - az * bz,
                     aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw], axis=-1)


def enu2ned(x, xp=np):
    """13-state frame flip world ENU/body FLU <-> world NED/body FRD (an involution); x: [13] or [..., 13] (rows are flipped
    independently, elementwise — a batch gives the same bits as its rows one by one).

    p, v: (x, y, z) -> (y, x, -z); body rates: (wx, wy, wz) -> (wx, -wy, -wz);
    attitude: q' = q_w (x) q (x) q_b with q_w = (0, s, s, 0), q_b = (0, 1, 0, 0), s = sqrt(1/2).
    Signature follows the reference call `enu2ned(curr_state, np)` (sde_control.py:400)."""
    x = xp.asarray(x)
    p, v, q, w = x[..., 0:3], x[..., 3:6], x[..., 6:10], x[..., 10:13]
    qw = xp.asarray([0.0, _SQ, _SQ, 0.0], dtype=x.dtype)
    qb = xp.asarray([0.0, 1.0, 0.0, 0.0], dtype=x.dtype)
    qn = _qmul(_qmul(qw, q, xp), qb, xp)
    out = xp.concatenate([xp.stack([p[..., 1], p[..., 0], -p[..., 2]], axis=-1), xp.stack([v[..., 1], v[..., 0], -v[..., 2]], axis=-1), qn,
                          xp.stack([w[..., 0], -w[..., 1], -w[..., 2]], axis=-1)], axis=-1)
    return out.astype(x.dtype)


ned2enu = enu2ned


def yaw_to_quat(yaw):
    return np.array([np.cos(0.5 * yaw), 0.0, 0.0, np.sin(0.5 * yaw)], dtype=np.float64)


class TrajectoryCSV:
    """Reference trajectory from a CSV with columns t,x,y,z,vx,vy,vz,ax,ay,az,yaw (extra columns ignored).

    `state(t)` returns the 13-state [p, v, q(yaw), 0] by linear interpolation between samples,
    clamped to the first/last row outside the time range (geometric_controller.cpp:224-236,251-269)."""

    REQUIRED = ["t", "x", "y", "z", "vx", "vy", "vz", "ax", "ay", "az", "yaw"]

    def __init__(self, path: str, ned: bool = False):
        path = os.path.expanduser(path)
        with open(path, newline="") as f:
            rd = csv.reader(f)
            header = [h.strip() for h in next(rd)]
            missing = [c for c in self.REQUIRED if c not in header]
            if missing:
                raise ValueError(f"trajectory file {path} lacks columns {missing}")
            idx = [header.index(c) for c in self.REQUIRED]
            rows = []
            for r in rd:
                if not r or all(not c.strip() for c in r):
                    continue
                if len(r) not in (len(header), len(header) - 1):
                    raise ValueError(f"incomplete row in trajectory file {path}")
                rows.append([float(r[i]) if i < len(r) and r[i].strip() else np.nan for i in idx])
        data = np.asarray(rows, dtype=np.float64)
        if data.shape[0] < 1:
            raise ValueError(f"trajectory file {path} has no samples")
        if np.any(np.diff(data[:, 0]) <= 0):
            raise ValueError("trajectory times must be strictly increasing")
        self.t, self.p, self.v, self.a, self.yaw = data[:, 0], data[:, 1:4], data[:, 4:7], data[:, 7:10], data[:, 10]
        self.ned = ned

    def state(self, t):
        t = np.asarray(t, dtype=np.float64)
        tc = np.clip(t, self.t[0], self.t[-1])
        i = np.clip(np.searchsorted(self.t, tc, side="right"), 1, len(self.t) - 1) if len(self.t) > 1 else np.zeros_like(tc, dtype=int)
        if len(self.t) > 1:
            t0, t1 = self.t[i - 1], self.t[i]
            al = ((tc - t0) / (t1 - t0))[..., None]
            p = self.p[i - 1] + al * (self.p[i] - self.p[i - 1])
            v = self.v[i - 1] + al * (self.v[i] - self.v[i - 1])
            yaw = self.yaw[i - 1] + al[..., 0] * (self.yaw[i] - self.yaw[i - 1])
        else:
            p = np.broadcast_to(self.p[0], t.shape + (3,))
            v = np.broadcast_to(self.v[0], t.shape + (3,))
            yaw = np.broadcast_to(self.yaw[0], t.shape)
        out = np.zeros(t.shape + (13,), dtype=np.float64)
        out[..., 0:3], out[..., 3:6] = p, v
        out[..., 6], out[..., 9] = np.cos(0.5 * yaw), np.sin(0.5 * yaw)
        out = out.astype(np.float32)
        if self.ned:
            out = enu2ned(out, np)
        return out

    __call__ = state
