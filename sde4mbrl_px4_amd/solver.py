"""Thin Python binding of the C ABI (include/sdempc.h): batched rollout / gradient / solve on MI355X.

Host code stays in Python (as in the reference, whose MPC node is Python calling compiled JAX
callables, sde_control.py:681-721); all hot-path arithmetic happens in csrc/libsdempc.so.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _abi
from ._abi import INFO_FIELDS, SdempcInfo


class SdempcError(RuntimeError):
    pass


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {tuple(a.shape)}")
    return a


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class SdeMpcSolver:
    """One solver handle = one (MPC config, model). Single-threaded, like the reference's solver
    objects (one blocking call at a time, sde_control.py:420)."""

    def __init__(self, mpc_cfg, model, max_batch: int = 1, device: int = 0, options=None):
        self.lib = _abi.load_library()
        self.cfg_py = mpc_cfg
        self.cfg, self._keep = mpc_cfg.to_cfg()
        blob = model.to_blob() if hasattr(model, "to_blob") else bytes(model)
        self._blob = C.create_string_buffer(blob, len(blob))
        self.H, self.P, self.m = mpc_cfg.horizon, mpc_cfg.num_particles, mpc_cfg.num_motors
        self.max_batch = int(max_batch)
        h = C.c_void_p()
        rc = self.lib.sdempc_create(C.byref(self.cfg), self._blob, len(blob), self.max_batch, C.byref(h))
        if rc != 0:
            raise SdempcError(f"sdempc_create failed ({rc}): {self.lib.sdempc_last_error(None).decode()}")
        self._h = h
        if device:
            self._check(self.lib.sdempc_set_device(self._h, int(device)))
        for k, v in (options or {}).items():
            self.set_option(k, v)

    # ---- execution options (include/sdempc.h SDEMPC_OPT_*): layout choices, never a bit of the results ----
    def set_option(self, name: str, value: int):
        self._check(self.lib.sdempc_set_option(self._h, _abi.OPTIONS[name], int(value)))

    def get_option(self, name: str) -> int:
        v = C.c_int32()
        self._check(self.lib.sdempc_get_option(self._h, _abi.OPTIONS[name], C.byref(v)))
        return int(v.value)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.sdempc_destroy(self._h)
            self._h = None

    def detach(self):
        """Forget the handle WITHOUT destroying it: for a handle inherited through fork(), whose HIP objects belong to the parent's
        context (sdempc_destroy would call hipFree / hipStreamDestroy on it). The few host bytes are leaked on purpose."""
        self._h = None

    def device_ready(self) -> bool:
        """Has this handle initialised the GPU (device buffers, stream)? Host-only query."""
        return bool(getattr(self, "_h", None)) and bool(self.lib.sdempc_device_ready(self._h))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise SdempcError(f"sdempc error {rc}: {self.lib.sdempc_last_error(self._h).decode()}")

    # ---- m_reset ------------------------------------------------------------------------------
    def reset(self, x=None, xdes=None):
        x = _f32(np.zeros(13) if x is None else x, (13,))
        xdes = _f32(x if xdes is None else xdes, (13,))
        yk = np.zeros((self.H, self.m), np.float32)
        info = SdempcInfo()
        self._check(self.lib.sdempc_reset(self._h, _fp(x), _fp(xdes), _fp(yk), C.byref(info)))
        return yk, {k: getattr(info, k) for k in INFO_FIELDS}

    # ---- batched host-pointer entry points ------------------------------------------------------
    def rollout(self, x0, u, xref, noise, want_traj=False, want_mean=False):
        x0 = _f32(x0)
        B = x0.shape[0]
        x0, u = _f32(x0, (B, 13)), _f32(u, (B, self.H, self.m))
        xref, noise = _f32(xref, (B, self.H + 1, 13)), _f32(noise, (B, self.P, self.H, 6))
        cost = np.zeros(B, np.float32)
        traj = np.zeros((B, self.P, self.H + 1, 13), np.float32) if want_traj else None
        mean = np.zeros((B, self.H + 1, 13), np.float32) if want_mean else None
        self._check(self.lib.sdempc_rollout_batch(self._h, B, _fp(x0), _fp(u), _fp(xref), _fp(noise), _fp(cost),
                                                  _fp(traj) if want_traj else None, _fp(mean) if want_mean else None))
        return cost, traj, mean

    def grad(self, x0, u, xref, noise):
        x0 = _f32(x0)
        B = x0.shape[0]
        x0, u = _f32(x0, (B, 13)), _f32(u, (B, self.H, self.m))
        xref, noise = _f32(xref, (B, self.H + 1, 13)), _f32(noise, (B, self.P, self.H, 6))
        cost = np.zeros(B, np.float32)
        g = np.zeros((B, self.H, self.m), np.float32)
        self._check(self.lib.sdempc_grad_batch(self._h, B, _fp(x0), _fp(u), _fp(xref), _fp(noise), _fp(cost), _fp(g)))
        return cost, g

    def solve(self, x0, xref, noise, u_init, stepsize_in):
        x0 = _f32(x0)
        B = x0.shape[0]
        x0, u_init = _f32(x0, (B, 13)), _f32(u_init, (B, self.H, self.m))
        xref, noise = _f32(xref, (B, self.H + 1, 13)), _f32(noise, (B, self.P, self.H, 6))
        stepsize_in = _f32(stepsize_in, (B,))
        uopt = np.zeros((B, self.H, self.m), np.float32)
        xevol = np.zeros((B, self.H + 1, 13), np.float32)
        info = (SdempcInfo * B)()
        self._check(self.lib.sdempc_solve_batch(self._h, B, _fp(x0), _fp(xref), _fp(noise), _fp(u_init), _fp(stepsize_in),
                                                _fp(uopt), _fp(xevol), info))
        info_np = np.frombuffer(info, dtype=np.float32).reshape(B, 8).copy()
        return uopt, xevol, info_np

    # ---- key-derived noise (SPEC.md §7): keys uint32[B][2], JAX threefry conventions ------------------
    @staticmethod
    def _keys(keys, B=None):
        k = np.ascontiguousarray(keys, dtype=np.uint32)
        if k.ndim != 2 or k.shape[1] != 2 or (B is not None and k.shape[0] != B):
            raise ValueError(f"keys must be uint32[B][2], got {k.shape}")
        return k

    def solve_keys(self, x0, xref, keys, u_init, stepsize_in):
        """m_mpc's mapping: the noise of instance b is normal(keys[b], (P, H, 6)), drawn on the device."""
        x0 = _f32(x0)
        B = x0.shape[0]
        x0, u_init = _f32(x0, (B, 13)), _f32(u_init, (B, self.H, self.m))
        xref, keys, stepsize_in = _f32(xref, (B, self.H + 1, 13)), self._keys(keys, B), _f32(stepsize_in, (B,))
        uopt = np.zeros((B, self.H, self.m), np.float32)
        xevol = np.zeros((B, self.H + 1, 13), np.float32)
        info = (SdempcInfo * B)()
        self._check(self.lib.sdempc_solve_batch_keys(self._h, B, _fp(x0), _fp(xref), keys.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                     _fp(u_init), _fp(stepsize_in), _fp(uopt), _fp(xevol), info))
        return uopt, xevol, np.frombuffer(info, dtype=np.float32).reshape(B, 8).copy()

    def noise_from_keys(self, keys):
        """The canonical noise tensors f32[B][P][H][6] the device draws from keys (inspection / parity tests)."""
        keys = self._keys(keys)
        B = keys.shape[0]
        out = np.zeros((B, self.P, self.H, 6), np.float32)
        self._check(self.lib.sdempc_noise_from_keys(self._h, B, keys.ctypes.data_as(C.POINTER(C.c_uint32)), _fp(out)))
        return out

    def noise_from_keys_dev(self, keys, noise_out, stream=0):
        """keys: host uint32[B][2]; noise_out: device pointer to sdempc_noise_dev_floats(B) floats (device layout)."""
        keys = self._keys(keys)
        self._check(self.lib.sdempc_noise_from_keys_dev(self._h, keys.shape[0], keys.ctypes.data_as(C.POINTER(C.c_uint32)), noise_out,
                                                        C.c_void_p(stream)))

    # ---- device-resident entry points (torch tensors or raw device pointers) ---------------------
    def noise_to_device_layout(self, noise):
        noise = _f32(noise)
        B = noise.shape[0]
        out = np.zeros(self.lib.sdempc_noise_dev_floats(self._h, B), np.float32)
        self._check(self.lib.sdempc_noise_to_device_layout(self._h, B, _fp(noise), _fp(out)))
        return out.reshape(B, (self.P + 31) // 32, self.H, 6, 32)

    def noise_to_device_layout_dev(self, B, noise_canonical, noise_out, stream=0):
        """Device pointers: canonical f32[B][P][H][6] -> f32[B][G][H][6][32] (LDS-tiled transpose on the GPU)."""
        self._check(self.lib.sdempc_noise_to_device_layout_dev(self._h, B, noise_canonical, noise_out, C.c_void_p(stream)))

    def traj_to_canonical_dev(self, B, traj_out, stream=0):
        """Particle x horizon tensor of the last rollout_dev(store_traj=True) / grad_dev -> canonical f32[B][P][H+1][13]."""
        self._check(self.lib.sdempc_traj_to_canonical_dev(self._h, B, traj_out, C.c_void_p(stream)))

    def solve_dev(self, B, x0, xref, noise_dev, u_init, stepsize, uopt, xevol, info, stream=0):
        """All arguments are device pointers (ints). info: f32[B][8]."""
        self._check(self.lib.sdempc_solve_batch_dev(self._h, B, x0, xref, noise_dev, u_init, stepsize, uopt, xevol, info,
                                                    C.c_void_p(stream)))

    def rollout_dev(self, B, x0, u, xref, noise_dev, cost, xmean=None, store_traj=False, stream=0):
        self._check(self.lib.sdempc_rollout_batch_dev(self._h, B, x0, u, xref, noise_dev, cost, xmean, int(store_traj),
                                                      C.c_void_p(stream)))

    def grad_dev(self, B, x0, u, xref, noise_dev, cost, grad, stream=0):
        self._check(self.lib.sdempc_grad_batch_dev(self._h, B, x0, u, xref, noise_dev, cost, grad, C.c_void_p(stream)))

    def solve_status(self):
        """After synchronising the stream of the last solve_dev call: raises SdempcError if a grid barrier of a cooperative layout
        gave up (results invalid; the handle then stays on the one-workgroup-per-instance layouts, so the call can be repeated)."""
        self._check(self.lib.sdempc_solve_status(self._h))

    def layout_fallbacks(self) -> int:
        return int(self.lib.sdempc_layout_fallbacks(self._h))

    def work_counters(self, reset: bool = False):
        """(solves, gradient evaluations, forward-only rollouts) done by this handle's solve launches so far (sdempc_work_counters)."""
        out = (C.c_uint64 * 4)()
        self._check(self.lib.sdempc_work_counters(self._h, out, int(reset)))
        return int(out[0]), int(out[1]), int(out[2])

    def last_kernel_name(self) -> str:
        """Kernel instantiation the last *_dev launch started, as rocprofv3 prints it (sdempc_last_kernel_name)."""
        buf = C.create_string_buffer(512)
        self._check(self.lib.sdempc_last_kernel_name(self._h, buf, 512))
        return buf.value.decode()

    def last_kernel_ms(self) -> float:
        return float(self.lib.sdempc_last_kernel_ms(self._h))
