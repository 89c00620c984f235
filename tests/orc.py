"""ctypes wrapper around oracle/liborc.so — the CPU restatement used as the parity checker.

Test infrastructure only (see oracle/sde_mpc_oracle.c header). Builds the library with
`make -C oracle` when it is missing or older than its source.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_DIR = os.path.join(ROOT, "oracle")
_LIB = None
_LIB_FAST = None


def _make(target=None):
    """`make -C oracle [target]`: always invoked — the Makefile knows every dependency (sources, include/sdempc.h whose sdempc_cfg
    layout the oracle shares, itself) and is a no-op when the library is current."""
    cmd = ["make", "-C", ORC_DIR, "-s"] + ([target] if target else [])
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"{' '.join(cmd)} failed:\n{r.stdout}")


def build():
    _make("liborc.so")
    return os.path.join(ORC_DIR, "liborc.so")


def _native_lib(name):
    """A -march=native build of oracle/sde_mpc_oracle.c (oracle/Makefile), for bench.py's CPU timing leg only. Rebuilt when missing,
    stale, or built for another CPU model (the GPU box's host is not the build container's)."""
    so = os.path.join(ORC_DIR, name + ".so")
    tag = os.path.join(ORC_DIR, name + ".cpu")
    try:
        here = next(l for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except Exception:
        here = ""
    if os.path.exists(so) and ((not os.path.exists(tag)) or open(tag).read() != here):
        os.remove(so)                    # built for another CPU model: force the rebuild; staleness otherwise is make's business
    _make(name + ".so")
    return C.CDLL(so)


def lib_fast():
    """liborc_fast.so: the checker's scalar code built -O3 -march=native (same bits, a little faster)."""
    global _LIB_FAST
    if _LIB_FAST is None:
        _LIB_FAST = _native_lib("liborc_fast")
    return _LIB_FAST


_LIB_VEC = None


def lib_vec():
    """liborc_vec.so: the particle-vectorised timing build (-DORC_VEC, 16 particles per call, orcv_* symbols; tolerance parity only)."""
    global _LIB_VEC
    if _LIB_VEC is None:
        _LIB_VEC = _native_lib("liborc_vec")
    return _LIB_VEC


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        for pre in ("orc_", "orcd_"):
            ft = C.c_float if pre == "orc_" else C.c_double
            for fn in ("rcp", "rsqrt", "tanh", "sigmoid"):
                f = getattr(_LIB, pre + fn)
                f.argtypes, f.restype = [ft], ft
        _LIB.orc_log.argtypes, _LIB.orc_log.restype = [C.c_float], C.c_float
        _LIB.orc_erfinv.argtypes, _LIB.orc_erfinv.restype = [C.c_float], C.c_float
        _LIB.orc_bits_to_normal.argtypes, _LIB.orc_bits_to_normal.restype = [C.c_uint32], C.c_float
    return _LIB


def set_threads(n):
    """Threads ONE solve / gradient / rollout of the checker spreads its particle loops over (oracle/sde_mpc_oracle.c: set_threads; same bits at any
    count). Process-wide, default 1: callers that run one solve per Python thread (bench.py's verifier) keep 1; golden generators and lone solves take the cores."""
    L = lib()
    L.orc_set_threads(int(n)); L.orcd_set_threads(int(n))


def _key(key):
    k = np.asarray(key, dtype=np.uint32).reshape(2)
    return (C.c_uint32 * 2)(int(k[0]), int(k[1]))


def threefry2x32(key, x0, x1):
    out = (C.c_uint32 * 2)()
    lib().orc_threefry2x32(_key(key), C.c_uint32(int(x0)), C.c_uint32(int(x1)), out)
    return int(out[0]), int(out[1])


def split(key, num=2):
    out = np.zeros((num, 2), np.uint32)
    lib().orc_split(_key(key), C.c_int(num), out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out


def random_bits(key, n):
    out = np.zeros(n, np.uint32)
    lib().orc_random_bits(_key(key), C.c_size_t(n), out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out


def normal(key, n):
    out = np.zeros(n, np.float32)
    lib().orc_normal(_key(key), C.c_size_t(n), _fp(out))
    return out


def noise_from_key(key, P, H):
    """SPEC.md §7.3: canonical f32[P][H][6] noise tensor of one solve."""
    return normal(key, P * H * 6).reshape(P, H, 6)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


TRANSC_DIR = os.path.join(ROOT, "tests", "golden", "transc")
import tempfile
TRANSC_BIN = os.path.join(tempfile.gettempdir(), f"sdempc_oracle_{os.getuid()}", "transc_tables.bin")      # 130 MiB, derived: kept out of the repository tree
TRANSC_BLOCKS = ["rcp_s0_e127", "rsq_s0_e127", "rsq_s0_e128"] + [f"exp_s{s}_e{e}" for e in range(97, 128) for s in (0, 1)]


import threading
_TRANSC_LOCK = threading.Lock()


def transc_tables(L=None):
    with _TRANSC_LOCK:          # (bench.py's verifier threads all arrive here at once)
        return _transc_tables(L)


def _transc_tables(L=None):
    """Maps the recorded answers of v_exp_f32 / v_rcp_f32 / v_rsq_f32 into the oracle (oracle/transc_model.c, SPEC.md §10a). The committed form is
    tests/golden/transc/<block>.i8.xz (2^23 differences of -1 / 0 / +1 ulp per block, 1.9 MB in all); the model reads them two bits per answer
    from one file (65 blocks x 2 MiB), written here on first use and whenever a committed block is newer."""
    import lzma
    L = L or lib()
    if L.orc_transc_ready():
        return
    src = [os.path.join(TRANSC_DIR, b + ".i8.xz") for b in TRANSC_BLOCKS]
    stale = (not os.path.exists(TRANSC_BIN)) or os.path.getsize(TRANSC_BIN) != len(src) << 21 or any(os.path.getmtime(f) > os.path.getmtime(TRANSC_BIN) for f in src)
    if stale:
        os.makedirs(os.path.dirname(TRANSC_BIN), exist_ok=True)
        import uuid
        tmp = TRANSC_BIN + f".{os.getpid()}.{uuid.uuid4().hex}.tmp"       # (other PROCESSES may be writing their own copy: the rename is atomic, the contents identical)
        with open(tmp, "wb") as out:
            for f in src:
                d = np.frombuffer(lzma.decompress(open(f, "rb").read()), dtype=np.int8)
                assert d.size == 1 << 23 and d.min() >= -1 and d.max() <= 1, f
                q = (d + 1).astype(np.uint8).reshape(-1, 4)
                out.write((q[:, 0] | (q[:, 1] << 2) | (q[:, 2] << 4) | (q[:, 3] << 6)).astype(np.uint8).tobytes())
        os.replace(tmp, TRANSC_BIN)
    L.orc_transc_open.argtypes = [C.c_char_p]
    rc = L.orc_transc_open(TRANSC_BIN.encode())
    if rc != 0:
        raise RuntimeError(f"orc_transc_open({TRANSC_BIN}) failed: {rc}")


def hw_eval(func, x):
    """v_rcp_f32 (0) / v_rsq_f32 (1) / v_exp_f32 (2) of a float32 array through the model"""
    L = lib()
    transc_tables(L)
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    L.orc_hw_eval.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
    L.orc_hw_eval.restype = None
    L.orc_hw_eval(func, x.ctypes.data, y.ctypes.data, x.size)
    return y


class Oracle:
    """Oracle bound to one (config, model blob). double=True selects the float64 build."""

    def __init__(self, mpc_cfg, model, double=False, fast=False, vec=False):
        self.cfg_py = mpc_cfg
        self.cfg, self._keep = mpc_cfg.to_cfg()
        self.blob = model.to_blob() if hasattr(model, "to_blob") else bytes(model)
        self._blobbuf = C.create_string_buffer(self.blob, len(self.blob))
        self.pre = "orcv_" if vec else ("orcd_" if double else "orc_")
        self._lib = lib_vec() if vec else (lib_fast() if fast else lib())
        if getattr(mpc_cfg, "math_mode", "exact") == "fast" and not vec:
            transc_tables(self._lib)          # (the float64 and the timing builds refuse math_mode fast: their entry points return SDEMPC_EINVAL)
        self.H, self.P, self.m = mpc_cfg.horizon, mpc_cfg.num_particles, mpc_cfg.num_motors

    def _fn(self, name):
        return getattr(self._lib, self.pre + name)

    def rollout(self, x0, u, xref, noise, want_traj=False, want_mean=False):
        H, P = self.H, self.P
        x0, u, xref, noise = _f32(x0), _f32(u), _f32(xref), _f32(noise)
        assert u.shape == (H, self.m) and xref.shape == (H + 1, 13) and noise.shape == (P, H, 6)
        cost = C.c_double()
        traj = np.zeros((P, H + 1, 13), np.float32) if want_traj else None
        mean = np.zeros((H + 1, 13), np.float32) if want_mean else None
        rc = self._fn("rollout")(C.byref(self.cfg), self._blobbuf, _fp(x0), _fp(u), _fp(xref), _fp(noise), C.byref(cost),
                                 _fp(traj) if want_traj else None, _fp(mean) if want_mean else None)
        assert rc == 0, rc
        return cost.value, traj, mean

    def grad(self, x0, u, xref, noise):
        x0, u, xref, noise = _f32(x0), _f32(u), _f32(xref), _f32(noise)
        cost = C.c_double()
        g = np.zeros((self.H, self.m), np.float64)
        rc = self._fn("grad")(C.byref(self.cfg), self._blobbuf, _fp(x0), _fp(u), _fp(xref), _fp(noise), C.byref(cost), _dp(g))
        assert rc == 0, rc
        return cost.value, g

    def cost_du(self, x0, u64, xref, noise):
        x0, xref, noise = _f32(x0), _f32(xref), _f32(noise)
        u64 = np.ascontiguousarray(u64, dtype=np.float64)
        cost = C.c_double()
        rc = self._fn("cost_du")(C.byref(self.cfg), self._blobbuf, _fp(x0), _dp(u64), _fp(xref), _fp(noise), C.byref(cost))
        assert rc == 0, rc
        return cost.value

    def solve(self, x0, xref, noise, u_init, stepsize_in, trace_cap=0):
        x0, xref, noise, u_init = _f32(x0), _f32(xref), _f32(noise), _f32(u_init)
        uopt = np.zeros((self.H, self.m), np.float32)
        xevol = np.zeros((self.H + 1, 13), np.float32)
        info = np.zeros(8, np.float32)
        trace = np.zeros((trace_cap, 4), np.float32) if trace_cap else None
        rc = self._fn("solve")(C.byref(self.cfg), self._blobbuf, _fp(x0), _fp(xref), _fp(noise), _fp(u_init),
                               C.c_float(stepsize_in), _fp(uopt), _fp(xevol), _fp(info),
                               _fp(trace) if trace_cap else None, C.c_int(trace_cap))
        assert rc == 0, rc
        return uopt, xevol, info, trace

    def solve_batch(self, x0, xref, noise, u_init, stepsize_in):
        B = x0.shape[0]
        x0, xref, noise, u_init, stepsize_in = _f32(x0), _f32(xref), _f32(noise), _f32(u_init), _f32(stepsize_in)
        uopt = np.zeros((B, self.H, self.m), np.float32)
        xevol = np.zeros((B, self.H + 1, 13), np.float32)
        info = np.zeros((B, 8), np.float32)
        rc = self._fn("solve_batch")(C.byref(self.cfg), self._blobbuf, C.c_int(B), _fp(x0), _fp(xref), _fp(noise), _fp(u_init),
                                     _fp(stepsize_in), _fp(uopt), _fp(xevol), _fp(info))
        assert rc == 0, rc
        return uopt, xevol, info

    def step(self, x, u, xi, t=0):
        x, u, xi = _f32(x), _f32(u), _f32(xi)
        xn = np.zeros(13, np.float32)
        eta = C.c_float()
        rc = self._fn("step")(C.byref(self.cfg), self._blobbuf, _fp(x), _fp(u), _fp(xi), C.c_int(t), _fp(xn), C.byref(eta))
        assert rc == 0, rc
        return xn, eta.value
