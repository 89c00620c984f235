"""Minimal stand-in for the three JAX entry points sde_control.py touches, so the reference node can
run unmodified apart from its import lines:

    jax.jit(f).lower(*a, **k).compile()      sde_control.py:694,702,713   -> returns f itself
    jax.random.PRNGKey(seed)                 sde_control.py:338,698       -> uint32[2]
    jax.random.split(key, n)                 sde_control.py:341           -> uint32[n,2]

Nothing is traced or compiled here: the solver callables already dispatch to pre-built HIP kernels.
Key splitting is NOT threefry-compatible."""
import os

import numpy as np


class _Compiled:
    """What `.lower(...).compile()` returns. With SDEMPC_PREFORK=shape in the environment, calls made by the process that
    "compiled" the callable are treated as the reference's pre-fork warm-up / shape probes (sde_control.py:706,717): they are
    answered by the callable's `shape_probe` (no HIP call, outputs of the right shapes only); calls from any other process
    (the forked `mpc_process`, sde_control.py:723-728) run the real solver. Without the variable every call is real."""

    def __init__(self, f):
        self._f = f
        self._pid = os.getpid()

    def __call__(self, *a, **k):
        if os.environ.get("SDEMPC_PREFORK") == "shape" and os.getpid() == self._pid:
            probe = getattr(getattr(self._f, "__self__", None), "shape_probe_" + getattr(self._f, "__name__", ""), None)
            if probe is not None:
                return probe(*a, **k)
        return self._f(*a, **k)


class _Lowered:
    def __init__(self, f):
        self._f = f

    def compile(self):
        return _Compiled(self._f)


class _Jitted:
    def __init__(self, f):
        self._f = f

    def __call__(self, *a, **k):
        return self._f(*a, **k)

    def lower(self, *a, **k):
        return _Lowered(self._f)


def jit(f, **_):
    return _Jitted(f)


class random:  # noqa: N801 - mirrors the module name jax.random
    @staticmethod
    def PRNGKey(seed):
        return np.array([0, int(seed) & 0xFFFFFFFF], dtype=np.uint32)

    @staticmethod
    def split(key, num=2):
        g = np.random.default_rng([int(v) for v in np.asarray(key).reshape(-1)])
        return g.integers(0, 2 ** 32, size=(num, 2), dtype=np.uint32)


numpy = np
