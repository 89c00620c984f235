"""Minimal stand-in for the three JAX entry points sde_control.py touches, so the reference node can
run unmodified apart from its import lines:

    jax.jit(f).lower(*a, **k).compile()      sde_control.py:694,702,713   -> returns f itself
    jax.random.PRNGKey(seed)                 sde_control.py:338,698       -> uint32[2]
    jax.random.split(key, n)                 sde_control.py:341           -> uint32[n,2]

Nothing is traced or compiled here: the solver callables already dispatch to pre-built HIP kernels.
Keys follow JAX's threefry2x32 conventions (prng.py; pinned by public known-answer values)."""
import logging
import os
import warnings

import numpy as np

from . import prng as _prng


_log = logging.getLogger("sde4mbrl_px4_amd")


class _Compiled:
    """What `.lower(...).compile()` returns.

    The reference builds its solvers in the parent and runs each compiled callable ONCE there, before it forks the worker — to warm
    it up and to learn the output shapes (sde_control.py:706-707, :717-719; the fork is :723-728). A real solve at that point would
    initialise HIP in the parent, and HIP state does not survive fork(). So the FIRST call of a compiled callable, when it is made by
    the process that compiled it, is answered by the callable's `shape_probe_*` twin: outputs of the right shapes and dtypes, no GPU
    work — announced by a RuntimeWarning and a WARNING-level log line, and marked in the returned opt_state (num_steps 0, NaN costs). Every later call — in the forked `mpc_process`, or in the same process for in-process users — runs the real
    solver. SDEMPC_PREFORK=solve switches the probe off (every call is real: only for parents that never fork)."""

    def __init__(self, f):
        self._f = f
        self._pid = os.getpid()
        self._calls_here = 0

    def __call__(self, *a, **k):
        if os.getpid() == self._pid:
            self._calls_here += 1
            if self._calls_here == 1 and os.environ.get("SDEMPC_PREFORK", "shape") != "solve":
                name = getattr(self._f, "__name__", "")
                probe = getattr(getattr(self._f, "__self__", None), "shape_probe_" + name, None)
                if probe is not None:
                    msg = (f"sde4mbrl_px4_amd: the first call of the compiled `{name}` in the process that compiled it is answered by a "
                           "SHAPE PROBE, not a solve (no GPU work: HIP must not be initialised before the reference node forks its "
                           "worker, sde_control.py:717-728). Its outputs have the right shapes only: uopt is the warm start, "
                           "opt_state.num_steps is 0 and its costs are NaN. In-process users who want this call solved set "
                           "SDEMPC_PREFORK=solve.")
                    warnings.warn(msg, RuntimeWarning, stacklevel=2)
                    _log.warning(msg)
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
        return _prng.PRNGKey(seed)

    @staticmethod
    def split(key, num=2):
        return _prng.split(key, num)


numpy = np
