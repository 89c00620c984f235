"""Key algebra of the reference's PRNG plumbing: JAX's default generator, threefry2x32 (SPEC.md §7.1).

The MPC node builds `jax.random.PRNGKey(seed)`, splits it three ways (sde_control.py:338-341) and threads one key into and
out of every solver call (sde_control.py:345-350,400-416,698,706,717). JAX is an external dependency that is not available
here; its key handling is a published, integer-exact algorithm (Salmon et al., SC'11; legacy non-partitionable counter layout
of `jax._src.prng`), restated below with numpy uint32 arithmetic and pinned in tests/test_prng_cpu.py by the Random123
known-answer vectors and by the split values printed in the JAX documentation. Host glue only: the noise itself is drawn on
the device (csrc/sdempc_prng.hip)."""
from __future__ import annotations

import numpy as np

_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))


def _rotl(x, r):
    return (x << np.uint32(r)) | (x >> np.uint32(32 - r))


def threefry2x32(key, x0, x1):
    """20-round threefry2x32 of counter words (x0, x1) under key uint32[2]; arrays broadcast."""
    key = np.asarray(key, dtype=np.uint32).reshape(2)
    x0 = np.array(x0, dtype=np.uint32, copy=True)
    x1 = np.array(x1, dtype=np.uint32, copy=True)
    ks = (key[0], key[1], key[0] ^ key[1] ^ np.uint32(0x1BD11BDA))
    with np.errstate(over="ignore"):
        x0 += ks[0]
        x1 += ks[1]
        for i in range(5):
            for r in _ROT[i & 1]:
                x0 += x1
                x1 = _rotl(x1, r)
                x1 ^= x0
            x0 += ks[(i + 1) % 3]
            x1 += ks[(i + 2) % 3] + np.uint32(i + 1)
    return x0, x1


def random_bits(key, n: int) -> np.ndarray:
    """n 32-bit words: counters 0..n-1 (a zero appended when n is odd); first half -> x0, second half -> x1; y0 ++ y1."""
    n = int(n)
    c = np.arange(n + (n & 1), dtype=np.uint32)
    c[n:] = 0
    h = c.size // 2
    y0, y1 = threefry2x32(key, c[:h], c[h:])
    return np.concatenate([y0, y1])[:n]


def PRNGKey(seed) -> np.ndarray:
    s = int(seed) & 0xFFFFFFFFFFFFFFFF
    return np.array([s >> 32, s & 0xFFFFFFFF], dtype=np.uint32)


def split(key, num: int = 2) -> np.ndarray:
    return random_bits(key, 2 * int(num)).reshape(int(num), 2)
