"""Test tiles for tools/mfma16_study/mfma16_probe (offline study of v_mfma_f32_32x32x16_{f16,bf16} numerics).

Every tile is A[32][16], B[16][32] (raw 16-bit patterns) and C[32][32] (f32); element (i, j) of a tile is one experiment
D_ij = sum_k A[i][k] * B[k][j] + C[i][j].  Families (python tools/mfma16_study/gen_cases.py OUTDIR):

  rand_narrow   operands ~ 2^[-2,1], C ~ 2^[-3,3] or 0              (what the MLP contractions look like)
  rand_wide     operands ~ 2^[-7,7], C ~ 2^[-20,20]
  single        one non-zero product per experiment at position k0, C at every offset   (rounding of p + C)
  two           two non-zero products (k0, k1) at every offset, C = 0                    (grouping, rounding between groups)
  cancel        p[k0] = +X, p[k1] = -X, p[k2] = small with full mantissa, C = 0          (alignment width / truncation)
  cancel_c      C = -X, p[k0] = +X, p[k2] = small                                        (where C enters)
  cancel2       p[k0] = +X, p[k1] = -X, two small products at k2, k3                     (how small addends combine)
  ties          X = 1 (+ odd ulp) and n products of a quarter / half ulp                 (fused or sequential rounding)
  subnormal     sub-normal operands, results near the f32 sub-normal range
The generator is deterministic (numpy PCG64, fixed seeds); fit_model.py regenerates nothing: it reads the files written here.
"""
import os, sys
import numpy as np

MAGIC = 0x4D464D41


class Fmt:
    def __init__(self, name):
        self.name = name
        if name == "f16":
            self.mb, self.emin, self.emax, self.dt = 10, -14, 15, 0
        else:
            self.mb, self.emin, self.emax, self.dt = 7, -126, 127, 1

    def bits(self, v):
        """float64 array of exactly representable values -> uint16 patterns"""
        v = np.asarray(v, np.float64)
        if self.name == "f16":
            h = v.astype(np.float16)
            assert np.array_equal(h.astype(np.float64), v), "not representable in f16"
            return h.view(np.uint16)
        f = v.astype(np.float32)
        assert np.array_equal(f.astype(np.float64), v)
        u = f.view(np.uint32)
        assert not np.any(u & 0xFFFF), "not representable in bf16"
        return (u >> 16).astype(np.uint16)

    def rand(self, rng, shape, elo, ehi, full=True):
        """sign * (1 + m / 2^mb) * 2^e, e uniform in [elo, ehi] (clipped to the normal range)"""
        elo, ehi = max(elo, self.emin), min(ehi, self.emax)
        m = rng.integers(0, 1 << self.mb, shape) if full else np.zeros(shape, np.int64)
        e = rng.integers(elo, ehi + 1, shape)
        s = rng.integers(0, 2, shape) * 2 - 1
        return s * (1.0 + m / float(1 << self.mb)) * np.exp2(e.astype(np.float64))


def rand_f32(rng, shape, elo, ehi):
    m = rng.integers(0, 1 << 23, shape)
    e = rng.integers(elo, ehi + 1, shape)
    s = rng.integers(0, 2, shape) * 2 - 1
    return (s * (1.0 + m / float(1 << 23)) * np.exp2(e.astype(np.float64))).astype(np.float32)


def families(F, seed):
    rng = np.random.default_rng(seed)
    out = {}
    Z = lambda T: (np.zeros((T, 32, 16)), np.zeros((T, 16, 32)), np.zeros((T, 32, 32), np.float32))

    # ---- random
    T = 256
    A = F.rand(rng, (T, 32, 16), -2, 1); B = F.rand(rng, (T, 16, 32), -2, 1); C = rand_f32(rng, (T, 32, 32), -3, 3)
    C[: T // 4] = 0.0
    out["rand_narrow"] = (A, B, C)
    A = F.rand(rng, (T, 32, 16), -7, 7); B = F.rand(rng, (T, 16, 32), -7, 7); C = rand_f32(rng, (T, 32, 32), -20, 20)
    out["rand_wide"] = (A, B, C)

    big = 14 if F.name == "f16" else 30          # X = 2^(2 big)
    span = 2 * big + (26 if F.name == "f16" else 40)     # offsets reachable below X

    # ---- single product + C at every offset: k0 = tile % 16
    T = 64
    A, B, C = Z(T)
    for t in range(T):
        k0 = t % 16
        A[t, :, k0] = F.rand(rng, 32, -4, 4); B[t, k0, :] = F.rand(rng, 32, -4, 4)
        C[t] = rand_f32(rng, (32, 32), -36, 36)
    out["single"] = (A, B, C)

    # ---- two products, C = 0
    T = 240
    A, B, C = Z(T)
    pairs = [(a, b) for a in range(16) for b in range(16) if a != b]
    for t in range(T):
        k0, k1 = pairs[t]
        A[t, :, k0] = F.rand(rng, 32, 0, 0); B[t, k0, :] = F.rand(rng, 32, 0, 0)
        A[t, :, k1] = F.rand(rng, 32, -min(big, 20), 0); B[t, k1, :] = F.rand(rng, 32, -min(big, 20), 0)
    out["two"] = (A, B, C)

    # ---- cancellation: +X, -X, small
    def cancel(T, with_c, nsmall):
        A, B, C = Z(T)
        meta = np.zeros((T, 4), np.int32)
        for t in range(T):
            ks = rng.permutation(16)[: 2 + nsmall]
            k0, k1 = int(ks[0]), int(ks[1])
            meta[t, : 2 + nsmall] = ks
            A[t, :, k0] = np.exp2(big); B[t, k0, :] = np.exp2(big)
            if with_c:
                C[t] = -np.exp2(2 * big); meta[t, 1] = -1
            else:
                A[t, :, k1] = -np.exp2(big); B[t, k1, :] = np.exp2(big)
            for k2 in ks[2:]:
                k2 = int(k2)
                # row i carries exponent big - (i-dependent), column j likewise: offsets 0 .. span below X
                ea = big - rng.integers(0, span // 2 + 1, 32); eb = big - rng.integers(0, span // 2 + 1, 32)
                A[t, :, k2] = F.rand(rng, 32, 0, 0) * np.exp2(np.clip(ea, F.emin, F.emax))
                B[t, k2, :] = F.rand(rng, 32, 0, 0) * np.exp2(np.clip(eb, F.emin, F.emax))
        return (A, B, C), meta
    out["cancel"], m1 = cancel(480, False, 1)
    out["cancel_c"], m2 = cancel(240, True, 1)
    out["cancel2"], m3 = cancel(480, False, 2)

    # ---- ties: X = 1 or 1 + 2^-23 via C; n products of 2^-25 / 2^-24 / 2^-26 at random positions
    T = 256
    A, B, C = Z(T)
    for t in range(T):
        n = 1 + t % 8
        ks = rng.permutation(16)
        use_c = (t // 8) % 2
        if use_c:
            C[t] = np.float32(1.0) + np.float32(2.0 ** -23) * rng.integers(0, 4, (32, 32)).astype(np.float32)
            tiny = ks[:n]
        else:
            A[t, :, ks[0]] = 1.0 + rng.integers(0, 2, 32) * 2.0 ** -F.mb; B[t, ks[0], :] = 1.0
            tiny = ks[1: 1 + n]
        for kk in tiny:
            A[t, :, kk] = np.exp2(-rng.integers(11, 14, 32).astype(np.float64)) * (rng.integers(0, 2, 32) * 2 - 1)
            B[t, kk, :] = np.exp2(-rng.integers(11, 14, 32).astype(np.float64))
    out["ties"] = (A, B, C)

    # ---- sub-normal operands / results
    T = 64
    A, B, C = Z(T)
    if F.name == "f16":
        sub = lambda shape: (rng.integers(0, 2, shape) * 2 - 1) * rng.integers(1, 1024, shape) * 2.0 ** -24
        A[:] = np.where(rng.random((T, 32, 16)) < 0.3, sub((T, 32, 16)), F.rand(rng, (T, 32, 16), -14, -8))
        B[:] = np.where(rng.random((T, 16, 32)) < 0.3, sub((T, 16, 32)), F.rand(rng, (T, 16, 32), -14, -8))
        C[:] = np.where(rng.random((T, 32, 32)) < 0.5, 0.0, rand_f32(rng, (T, 32, 32), -40, -20))
    else:
        sub = lambda shape: (rng.integers(0, 2, shape) * 2 - 1) * rng.integers(1, 128, shape) * 2.0 ** -133
        A[:] = np.where(rng.random((T, 32, 16)) < 0.3, sub((T, 32, 16)), F.rand(rng, (T, 32, 16), -70, -60))
        B[:] = np.where(rng.random((T, 16, 32)) < 0.3, sub((T, 16, 32)), F.rand(rng, (T, 16, 32), -70, -58))
        C[:] = np.where(rng.random((T, 32, 32)) < 0.5, 0.0, (rand_f32(rng, (T, 32, 32), -20, 0).astype(np.float64) * 2.0 ** -125).astype(np.float32))
    out["subnormal"] = (A, B, C)
    return out, {"cancel": m1, "cancel_c": m2, "cancel2": m3}


def write(path, F, A, B, C):
    T = A.shape[0]
    with open(path, "wb") as f:
        np.array([MAGIC, T, F.dt, 0], np.int32).tofile(f)
        a = F.bits(A); b = F.bits(B)
        for t in range(T):
            a[t].tofile(f); b[t].tofile(f); C[t].astype(np.float32).tofile(f)


if __name__ == "__main__":
    outdir = sys.argv[1]
    os.makedirs(outdir, exist_ok=True)
    for name, seed in (("f16", 101), ("bf16", 202)):
        F = Fmt(name)
        fam, meta = families(F, seed)
        for k, (A, B, C) in fam.items():
            write(os.path.join(outdir, f"{name}_{k}.bin"), F, A, B, C)
            print(name, k, A.shape[0], "tiles")
        np.savez(os.path.join(outdir, f"{name}_meta.npz"), **meta)
