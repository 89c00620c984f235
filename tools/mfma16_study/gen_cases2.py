"""Second round of tiles for the MFMA numerics study: the accumulator-dominant regime near rounding ties.

Experiment (i, j) of a tile: C = +-(1.m) 2^E; product a = +-(n + 1/2) u 2^s at ka (u = 2^(E-23): C's ulp; s in {-1, 0});
product b = +-2^-q u 2^s at kb; optionally product c = 2^(E-d) at kc (sets the largest product exponent d below C).
Row i carries n, the signs of a and b; column j carries q, s.  C's mantissa pattern varies per element.
python tools/mfma16_study/gen_cases2.py OUTDIR
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_cases import Fmt, write


def accdom(F, seed, T=320):
    rng = np.random.default_rng(seed)
    A = np.zeros((T, 32, 16)); B = np.zeros((T, 16, 32)); C = np.zeros((T, 32, 32), np.float32)
    meta = np.zeros((T, 8), np.int32)
    for t in range(T):
        E = int(rng.choice([0, 3, -5])) if F.name == "bf16" else int(rng.choice([20, 18]))
        same_chunk = t % 4 != 3
        base = int(rng.integers(0, 2)) * 8
        ks = base + rng.permutation(8)[:3] if same_chunk else np.array([rng.integers(0, 8), 8 + rng.integers(0, 8), rng.integers(0, 16)])
        ka, kb, kc = int(ks[0]), int(ks[1]), int(ks[2])
        if kc in (ka, kb):
            kc = [k for k in range(16) if k not in (ka, kb)][0]
        d = int(rng.choice([-1, 0, 1, 2, 4, 6, 7, 8, 9, 10, 12, 16])) if t % 2 else -99       # -99: no product c
        qlo = int(rng.choice([1, 5, 9]))
        meta[t] = [E, ka, kb, kc, d, qlo, 0, 0]
        n = rng.integers(0, 128, 32); sa = rng.integers(0, 2, 32) * 2 - 1; sb = rng.integers(0, 2, 32) * 2 - 1
        q = qlo + (np.arange(32) % 16); s = -(np.arange(32) // 16)
        mb_ = 1.0 + rng.integers(0, 4, 32) / 4.0        # b's mantissa: 1, 1.25, 1.5, 1.75 (more low bits to see)
        def split(te):                                 # 2^te = 2^ea * 2^eb with both factors in the format's range
            te = np.asarray(te); ea = np.clip(te // 2, -12, 6); return ea, te - ea
        ea, eb = split(np.full(32, E - 23) )
        A[t, :, ka] = sa * (n + 0.5) * np.exp2(ea.astype(np.float64))      # 8 significant bits
        B[t, ka, :] = np.exp2((eb + s).astype(np.float64))
        ea, eb = split(E - 23 + s - q)
        A[t, :, kb] = sb * mb_ * np.exp2(float(ea.min()))
        B[t, kb, :] = np.exp2((E - 23 + s - q - ea.min()).astype(np.float64))
        if d != -99:
            ea, eb = split(E - d)
            A[t, :, kc] = np.exp2(float(ea)); B[t, kc, :] = np.exp2(float(eb))
        pat = rng.integers(0, 6, (32, 32))
        m = rng.integers(0, 1 << 23, (32, 32))
        m = np.where(pat == 0, 0, m)                       # power of two
        m = np.where(pat == 1, (1 << 23) - 1, m)           # all ones
        m = np.where(pat == 2, rng.integers(0, 4, (32, 32)), m)   # just above a power of two
        sc = rng.integers(0, 2, (32, 32)) * 2 - 1
        C[t] = (sc * (1.0 + m / float(1 << 23)) * 2.0 ** E).astype(np.float32)
    return (A, B, C), meta


def boundary(F, seed, T=256):
    """cancellation at the regime boundary: C = -(1.m) 2^(E+d) is NOT cancelled; product X = 2^E, -X at k0, k1; small p2 with bits"""
    rng = np.random.default_rng(seed)
    A = np.zeros((T, 32, 16)); B = np.zeros((T, 16, 32)); C = np.zeros((T, 32, 32), np.float32)
    meta = np.zeros((T, 8), np.int32)
    for t in range(T):
        ks = rng.permutation(8)[:3] + 8 * int(rng.integers(0, 2))
        d = int(rng.integers(-2, 14))
        if F.name == "f16": d += 0
        meta[t] = [0, ks[0], ks[1], ks[2], d, 0, 0, 0]
        A[t, :, ks[0]] = 1.0; B[t, ks[0], :] = 1.0
        A[t, :, ks[1]] = -1.0; B[t, ks[1], :] = 1.0
        ea = -rng.integers(0, 14, 32); eb = -rng.integers(0, 14, 32)
        A[t, :, ks[2]] = F.rand(rng, 32, 0, 0) * np.exp2(ea.astype(np.float64))
        B[t, ks[2], :] = F.rand(rng, 32, 0, 0) * np.exp2(eb.astype(np.float64))
        m = rng.integers(0, 1 << 23, (32, 32)); m = np.where(rng.random((32, 32)) < 0.3, 0, m)
        sc = rng.integers(0, 2, (32, 32)) * 2 - 1
        C[t] = (sc * (1.0 + m / float(1 << 23)) * 2.0 ** d).astype(np.float32)
    return (A, B, C), meta


def rand_mix(F, seed, T=512):
    from gen_cases import rand_f32
    rng = np.random.default_rng(seed + 7)
    A = F.rand(rng, (T, 32, 16), -10, 1); B = F.rand(rng, (T, 16, 32), -10, 1); C = rand_f32(rng, (T, 32, 32), -16, 4)
    # sparsify a third of the tiles (few active products: regimes with a dominant accumulator)
    keep = rng.random((T, 1, 16)) < 0.35
    A[: T // 3] = np.where(keep[: T // 3], A[: T // 3], 0.0)
    return (A, B, C), np.zeros((T, 8), np.int32)


if __name__ == "__main__":
    outdir = sys.argv[1]
    os.makedirs(outdir, exist_ok=True)
    for name, seed in (("f16", 303), ("bf16", 404)):
        F = Fmt(name)
        metas = {}
        for fam, fn in (("accdom", accdom), ("boundary", boundary), ("rand_mix", rand_mix)):
            (A, B, C), meta = fn(F, seed)
            write(os.path.join(outdir, f"{name}_{fam}.bin"), F, A, B, C)
            metas[fam] = meta
            print(name, fam, A.shape[0], "tiles")
        np.savez(os.path.join(outdir, f"{name}_meta2.npz"), **metas)
