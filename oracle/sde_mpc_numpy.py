"""sde_mpc_numpy.py — SECOND, independent CPU restatement of SPEC.md (NumPy, particle-vectorised). TEST INFRASTRUCTURE ONLY.

Only tests/ may import this file (the same rule as for oracle/sde_mpc_oracle.c). PARITY UNPINNED against the reference for the
same reason as the C oracle: the arithmetic of the path is not in /root/reference (sde_control.py:12-13 imports it from the
un-vendored package sde4mbrl). What this file adds is a cross-check of the oracle itself: it is written from SPEC.md in another
language and another loop order (all particles at once instead of one particle at a time), and the tests require

  * forward rollout, particle x horizon tensor, mean trajectory, expected cost, control cost: BIT-IDENTICAL to the C oracle
    (SPEC.md §3-§6; float32 with an exact software fma, see `fma`);
  * the adjoint sweep and gradient assembly (§5.4-§5.5), written here a second time in float32 with the same software fma, all
    particles at once: gradient BIT-IDENTICAL to the C oracle's;
  * the accelerated proximal gradient loop of SPEC.md §8 on THIS file's own cost and gradient (`solve_own`: nothing of the C oracle
    takes part), own reductions (§6.2) and decision logic: BIT-IDENTICAL controls and telemetry;
  * and, independent of both hand-derived adjoints: the float32 gradient against reverse-mode automatic differentiation
    (torch.autograd, float64, libm activations) of the same model written here a third time (`torch_cost`).

Follows (reference): call shapes sde_control.py:702-719,400-416; YAML keys launch/iris_sitl_traj_mpc.yaml:8-85.
"""
import numpy as np

F = np.float32
U32 = np.uint32


# ------------------------------------------------------------------------------------------------------------------------------
# exact float32 fma on arrays: the product of two float32 is exact in float64; the float64 sum is corrected to round-to-odd
# with the TwoSum error term, and a round-to-odd value with >= 2 spare bits rounds to float32 exactly like the infinitely
# precise result (no double rounding).
# ------------------------------------------------------------------------------------------------------------------------------
def fma(a, b, c):
    a, b, c = np.asarray(a, F), np.asarray(b, F), np.asarray(c, F)
    p = a.astype(np.float64) * b.astype(np.float64)
    c64 = c.astype(np.float64)
    with np.errstate(invalid="ignore", over="ignore"):
        s = p + c64
        t = s - p
        e = (p - (s - t)) + (c64 - t)
    s = np.asarray(s)
    fix = np.isfinite(s) & np.isfinite(e) & (e != 0.0) & ((s.view(np.uint64) & np.uint64(1)) == 0)
    s = np.where(fix, np.nextafter(s, np.where(e > 0, np.inf, -np.inf)), s)
    with np.errstate(over="ignore", invalid="ignore"):
        return s.astype(F)


def bits(x):
    return np.asarray(x, F).view(U32)


def from_bits(u):
    return np.asarray(u, U32).view(F)


def clamp(x, lo, hi):
    """SPEC.md §3.6: !(x > lo) ? lo : (x > hi ? hi : x) — a NaN maps to lo."""
    x = np.asarray(x, F)
    with np.errstate(invalid="ignore"):
        return np.where(~(x > F(lo)), F(lo), np.where(x > F(hi), F(hi), x)).astype(F)


def rcp(d):
    d = np.asarray(d, F)
    y = from_bits(U32(0x7EF311C7) - bits(d))
    for _ in range(3):
        e = fma(-d, y, F(1.0))
        y = fma(y, e, y)
    return y


def rsqrt(a):
    a = np.asarray(a, F)
    y = from_bits(U32(0x5F3759DF) - (bits(a) >> U32(1)))
    h = F(0.5) * a
    for _ in range(3):
        t = y * y
        t = fma(-h, t, F(1.5))
        y = y * t
    return y


_EXP2 = [F(0.001327647129073739), F(0.009675540961325169), F(0.05550713092088699), F(0.24022120237350464), F(0.6931469440460205),
         F(1.0000001192092896)]


def exp2c(x, c):
    x = np.asarray(x, F)
    t2 = fma(x, F(c), F(12582912.0))
    n = t2 - F(12582912.0)
    f = fma(x, F(c), -n)
    p = np.full_like(f, _EXP2[0])
    for k in _EXP2[1:]:
        p = fma(p, f, k)
    with np.errstate(over="ignore"):
        return from_bits(bits(p) + (bits(t2) << U32(23)))


def tanh4(a):
    """a: [..., 4] — four values share one reciprocal (SPEC.md §3.4)."""
    d = [F(1.0) + exp2c(clamp(a[..., i], -9.0, 9.0), 2.885390043258667) for i in range(4)]
    p2 = d[0] * d[1]
    p3 = p2 * d[2]
    p4 = p3 * d[3]
    r = rcp(p4)
    r3 = r * p3
    r = r * d[3]
    r2 = r * p2
    r = r * d[2]
    r1 = r * d[0]
    r0 = r * d[1]
    return np.stack([fma(F(-2.0), ri, F(1.0)) for ri in (r0, r1, r2, r3)], axis=-1)


def tanh_units(a):
    """a: [..., 32] hidden pre-activations; groups of 4 consecutive units."""
    return tanh4(a.reshape(a.shape[:-1] + (8, 4))).reshape(a.shape)


def sigmoid(x):
    return rcp(F(1.0) + exp2c(clamp(x, -30.0, 30.0), -1.4426950216293335))


# ------------------------------------------------------------------------------------------------------------------------------
# SPEC.md §10a written a second time: what v_rcp_f32 / v_rsq_f32 / v_exp_f32 return, from the structure rules and the recorded binades
# (tests/golden/transc/*.i8.xz: differences of -1 / 0 / +1 unit in the last place against an IEEE-reproducible reference). Independent
# of oracle/transc_model.c in language and form (whole arrays at once, blocks read straight from the committed files).
# ------------------------------------------------------------------------------------------------------------------------------
class HwTransc:
    def __init__(self, block_dir):
        self.dir, self.blocks = block_dir, {}

    def delta(self, name, m):
        if name not in self.blocks:
            import lzma, os
            self.blocks[name] = np.frombuffer(lzma.decompress(open(os.path.join(self.dir, name + ".i8.xz"), "rb").read()), np.int8)
        return self.blocks[name][m].astype(np.int64)

    @staticmethod
    def _fields(x):
        u = bits(np.asarray(x, F)).astype(np.int64)
        return u, (u >> 23) & 255, u & 0x7FFFFF, u >> 31

    @staticmethod
    def _pack(sign, expo, mant):
        return from_bits(((sign << 31) | (expo << 23) | mant).astype(np.uint32))

    def rcp(self, x):
        x = np.atleast_1d(np.asarray(x, F))
        u, e, m, s = self._fields(x)
        one_m = self._pack(0 * s, 127 + 0 * e, m)                                   # 1.m
        with np.errstate(all="ignore"):
            t = bits((1.0 / one_m.astype(np.float64)).astype(F)).astype(np.int64) + self.delta("rcp_s0_e127", m)
        ef = ((t >> 23) & 255) + 127 - e
        out = self._pack(s, np.clip(ef, 0, 254), t & 0x7FFFFF)
        out = np.where(ef < 1, self._pack(s, 0 * e, 0 * m), out)                    # below the normal range: +-0
        out = np.where(e == 0, self._pack(s, 255 + 0 * e, 0 * m), out)              # +-0, sub-normal: +-inf
        out = np.where(e == 255, np.where(m != 0, from_bits((u | 0x400000).astype(np.uint32)), self._pack(s, 0 * e, 0 * m)), out)
        return out

    def rsq(self, x):
        x = np.atleast_1d(np.asarray(x, F))
        u, e, m, s = self._fields(x)
        ee = e - 127
        par = ee & 1
        k = (ee - par) >> 1
        x0 = self._pack(0 * s, 127 + par, m)                                        # 2^p * 1.m in [1, 4)
        with np.errstate(all="ignore"):
            ref = bits((1.0 / np.sqrt(x0.astype(np.float64))).astype(F)).astype(np.int64)
        t = ref + np.where(par == 1, self.delta("rsq_s0_e128", m), self.delta("rsq_s0_e127", m))
        out = self._pack(0 * s, np.clip(((t >> 23) & 255) - k, 0, 254), t & 0x7FFFFF)
        nan = from_bits(np.full(x.shape, 0xFFC00000, np.uint32))
        out = np.where(s == 1, nan, out)
        out = np.where(e == 0, self._pack(s, 255 + 0 * e, 0 * m), out)
        out = np.where(e == 255, np.where(m != 0, from_bits((u | 0x400000).astype(np.uint32)), np.where(s == 1, nan, F(0.0))), out)
        return out

    @staticmethod
    def _exp2_f64(x):
        """2^x in float64 from multiplications and additions (Taylor series of 2^(r + 1/2), |r| <= 1/2, degree 20): the reference of the exp blocks"""
        n = np.floor(x)
        t = ((x - n) - 0.5) * 0.6931471805599453
        p = np.full_like(t, 1.0 / 2432902008176640000.0)
        for k in range(19, 0, -1):
            c = 1.0
            for j in range(2, k + 1):
                c *= j
            p = p * t + 1.0 / c
        p = p * t + 1.0
        return np.ldexp(p * 1.4142135623730951, n.astype(np.int64))

    def _exp_block(self, e, s, m, x):
        """reference + recorded difference for inputs whose exponent field is e (97..127), all of one (e, sign) per call"""
        t = bits(self._exp2_f64(x.astype(np.float64)).astype(F)).astype(np.int64)
        return t + self.delta(f"exp_s{s}_e{e}", m)

    def exp2(self, x):
        x = np.atleast_1d(np.asarray(x, F))
        u, e, m, s = self._fields(x)
        out = np.ones(x.shape, F)                                                   # |x| < 2^-30 (zero, sub-normal): exactly 1
        for ee in np.unique(e[(e >= 97) & (e <= 127)]):
            for ss in (0, 1):
                sel = (e == ee) & (s == ss)
                if sel.any():
                    out[sel] = from_bits(self._exp_block(int(ee), ss, m[sel], x[sel]).astype(np.uint32))
        big = (e >= 128) & (e <= 133)                                               # 2 <= |x| < 128: the answer of +-(1 + frac), exponent moved by +-k
        if big.any():
            fixed = (m[big] | 0x800000) << (e[big] - 127)
            k, frac, sb = (fixed >> 23) - 1, fixed & 0x7FFFFF, s[big]
            x0 = self._pack(sb, 127 + 0 * sb, frac)
            t = np.empty(frac.shape, np.int64)
            for ss in (0, 1):
                sel = sb == ss
                if sel.any():
                    t[sel] = self._exp_block(127, ss, frac[sel], x0[sel])
            ef = ((t >> 23) & 255) + np.where(sb == 1, -k, k)
            r = self._pack(0 * sb, np.clip(ef, 0, 254), t & 0x7FFFFF)
            r = np.where(ef < 1, F(0.0), np.where(ef > 254, F(np.inf), r))
            out[big] = r
        out = np.where((e >= 134) & (e < 255), np.where(s == 1, F(0.0), F(np.inf)), out)
        out = np.where(e == 255, np.where(m != 0, from_bits((u | 0x400000).astype(np.uint32)), np.where(s == 1, F(0.0), F(np.inf))), out)
        return out.astype(F)


def korder():
    """SPEC.md §4: k(r, h) = (r & 3) + 8 (r >> 2) + 4 h, r = 0..15, inner h = 0, 1."""
    return [(r & 3) + 8 * (r >> 2) + 4 * h for r in range(16) for h in (0, 1)]


def half_sums(w, a):
    """(P_0 + P_1) of SPEC.md §4: w [32], a [..., 32] -> [...]"""
    P = []
    for h in (0, 1):
        acc = np.zeros(a.shape[:-1], F)
        for r in range(16):
            k = (r & 3) + 8 * (r >> 2) + 4 * h
            acc = fma(w[k], a[..., k], acc)
        P.append(acc)
    return P[0] + P[1]


# ------------------------------------------------------------------------------------------------------------------------------
# SPEC.md §9a written a second time: v_mfma_f32_32x32x16_{f16,bf16} in exact integer arithmetic (Python ints; scalar, slow, for small
# cases). Independent of oracle/mfma16_model.c in language and in form (no fast paths, no fixed-width integers); pinned by the same
# recorded hardware answers (tests/golden/mfma16_*.npz).
# ------------------------------------------------------------------------------------------------------------------------------
def _dec16(bits16, bf16):
    """-> (signed integer significand m, exponent e of its lsb, exponent ex that enters the exponent sum); finite operands only"""
    h = int(bits16)
    sgn = h >> 15
    if bf16:
        ef, f, nb, bias = (h >> 7) & 255, h & 127, 7, 127
    else:
        ef, f, nb, bias = (h >> 10) & 31, h & 1023, 10, 15
    assert ef != (255 if bf16 else 31), "non-finite operand"
    m = (f | (1 << nb)) if ef else f
    eu = (ef - bias) if ef else (1 - bias)
    return (-m if sgn else m), eu - nb, eu


import struct as _struct


def _f32_bits(x):
    return _struct.unpack("<I", _struct.pack("<f", x))[0]


def _f32_fields(x):
    u = _f32_bits(x)
    ef, f = (u >> 23) & 255, u & 0x7FFFFF
    m = (f | 0x800000) if ef else f
    return (-m if u >> 31 else m), (ef if ef else 1) - 150


def _round_rne_f32(v, g):
    """exact value v * 2^g (python ints) -> float32, round to nearest even, sub-normals on the 2^-149 grid"""
    if v == 0:
        return F(0.0)
    a = abs(v)
    E = g + a.bit_length() - 1
    lsb = max(E - 23, -149)
    sh = lsb - g
    if sh <= 0:
        q = a << (-sh)
    else:
        q, rem, half = a >> sh, a & ((1 << sh) - 1), 1 << (sh - 1)
        if rem > half or (rem == half and (q & 1)):
            q += 1
    if q >> 24:                                       # the rounding carried into the next binade
        q >>= 1
        lsb += 1
    if q < (1 << 23):
        u = q                                         # sub-normal result (lsb is -149)
    else:
        eb = lsb + 150
        u = 0x7F800000 if eb >= 255 else ((eb << 23) | (q & 0x7FFFFF))
    return F(_struct.unpack("<f", _struct.pack("<I", u | (0x80000000 if v < 0 else 0)))[0])


def mfma16_group(products, acc):
    """One group (eight products) of SPEC.md §9a on top of the running value acc: products = [(m, e, ex_sum), ...]"""
    nz = [(m, e, x) for (m, e, x) in products if m != 0]
    if not nz:
        return F(acc)
    g = max(x for _, _, x in nz) - 24                # grid: 24 bits below the largest exponent SUM
    S = 0
    for m, e, _ in nz:
        sh = g - e
        mag = abs(m) << (-sh) if sh <= 0 else abs(m) >> sh          # truncation toward zero
        S += -mag if m < 0 else mag
    am, ae = _f32_fields(acc)
    sh = g - ae
    v = S + ((am << (-sh)) if sh <= 0 else (am >> sh))                # the running value joins by a two's-complement floor
    if v == 0:
        return F(0.0)
    bl = abs(v).bit_length()
    if bl > 32:                                       # 32 leading bits, floor below them
        v >>= bl - 32
        g += bl - 32
    return _round_rne_f32(v, g)


def mfma16_dot(bf16, a16, b16, c):
    """D = sum_k a_k b_k + C of one output element: k = 0..7, then k = 8..15"""
    pr = []
    for k in range(16):
        ma, ea, xa = _dec16(a16[k], bf16)
        mb, eb, xb = _dec16(b16[k], bf16)
        pr.append((ma * mb, ea + eb, xa + xb))
    return mfma16_group(pr[8:], mfma16_group(pr[:8], F(c)))


def bf16_limbs(x):
    """SPEC.md §9b: three truncations to bf16, both subtractions exact; -> three uint16 patterns"""
    x = F(x)
    out = []
    for _ in range(3):
        hi = U32(int(np.asarray(x, F).view(U32)) & 0xFFFF0000)
        out.append(int(hi) >> 16)
        x = F(x - np.asarray(hi, U32).view(F))
    return out


def f16_rtz_bits(x):
    """SPEC.md §9: round toward zero to binary16 (finite overflow saturates at 65504); -> uint16 pattern"""
    x = F(x)
    h = np.float16(x)
    if np.isinf(h) and np.isfinite(x):
        h = np.float16(np.copysign(65504.0, x))
    elif abs(F(h)) > abs(x):
        h = np.nextafter(h, np.float16(0.0))
    return int(np.asarray(h, np.float16).view(np.uint16))


def slot_unit(hf, k):
    r, h = 8 * hf + (k & 7), k >> 3
    return (r & 3) + 8 * (r >> 2) + 4 * h


def x3_contract(W, v, c):
    """out[i] = c[i] + sum_k W[i][k] v[k] as the twelve instructions of SPEC.md §9b (W [32,32], v [32], c [32] or None)"""
    WA, VB = (2, 1, 1, 0, 0, 0), (0, 1, 0, 2, 1, 0)
    vl = [bf16_limbs(v[k]) for k in range(32)]
    out = np.zeros(32, F)
    for i in range(32):
        wl = [bf16_limbs(W[i, k]) for k in range(32)]
        acc = F(c[i]) if c is not None else F(0.0)
        for s6 in range(6):
            for hf in range(2):
                a16 = [wl[slot_unit(hf, k)][WA[s6]] for k in range(16)]
                b16 = [vl[slot_unit(hf, k)][VB[s6]] for k in range(16)]
                acc = mfma16_dot(True, a16, b16, acc)
        out[i] = acc
    return out


def f16_limbs(x):
    """SPEC.md §10c: two roundings to nearest even to binary16 (NumPy's float16 conversion), the subtraction exact; -> two uint16 patterns"""
    x = F(x)
    with np.errstate(all="ignore"):
        h1 = np.float16(x)
        h2 = np.float16(F(x - F(h1)))
    return [int(np.asarray(h1, np.float16).view(np.uint16)), int(np.asarray(h2, np.float16).view(np.uint16))]


def h2_contract(W, v, c):
    """out[i] = c[i] + sum_k W[i][k] v[k] as the eight f16 instructions of SPEC.md §10c: limb products (w2,v2) (w2,v1) (w1,v2) (w1,v1)"""
    WA, VB = (1, 1, 0, 0), (1, 0, 1, 0)
    vl = [f16_limbs(v[k]) for k in range(32)]
    out = np.zeros(32, F)
    for i in range(32):
        wl = [f16_limbs(W[i, k]) for k in range(32)]
        acc = F(c[i]) if c is not None else F(0.0)
        for s4 in range(4):
            for hf in range(2):
                a16 = [wl[slot_unit(hf, k)][WA[s4]] for k in range(16)]
                b16 = [vl[slot_unit(hf, k)][VB[s4]] for k in range(16)]
                acc = mfma16_dot(False, a16, b16, acc)
        out[i] = acc
    return out


# ------------------------------------------------------------------------------------------------------------------------------
class Model:
    """SPEC.md §2 blob."""

    def __init__(self, blob: bytes):
        hd = np.frombuffer(blob, np.int32, 16)
        assert hd[0] == 0x31454453 and hd[1] == 1
        self.m = int(hd[2])
        f = np.frombuffer(blob, F, 2120, 64).copy()
        self.inv_mass, self.grav = f[0], f[1]
        self.J, self.iJ = f[2:5], f[5:8]
        self.ct2, self.ct1, self.ct0, self.cm2, self.cm1 = f[8:13]
        self.rx, self.ry, self.dir = f[16:24], f[24:32], f[32:40]
        self.sF, self.sT = f[40:43], f[43:46]
        self.sigma = f[48:54]
        self.W1z = f[56:440].reshape(64, 6)
        self.b1 = f[440:504]
        self.W1u = f[504:760].reshape(32, 8)
        self.W2 = f[760:1784].reshape(32, 32)
        self.b2 = f[1784:1816]
        self.W3 = f[1816:2072].reshape(8, 32)
        self.b3 = f[2072:2080]
        self.w3n = f[2080:2112]
        self.b3n = f[2112]


class Restatement:
    """One (config, model): rollout / cost in float32, bit for bit as SPEC.md writes them; APG loop of §8."""

    def __init__(self, cfg, model, hw=None):
        self.cfg = cfg
        self.M = Model(model.to_blob() if hasattr(model, "to_blob") else bytes(model))
        self.H, self.P, self.m = cfg.horizon, cfg.num_particles, cfg.num_motors
        self.mlp = cfg.mlp_dtype                    # "f32", "f16" (SPEC.md §9) or "f32x3" (§9b); the matrix-pipe modes go through mfma16_dot: small cases only
        q = None
        if self.mlp == "f16":                        # layer-1 (state inputs) and layer-2 weights live in fp16
            q = np.vectorize(lambda w: np.asarray(np.uint16(f16_rtz_bits(w))).view(np.float16).astype(F))
            self.M.W1z, self.M.W2 = q(self.M.W1z).astype(F), q(self.M.W2).astype(F)
        # math_mode fast (SPEC.md §10, §10b): activations on the modelled instructions (hw: a HwTransc), the hidden activation kept as r = 1 / (1 + 2^a')
        # with its pre-scale and affine map in the forward weights; V holds what the vector-Jacobian products use
        M = self.M
        self.fast = getattr(cfg, "math_mode", "exact") == "fast"
        self.hw = hw
        self.V = dict(W1z=M.W1z, W1u=M.W1u, W2=M.W2, W3=M.W3, w3n=M.w3n)
        if self.fast:
            assert hw is not None, "math_mode fast needs the recorded instruction answers (HwTransc)"
            c = F(2.885390043258667)
            self.V = dict(W1z=M.W1z.copy(), W1u=M.W1u.copy(), W2=(F(4) * M.W2).astype(F), W3=(F(4) * M.W3).astype(F), w3n=(F(4) * M.w3n).astype(F))
            cW2 = (c * M.W2).astype(F)
            W1z = (c * M.W1z).astype(F)
            if self.mlp == "f16":
                cW2, W1z = q(cW2).astype(F), q(W1z).astype(F)
            b2 = (c * M.b2).astype(F)
            for k in range(32):                      # ascending k, one rounding per addition
                b2 = (b2 + cW2[:, k]).astype(F)
            b3, b3n = M.b3.copy(), F(M.b3n)
            for k in range(32):
                b3 = (b3 + M.W3[:, k]).astype(F)
                b3n = F(b3n + M.w3n[k])
            M.W1z, M.b1, M.W1u = W1z, (c * M.b1).astype(F), (c * M.W1u).astype(F)
            M.W2, M.b2 = (F(-2) * cW2).astype(F), b2
            if self.mlp == "f16":                    # the re-quantised c w is what the forward pass evaluates: its derivative uses those weights (§10b)
                self.V["W1z"] = (M.W1z / c).astype(F)
                self.V["W2"] = ((F(-2) * M.W2).astype(F) / c).astype(F)
            M.W3, M.b3, M.w3n, M.b3n = (F(-2) * M.W3).astype(F), b3, (F(-2) * M.w3n).astype(F), b3n
        # SPEC.md §10e (math_mode fast + f32x3): the adjoint's contractions from binary16 limbs behind a per-particle power-of-two scale; its offset from
        # bounds on the scaled quantities (float32, absolute column sums in ascending index order)
        self.adjmp = self.fast and self.mlp in ("f32x3", "f16")
        if self.adjmp:
            B3, Bn, C2 = F(0), F(0), F(0)
            for k in range(32):
                s3, s2 = F(0), F(0)
                for i in range(6):
                    s3 = F(s3 + abs(M.W3[i, k]))
                for j in range(32):
                    s2 = F(s2 + abs(self.V["W2"][j, k]))
                B3, C2, Bn = max(B3, s3), max(C2, s2), max(Bn, F(abs(M.w3n[k])))
            bound = max(F(B3 * F(0.25)), F(Bn * F(0.25)), F(F(C2 * F(B3 * F(0.25))) * F(0.25)))
            self.adj_eoff = 10
            if bound > 0 and np.isfinite(bound):
                self.adj_eoff = min(10, 14 - int(np.frexp(bound)[1]))
            self.adj_eoff = max(self.adj_eoff, -40)
            # A-operand matrices of the two narrow contractions: rows 0..5 W1z^T (density / drift tile), drift rows 6..6+m-1 W1u^T, the rest zero
            self.Zn = np.zeros((32, 32), F); self.Zd = np.zeros((32, 32), F)
            self.Zn[:6, :] = self.V["W1z"][32:, :].T
            self.Zd[:6, :] = self.V["W1z"][:32, :].T
            self.Zd[6:6 + self.m, :] = self.V["W1u"][:, :self.m].T
            self.W3t = np.zeros((32, 32), F)             # (-2 W3)^T, the six output adjoints in k slots 0..5 of K-half 0
            for i in range(6):
                self.W3t[:, slot_unit(0, i)] = M.W3[i, :]
        self.dt = np.asarray(cfg.time_steps, F)
        self.sdt = np.stack([self.M.sigma * F(np.sqrt(F(d))) for d in self.dt]).astype(F)       # sigma_i * sqrtf(dt_t), host float32
        disc = []
        d = F(1.0) / F(self.H)
        for _ in range(self.H + 1):
            disc.append(d)
            d = F(d * F(cfg.discount))
        self.disc = np.asarray(disc, F)
        self.invP = F(1.0) / F(self.P)

    # ---- activations: SPEC.md §3, or §10 / §10b on the modelled instructions ----
    def act(self, a):
        if self.fast:
            return self.hw.rcp(F(1.0) + self.hw.exp2(a.reshape(-1))).reshape(a.shape)
        return tanh_units(a)

    def dact(self, h):
        return fma(-h, h, h) if self.fast else fma(-h, h, F(1))

    def sig(self, x):
        if self.fast:
            return self.hw.rcp(F(1.0) + self.hw.exp2((np.asarray(x, F) * F(-1.4426950216293335)).astype(F)))
        return sigmoid(x)

    def rsq(self, a):
        return self.hw.rsq(a) if self.fast else rsqrt(a)

    # ---- §5.1 ----
    def ustep(self, u):
        M, m = self.M, self.m
        c = M.b1[:32].copy()
        for j in range(m):
            c = fma(M.W1u[:, j], u[j], c)
        Tz = t0 = t1 = t2 = F(0.0)
        for j in range(m):
            T = fma(fma(M.ct2, u[j], M.ct1), u[j], M.ct0)
            Mq = M.dir[j] * (fma(M.cm2, u[j], M.cm1) * u[j])
            Tz = F(Tz + T)
            t0 = fma(M.ry[j], T, t0)
            t1 = fma(-M.rx[j], T, t1)
            t2 = F(t2 + Mq)
        return c, F(Tz), (F(t0), F(t1), F(t2))

    # ---- §5.2: all particles at once; x [P,13], xi [P,6] ----
    def step(self, x, xi, u_terms, t, want_aux=False):
        M = self.M
        c, Tz, tau = u_terms
        dt = self.dt[t]
        p, v, q, om = x[:, 0:3], x[:, 3:6], x[:, 6:10], x[:, 10:13]
        qw, qx, qy, qz = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        xx, yy, zz = qx * qx, qy * qy, qz * qz
        xy, xz, yz, wx, wy, wz = qx * qy, qx * qz, qy * qz, qw * qx, qw * qy, qw * qz
        R = [fma(F(-2), yy + zz, F(1)), F(2) * (xy - wz), F(2) * (xz + wy),
             F(2) * (xy + wz), fma(F(-2), xx + zz, F(1)), F(2) * (yz - wx),
             F(2) * (xz - wy), F(2) * (yz + wx), fma(F(-2), xx + yy, F(1))]
        vb = [fma(R[6 + j], v[:, 2], fma(R[3 + j], v[:, 1], R[j] * v[:, 0])) for j in range(3)]
        z = vb + [om[:, 0], om[:, 1], om[:, 2]]
        a_d = np.broadcast_to(c, (x.shape[0], 32)).astype(F).copy()
        a_n = np.broadcast_to(M.b1[32:], (x.shape[0], 32)).astype(F).copy()
        if self.mlp == "f16":                        # ONE instruction per tile: products k = 0..5, ten zero products
            for p_ in range(x.shape[0]):
                zb = [f16_rtz_bits(z[k][p_]) for k in range(6)] + [0] * 10
                for r in range(32):
                    a_d[p_, r] = mfma16_dot(False, [f16_rtz_bits(M.W1z[r, k]) for k in range(6)] + [0] * 10, zb, a_d[p_, r])
                    a_n[p_, r] = mfma16_dot(False, [f16_rtz_bits(M.W1z[32 + r, k]) for k in range(6)] + [0] * 10, zb, a_n[p_, r])
        else:
            for k in range(6):
                a_d = fma(M.W1z[:32, k][None, :], z[k][:, None], a_d)
                a_n = fma(M.W1z[32:, k][None, :], z[k][:, None], a_n)
        h1d, h1n = self.act(a_d), self.act(a_n)
        a2 = np.broadcast_to(M.b2, (x.shape[0], 32)).astype(F).copy()
        if self.mlp == "f32x3":
            for p_ in range(x.shape[0]):
                a2[p_] = h2_contract(M.W2, h1d[p_], M.b2) if self.fast else x3_contract(M.W2, h1d[p_], M.b2)       # (§10c: math_mode fast's forward contraction)
        elif self.mlp == "f16":                      # TWO chained instructions (hf = 0, 1)
            for p_ in range(x.shape[0]):
                hb = [f16_rtz_bits(h1d[p_, k]) for k in range(32)]
                for i in range(32):
                    acc = M.b2[i]
                    for hf in range(2):
                        acc = mfma16_dot(False, [f16_rtz_bits(M.W2[i, slot_unit(hf, k)]) for k in range(16)], [hb[slot_unit(hf, k)] for k in range(16)], acc)
                    a2[p_, i] = acc
        else:
            for k in korder():
                a2 = fma(M.W2[:, k][None, :], h1d[:, k][:, None], a2)
        h2 = self.act(a2)
        o = [half_sums(M.W3[i], h2) + M.b3[i] for i in range(6)]
        eta = self.sig(half_sums(M.w3n, h1n) + M.b3n)
        Fb = [M.sF[0] * o[0], M.sF[1] * o[1], fma(M.sF[2], o[2], Tz)]
        acc = [fma(R[3 * i + 2], Fb[2], fma(R[3 * i + 1], Fb[1], R[3 * i] * Fb[0])) * M.inv_mass for i in range(3)]
        acc[2] = acc[2] - M.grav
        taub = [fma(M.sT[i], o[3 + i], tau[i]) for i in range(3)]
        Jom = [M.J[i] * om[:, i] for i in range(3)]
        cr = [fma(om[:, 1], Jom[2], -(om[:, 2] * Jom[1])), fma(om[:, 2], Jom[0], -(om[:, 0] * Jom[2])), fma(om[:, 0], Jom[1], -(om[:, 1] * Jom[0]))]
        dom = [(taub[i] - cr[i]) * M.iJ[i] for i in range(3)]
        dq = [F(-0.5) * fma(qz, om[:, 2], fma(qy, om[:, 1], qx * om[:, 0])),
              F(0.5) * fma(-qz, om[:, 1], fma(qy, om[:, 2], qw * om[:, 0])),
              F(0.5) * fma(-qx, om[:, 2], fma(qz, om[:, 0], qw * om[:, 1])),
              F(0.5) * fma(-qy, om[:, 0], fma(qx, om[:, 1], qw * om[:, 2]))]
        xn = np.empty_like(x)
        for i in range(3):
            xn[:, i] = fma(v[:, i], dt, p[:, i])
            xn[:, 3 + i] = fma((self.sdt[t, i] * eta), xi[:, i], fma(acc[i], dt, v[:, i]))
            xn[:, 10 + i] = fma((self.sdt[t, 3 + i] * eta), xi[:, 3 + i], fma(dom[i], dt, om[:, i]))
        qt = [fma(dq[i], dt, q[:, i]) for i in range(4)]
        n2 = fma(qt[3], qt[3], fma(qt[2], qt[2], fma(qt[1], qt[1], qt[0] * qt[0])))
        rn = self.rsq(n2)
        for i in range(4):
            xn[:, 6 + i] = qt[i] * rn
        if want_aux:
            return xn, eta, dict(R=R, h1d=h1d, h1n=h1n, h2=h2, eta=eta, Fb=Fb, Jom=Jom, rn=rn, qn=[xn[:, 6 + i] for i in range(4)])
        return xn, eta

    # ---- §5.4: vector-Jacobian product of one step, all particles at once (own writing of the hand-derived adjoint; float32, exact fma) ----
    def step_vjp(self, x, xi, t, A, L, etabar_cost):
        """x [P,13] = x_t, xi [P,6], A = auxiliaries of step(x_t), L [P,13] = adjoint of x_{t+1} (stage-cost gradient folded in),
        etabar_cost [P] = direct d(cost)/d(eta). -> (lam [P,13], gu [P,m], gT [P], gtau [P,3])"""
        M, m, V = self.M, self.m, self.V
        dt, sdt = self.dt[t], self.sdt[t]
        v, q, om = [x[:, 3 + i] for i in range(3)], [x[:, 6 + i] for i in range(4)], [x[:, 10 + i] for i in range(3)]
        Lp, Lv, Lq, Lo = [L[:, i] for i in range(3)], [L[:, 3 + i] for i in range(3)], [L[:, 6 + i] for i in range(4)], [L[:, 10 + i] for i in range(3)]
        R, qn, rn, Jom, Fb, eta = A["R"], A["qn"], A["rn"], A["Jom"], A["Fb"], A["eta"]
        qw, qx, qy, qz = q
        eb = np.asarray(etabar_cost, F)
        for i in range(3):
            eb = fma(Lv[i] * sdt[i], xi[:, i], eb)
        for i in range(3):
            eb = fma(Lo[i] * sdt[3 + i], xi[:, 3 + i], eb)
        ebraw = eb * (eta * (F(1) - eta))
        dotq = fma(qn[3], Lq[3], fma(qn[2], Lq[2], fma(qn[1], Lq[1], qn[0] * Lq[0])))
        qtb = [rn * fma(-qn[i], dotq, Lq[i]) for i in range(4)]
        dqb = [qtb[i] * dt for i in range(4)]
        taub_b = [(Lo[i] * dt) * M.iJ[i] for i in range(3)]
        crb = [-taub_b[i] for i in range(3)]
        omb = [Lo[0] + fma(Jom[1], crb[2], -(Jom[2] * crb[1])),
               Lo[1] + fma(Jom[2], crb[0], -(Jom[0] * crb[2])),
               Lo[2] + fma(Jom[0], crb[1], -(Jom[1] * crb[0]))]
        Jb = [fma(crb[1], om[2], -(crb[2] * om[1])), fma(crb[2], om[0], -(crb[0] * om[2])), fma(crb[0], om[1], -(crb[1] * om[0]))]
        omb = [fma(M.J[i], Jb[i], omb[i]) for i in range(3)]
        Fwb = [(Lv[i] * dt) * M.inv_mass for i in range(3)]
        Fbb = [fma(R[6 + j], Fwb[2], fma(R[3 + j], Fwb[1], R[j] * Fwb[0])) for j in range(3)]
        ob = [M.sF[i] * Fbb[i] for i in range(3)] + [M.sT[i] * taub_b[i] for i in range(3)]
        gT, gtau = Fbb[2], np.stack(taub_b, axis=1)
        # MLP part: abar2 = (W3^T obar) (1 - h2^2); hbar1d = W2^T abar2 in the k order of §4; abar1 = hbar1 (1 - h1^2)
        h1d, h1n, h2 = A["h1d"], A["h1n"], A["h2"]
        if self.adjmp:
            zb, gu = self.mlp_vjp_scaled(ob, ebraw, h1d, h1n, h2)
            return self._vjp_tail(x, dt, v, om, q, R, Fb, Fwb, zb, gu, qtb, dqb, omb, Lp, Lv, gT, gtau)
        hb2 = np.zeros_like(h2)
        for i in range(6):
            hb2 = fma(V["W3"][i][None, :], ob[i][:, None], hb2)
        a2b = hb2 * self.dact(h2)
        hb1 = np.zeros_like(h1d)
        if self.mlp == "f32x3":                       # W2^T abar2 as the twelve instructions of SPEC.md §9b
            for p_ in range(x.shape[0]):
                hb1[p_] = x3_contract(V["W2"].T, a2b[p_], None)
        else:
            for i in korder():
                hb1 = fma(V["W2"][i, :][None, :], a2b[:, i][:, None], hb1)
        a1d = hb1 * self.dact(h1d)
        a1n = (V["w3n"][None, :] * ebraw[:, None]) * self.dact(h1n)
        zb = []
        for k in range(6):
            P0, P1 = np.zeros(x.shape[0], F), np.zeros(x.shape[0], F)
            for tile, a1 in ((32, a1n), (0, a1d)):                       # density tile first, then the drift tile
                for r in range(16):
                    k0, k1 = (r & 3) + 8 * (r >> 2), (r & 3) + 8 * (r >> 2) + 4
                    P0 = fma(V["W1z"][tile + k0, k], a1[:, k0], P0)
                    P1 = fma(V["W1z"][tile + k1, k], a1[:, k1], P1)
            zb.append(P0 + P1)
        gu = np.stack([half_sums(V["W1u"][:, j], a1d) for j in range(m)], axis=1)
        return self._vjp_tail(x, dt, v, om, q, R, Fb, Fwb, zb, gu, qtb, dqb, omb, Lp, Lv, gT, gtau)

    def mlp_vjp_scaled(self, ob, ebraw, h1d, h1n, h2):
        """SPEC.md §10e: per particle, the seven output adjoints scaled by sigma = -2 * 2^(eoff - e) (e: the binary exponent of the largest of them, as
        frexp returns it, at least -100; 0 for zero / non-finite), the forward output weights (-2 W3, -2 w3n), and the three contractions — (4 W2)^T abar2,
        Zn abar1n, Zd abar1d — each from two binary16 limbs of either operand (eight instructions, C = 0); rows 0..7 of the two narrow results are added;
        everything is unscaled by 2^(e - eoff) at the end."""
        M, m, V = self.M, self.m, self.V
        P = ebraw.shape[0]
        zb, gu = np.zeros((P, 6), F), np.zeros((P, m), F)
        for p_ in range(P):
            vals = [abs(F(ebraw[p_]))] + [abs(F(ob[i][p_])) for i in range(6)]
            mx = F(0)
            for v_ in vals:                        # a maximum that ignores NaNs (fmaxf / v_max_f32)
                if not np.isnan(v_) and (np.isnan(mx) or v_ > mx):
                    mx = v_
            e = int(np.frexp(mx)[1]) if (mx > 0 and np.isfinite(mx)) else 0
            e = max(e, -100)
            sig = F(np.ldexp(F(-2), self.adj_eoff - e))
            inv = F(np.ldexp(F(1), e - self.adj_eoff))
            with np.errstate(all="ignore"):
                obs = [F(F(ob[i][p_]) * sig) for i in range(6)]
                ebs = F(F(ebraw[p_]) * sig)
                ov = np.zeros(32, F)
                for i in range(6):
                    ov[slot_unit(0, i)] = obs[i]             # k slot i of K-half 0
                hb = h2_contract(self.W3t, ov, None)
                a2 = (hb * self.dact(h2[p_])).astype(F)
                hb1 = h2_contract(V["W2"].T, a2, None)
                a1d = (hb1 * self.dact(h1d[p_])).astype(F)
                a1n = ((M.w3n * ebs).astype(F) * self.dact(h1n[p_])).astype(F)
                zn = h2_contract(self.Zn, a1n, None)
                zd = h2_contract(self.Zd, a1d, None)
                zd[:8] = (zd[:8] + zn[:8]).astype(F)
                zb[p_] = (zd[:6] * inv).astype(F)
                gu[p_] = (zd[6:6 + m] * inv).astype(F)
        return [zb[:, k] for k in range(6)], gu

    def _vjp_tail(self, x, dt, v, om, q, R, Fb, Fwb, zb, gu, qtb, dqb, omb, Lp, Lv, gT, gtau):
        qw, qx, qy, qz = q
        omb = [omb[i] + zb[3 + i] for i in range(3)]
        vbar = [fma(Lp[i], dt, Lv[i]) + fma(R[3 * i + 2], zb[2], fma(R[3 * i + 1], zb[1], R[3 * i] * zb[0])) for i in range(3)]
        Rb = [fma(v[i], zb[j], Fwb[i] * Fb[j]) for i in range(3) for j in range(3)]
        qb = [fma(F(0.5), fma(dqb[3], om[2], fma(dqb[2], om[1], dqb[1] * om[0])), qtb[0]),
              fma(F(0.5), fma(dqb[3], om[1], fma(-dqb[2], om[2], -(dqb[0] * om[0]))), qtb[1]),
              fma(F(0.5), fma(-dqb[3], om[0], fma(dqb[1], om[2], -(dqb[0] * om[1]))), qtb[2]),
              fma(F(0.5), fma(dqb[2], om[0], fma(-dqb[1], om[1], -(dqb[0] * om[2]))), qtb[3])]
        omb = [fma(F(0.5), fma(-dqb[3], qy, fma(dqb[2], qz, fma(dqb[1], qw, -(dqb[0] * qx)))), omb[0]),
               fma(F(0.5), fma(dqb[3], qx, fma(dqb[2], qw, fma(-dqb[1], qz, -(dqb[0] * qy)))), omb[1]),
               fma(F(0.5), fma(dqb[3], qw, fma(-dqb[2], qx, fma(dqb[1], qy, -(dqb[0] * qz)))), omb[2])]
        s01, d10 = Rb[1] + Rb[3], Rb[3] - Rb[1]
        s02, d02 = Rb[2] + Rb[6], Rb[2] - Rb[6]
        s12, d21 = Rb[5] + Rb[7], Rb[7] - Rb[5]
        qb = [fma(F(2), fma(qx, d21, fma(qy, d02, qz * d10)), qb[0]),
              fma(F(2), fma(qw, d21, fma(qz, s02, qy * s01)), fma(F(-4) * qx, Rb[4] + Rb[8], qb[1])),
              fma(F(2), fma(qz, s12, fma(qw, d02, qx * s01)), fma(F(-4) * qy, Rb[0] + Rb[8], qb[2])),
              fma(F(2), fma(qy, s12, fma(qx, s02, qw * d10)), fma(F(-4) * qz, Rb[0] + Rb[4], qb[3]))]
        lam = np.stack(Lp + vbar + qb + omb, axis=1).astype(F)
        return lam, gu, gT, gtau

    # gradient of the stage cost of §5.3 at x [P,13]
    def stage_cost_grad(self, x, xr):
        C = self.cfg
        gx = np.zeros_like(x)
        for w_, off in ((C.perr, 0), (C.verr, 3), (C.werr, 10)):
            for i in range(3):
                e = x[:, off + i] - xr[off + i]
                gx[:, off + i] = F(2) * (F(w_[i]) * e)
        qw, qx, qy, qz = x[:, 6], x[:, 7], x[:, 8], x[:, 9]
        rw, rx, ry, rz = xr[6], xr[7], xr[8], xr[9]
        ex = fma(rz, qy, fma(-ry, qz, fma(-rx, qw, rw * qx)))
        ey = fma(-rz, qx, fma(-ry, qw, fma(rx, qz, rw * qy)))
        ez = fma(-rz, qw, fma(ry, qx, fma(-rx, qy, rw * qz)))
        a, b, c = F(2) * (F(C.qerr[0]) * ex), F(2) * (F(C.qerr[1]) * ey), F(2) * (F(C.qerr[2]) * ez)
        gx[:, 6] = fma(-rz, c, fma(-ry, b, -rx * a))
        gx[:, 7] = fma(ry, c, fma(-rz, b, rw * a))
        gx[:, 8] = fma(-rx, c, fma(rw, b, rz * a))
        gx[:, 9] = fma(rw, c, fma(rx, b, -ry * a))
        if C.state_id:
            for k, i in enumerate(C.state_id):
                w = F(F(C.state_penalty[k]) * F(C.constr_pen))
                with np.errstate(invalid="ignore"):
                    hi = x[:, i] - F(C.state_bound[k][1])
                    hi = np.where(hi < 0, F(0), hi).astype(F)
                    lo = F(C.state_bound[k][0]) - x[:, i]
                    lo = np.where(lo < 0, F(0), lo).astype(F)
                gx[:, i] = fma(F(2) * w, hi - lo, gx[:, i])
        return gx

    # ---- §5.3 ----
    def stage_cost(self, x, xr):
        C = self.cfg
        l = np.zeros(x.shape[0], F)
        for w_, off in ((C.perr, 0), (C.verr, 3), (C.werr, 10)):
            for i in range(3):
                e = x[:, off + i] - xr[off + i]
                l = fma(F(w_[i]) * e, e, l)
        qw, qx, qy, qz = x[:, 6], x[:, 7], x[:, 8], x[:, 9]
        rw, rx, ry, rz = xr[6], xr[7], xr[8], xr[9]
        ex = fma(rz, qy, fma(-ry, qz, fma(-rx, qw, rw * qx)))
        ey = fma(-rz, qx, fma(-ry, qw, fma(rx, qz, rw * qy)))
        ez = fma(-rz, qw, fma(ry, qx, fma(-rx, qy, rw * qz)))
        for w_, e in ((C.qerr[0], ex), (C.qerr[1], ey), (C.qerr[2], ez)):
            l = fma(F(w_) * e, e, l)
        if C.state_id:                                            # state_constr, penalty form (SPEC.md §5.3)
            for k, i in enumerate(C.state_id):
                w = F(F(C.state_penalty[k]) * F(C.constr_pen))
                with np.errstate(invalid="ignore"):
                    hi = x[:, i] - F(C.state_bound[k][1])
                    hi = np.where(hi < 0, F(0), hi).astype(F)
                    lo = F(C.state_bound[k][0]) - x[:, i]
                    lo = np.where(lo < 0, F(0), lo).astype(F)
                l = fma(w * hi, hi, l)
                l = fma(w * lo, lo, l)
        return l

    def particle_sum(self, vals):
        """SPEC.md §6.1; vals [P, ...] -> [...]"""
        P = vals.shape[0]
        G = (P + 31) // 32
        S = [np.zeros(vals.shape[1:], F) for _ in range(4)]
        for g in range(G):
            v = np.zeros((32,) + vals.shape[1:], F)
            n = min(32, P - 32 * g)
            v[:n] = vals[32 * g:32 * g + n]
            for s in (16, 8, 4, 2, 1):
                v = v + v[np.arange(32) ^ s]
            S[g % 4] = S[g % 4] + v[0]
        return ((S[0] + S[1]) + S[2]) + S[3]

    @staticmethod
    def dot256(a, b=None):
        """SPEC.md §6.2 over flat arrays."""
        a = np.asarray(a, F).reshape(-1)
        b = np.ones_like(a) if b is None else np.asarray(b, F).reshape(-1)
        N = a.size
        v = np.zeros(256, F)
        for e0 in range(0, N, 256):
            n = min(256, N - e0)
            v[:n] = fma(a[e0:e0 + n], b[e0:e0 + n], v[:n])
        w = []
        for k in range(4):
            x = v[64 * k:64 * k + 64]
            for s in (32, 16, 8, 4, 2, 1):
                x = x + x[np.arange(64) ^ s]
            w.append(x[0])
        return F(((w[0] + w[1]) + w[2]) + w[3])

    def control_cost(self, u):
        C, H, m = self.cfg, self.H, self.m
        u = np.asarray(u, F)
        el = np.zeros((H, m), F)
        for t in range(H):
            for j in range(m):
                du = F(u[t, j] - F(C.uref[j]))
                c = F(F(F(C.uerr) * du) * du)
                if t >= 1:
                    ds = F(u[t, j] - u[t - 1, j])
                    c = fma(F(F(C.u_slew_coeff) * ds), ds, c)
                    if C.u_slew_constr is not None:
                        lo_b, hi_b = C.u_slew_constr[j]
                        hi = max(F(0.0), F(ds - F(hi_b)))
                        lo = max(F(0.0), F(F(lo_b) - ds))
                        c = fma(F(F(C.u_slew_constr_coeff) * hi), hi, c)
                        c = fma(F(F(C.u_slew_constr_coeff) * lo), lo, c)
                el[t, j] = F(self.disc[t] * F(c))
        return self.dot256(el)

    def control_cost_grad(self, u):
        """analytic gradient of control_cost (SPEC.md §5.5)"""
        C, H, m = self.cfg, self.H, self.m
        u = np.asarray(u, F)
        dw = np.zeros((H, m), F)
        for t in range(1, H):
            for j in range(m):
                ds = F(u[t, j] - u[t - 1, j])
                d = F(F(F(2) * F(C.u_slew_coeff)) * ds)
                if C.u_slew_constr is not None:
                    lo_b, hi_b = C.u_slew_constr[j]
                    hi = max(F(0.0), F(ds - F(hi_b)))
                    lo = max(F(0.0), F(F(lo_b) - ds))
                    d = fma(F(F(2) * F(C.u_slew_constr_coeff)), F(hi - lo), d)
                dw[t, j] = d
        g = np.zeros((H, m), F)
        for t in range(H):
            for j in range(m):
                du = F(u[t, j] - F(C.uref[j]))
                gg = F(self.disc[t] * fma(F(F(2) * F(C.uerr)), du, dw[t, j]))
                if t + 1 < H:
                    gg = fma(-self.disc[t + 1], dw[t + 1, j], gg)
                g[t, j] = gg
        return g

    def cost_grad(self, x0, u, xref, noise):
        """-> (expected cost, gradient [H,m]) by the adjoint sweep of SPEC.md §5.4-§5.5, all particles at once"""
        M, H, P, m = self.M, self.H, self.P, self.m
        x0, u, xref, noise = (np.asarray(a, F) for a in (x0, u, xref, noise))
        res = F(self.cfg.res_mult)
        ut = [self.ustep(u[t]) for t in range(H)]
        x = np.broadcast_to(x0, (P, 13)).astype(F).copy()
        traj = [x]
        J = np.zeros(P, F)
        for t in range(H):
            xn, eta = self.step(x, noise[:, t], ut[t], t)
            l = fma(res * eta, eta, self.stage_cost(xn, xref[t + 1]))
            J = fma(self.disc[t], l, J)
            x = xn
            traj.append(x)
        lam = np.zeros((P, 13), F)
        g = np.zeros((H, m), F)
        gcu = self.control_cost_grad(u)
        for t in range(H - 1, -1, -1):
            xt, x1 = traj[t], traj[t + 1]
            _, _, A = self.step(xt, noise[:, t], ut[t], t, want_aux=True)
            gx = self.stage_cost_grad(x1, xref[t + 1])
            L = fma(self.disc[t], gx, lam)
            ebc = self.disc[t] * ((F(2) * res) * A["eta"])
            lam, gu, gT, gtau = self.step_vjp(xt, noise[:, t], t, A, L, ebc)
            S_gu, S_T, S_tau = self.particle_sum(gu), self.particle_sum(gT), self.particle_sum(gtau)
            for j in range(m):
                uj = u[t, j]
                dT = fma(F(2) * M.ct2, uj, M.ct1)
                dM = F(M.dir[j] * fma(F(2) * M.cm2, uj, M.cm1))
                a = S_gu[j]
                a = fma(S_T, dT, a)
                a = fma(S_tau[0], F(M.ry[j] * dT), a)
                a = fma(S_tau[1], F(-(M.rx[j] * dT)), a)
                a = fma(S_tau[2], dM, a)
                g[t, j] = fma(a, self.invP, gcu[t, j])
        tot = self.particle_sum(J)
        return F(fma(tot, self.invP, self.control_cost(u))), g

    def solve_own(self, x0, xref, noise, u_init, stepsize_in):
        """SPEC.md §8 with THIS file's cost and gradient: nothing of the C oracle takes part."""
        return self.solve(lambda uu: self.rollout(x0, uu, xref, noise)[0], lambda yy: self.cost_grad(x0, yy, xref, noise), u_init, stepsize_in)

    def rollout(self, x0, u, xref, noise):
        """-> (expected cost, traj [P,H+1,13], mean trajectory [H+1,13])"""
        H, P = self.H, self.P
        x0, u, xref, noise = (np.asarray(a, F) for a in (x0, u, xref, noise))
        x = np.broadcast_to(x0, (P, 13)).astype(F).copy()
        traj = np.empty((P, H + 1, 13), F)
        traj[:, 0] = x
        J = np.zeros(P, F)
        for t in range(H):
            xn, eta = self.step(x, noise[:, t], self.ustep(u[t]), t)
            l = self.stage_cost(xn, xref[t + 1])
            l = fma(F(self.cfg.res_mult) * eta, eta, l)
            J = fma(self.disc[t], l, J)
            x = xn
            traj[:, t + 1] = x
        tot = self.particle_sum(J)
        xmean = (self.particle_sum(traj) * self.invP).astype(F)
        return F(fma(tot, self.invP, self.control_cost(u))), traj, xmean

    # ---- §8: the optimiser, with cost / gradient supplied by callables (the tests pass the C oracle's) ----
    def solve(self, cost_fn, grad_fn, u_init, stepsize_in):
        C, H, m = self.cfg, self.H, self.m
        lo, hi = np.asarray([b[0] for b in C.input_bound], F), np.asarray([b[1] for b in C.input_bound], F)
        if not C.enforce_ubound:
            lo, hi = np.full(m, -np.inf, F), np.full(m, np.inf, F)
        proj = lambda v: np.stack([clamp(v[:, j], lo[j], hi[j]) for j in range(m)], axis=1)
        beta = [F(C.beta_init)] + [F(F(i + 1) / F(i + 4)) for i in range(1, C.max_iter + 2)]
        if C.moment_scale is not None:
            beta = [beta[0]] + [F(F(C.moment_scale) * b) for b in beta[1:]]
        xk = proj(np.asarray(u_init, F))
        yk = xk.copy()
        c_init = c_x = F(cost_fn(xk))
        s = F(stepsize_in)
        gsq, sum_ls, sum_s = F(0), F(0), F(0)
        kr = noimp = nit = nls_tot = 0
        plain = True
        for k in range(C.max_iter):
            c_y, g = grad_fn(yk)
            c_y, g = F(c_y), np.asarray(g, F)
            gsq = self.dot256(g, g)
            if not (gsq < np.inf):
                break
            c_n, nls = F(0), 0
            if C.ls_maxls > 0:
                if k > 0 and C.ls_reset_option == "increase":
                    s = F(s * F(C.ls_increase_factor))
                if s > F(C.ls_max_stepsize):
                    s = F(C.ls_max_stepsize)
                for jl in range(C.ls_maxls):
                    xn = proj(fma(-s, g, yk))
                    d1 = xn - yk
                    c_n = F(cost_fn(xn))
                    gd = self.dot256(g, d1)
                    nls = jl + 1
                    if c_n <= fma(F(C.ls_coef), gd, c_y):
                        break
                    if jl < C.ls_maxls - 1:
                        s = F(s * F(C.ls_decrease_factor))
            else:
                s = F(C.stepsize)
                xn = proj(fma(-s, g, yk))
                c_n, nls = F(cost_fn(xn)), 1
            sum_ls, sum_s, nit, nls_tot = F(sum_ls + F(nls)), F(sum_s + s), k + 1, nls_tot + nls
            stop = abs(F(c_n - c_x)) <= fma(F(C.rtol), abs(c_x), F(C.atol))
            if c_n < c_x:
                if self.dot256(yk - xn, xn - xk) > 0:
                    yk, kr, plain = xn.copy(), 0, True
                else:
                    yk, kr, plain = proj(fma(beta[kr], xn - xk, xn)), kr + 1, False
                xk, c_x, noimp = xn.copy(), c_n, 0
            else:
                if not plain:
                    stop = False
                yk, kr, plain, noimp = xk.copy(), 0, True, noimp + 1
            if noimp >= C.max_no_improvement_iter:
                stop = True
            if stop:
                break
        fn = F(nit)
        info = np.array([sum_ls / fn if nit else 0, s, fn, gsq, sum_s / fn if nit else 0, c_init, c_x, nls_tot], F)
        return xk, info


# ------------------------------------------------------------------------------------------------------------------------------
# Third writing of the model, in torch float64 with libm activations, for reverse-mode differentiation of the expected cost
# ------------------------------------------------------------------------------------------------------------------------------
def torch_cost(cfg, model, x0, u, xref, noise):
    """Expected cost J(u) in float64 (true tanh / sigmoid / 1/sqrt) as a differentiable torch scalar; u: torch [H, m] float64."""
    import torch
    M = Model(model.to_blob() if hasattr(model, "to_blob") else bytes(model))
    T = lambda a: torch.as_tensor(np.asarray(a, np.float64))
    H, P, m = cfg.horizon, cfg.num_particles, cfg.num_motors
    dt = np.asarray(cfg.time_steps, F).astype(np.float64)
    sdt = np.stack([M.sigma * F(np.sqrt(F(d))) for d in np.asarray(cfg.time_steps, F)]).astype(np.float64)
    disc = [float(F(1.0) / F(H))]
    for _ in range(H):
        disc.append(float(F(F(disc[-1]) * F(cfg.discount))))
    W1z, W1u, W2, W3 = T(M.W1z), T(M.W1u[:, :m]), T(M.W2), T(M.W3[:6])
    b1, b2, b3, w3n = T(M.b1), T(M.b2), T(M.b3[:6]), T(M.w3n)
    x = T(x0).repeat(P, 1)
    xr, nz = T(xref), T(noise)
    rx, ry, dr = T(M.rx[:m]), T(M.ry[:m]), T(M.dir[:m])
    J = torch.zeros(P, dtype=torch.float64)
    for t in range(H):
        ut = u[t]
        Tj = (float(M.ct2) * ut + float(M.ct1)) * ut + float(M.ct0)
        Mq = dr * ((float(M.cm2) * ut + float(M.cm1)) * ut)
        Tz, tau = Tj.sum(), torch.stack([(ry * Tj).sum(), (-rx * Tj).sum(), Mq.sum()])
        p, v, q, om = x[:, 0:3], x[:, 3:6], x[:, 6:10], x[:, 10:13]
        qw, qx, qy, qz = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        R = torch.stack([torch.stack([1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy)], -1),
                         torch.stack([2 * (qx * qy + qw * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qw * qx)], -1),
                         torch.stack([2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), 1 - 2 * (qx * qx + qy * qy)], -1)], 1)   # [P,3,3]
        vb = torch.einsum("pij,pi->pj", R, v)
        z = torch.cat([vb, om], 1)
        h1d = torch.tanh(z @ W1z[:32].T + (W1u @ ut + b1[:32]))
        h1n = torch.tanh(z @ W1z[32:].T + b1[32:])
        h2 = torch.tanh(h1d @ W2.T + b2)
        o = h2 @ W3.T + b3
        eta = torch.sigmoid(h1n @ w3n + float(M.b3n))
        Fb = torch.stack([float(M.sF[0]) * o[:, 0], float(M.sF[1]) * o[:, 1], float(M.sF[2]) * o[:, 2] + Tz], 1)
        acc = torch.einsum("pij,pj->pi", R, Fb) * float(M.inv_mass)
        acc = acc - torch.tensor([0.0, 0.0, float(M.grav)], dtype=torch.float64)
        taub = T(M.sT) * o[:, 3:6] + tau
        Jm = T(M.J)
        dom = (taub - torch.cross(om, Jm * om, dim=1)) * T(M.iJ)
        dq = 0.5 * torch.stack([-(qx * om[:, 0] + qy * om[:, 1] + qz * om[:, 2]),
                                qw * om[:, 0] + qy * om[:, 2] - qz * om[:, 1],
                                qw * om[:, 1] + qz * om[:, 0] - qx * om[:, 2],
                                qw * om[:, 2] + qx * om[:, 1] - qy * om[:, 0]], 1)
        s = T(sdt[t])
        pn = p + v * dt[t]
        vn = v + acc * dt[t] + (s[:3] * eta[:, None]) * nz[:, t, :3]
        on = om + dom * dt[t] + (s[3:] * eta[:, None]) * nz[:, t, 3:]
        qt = q + dq * dt[t]
        qn = qt / qt.norm(dim=1, keepdim=True)
        x = torch.cat([pn, vn, qn, on], 1)
        r = xr[t + 1]
        l = (T(cfg.perr) * (pn - r[0:3]) ** 2).sum(1) + (T(cfg.verr) * (vn - r[3:6]) ** 2).sum(1) + (T(cfg.werr) * (on - r[10:13]) ** 2).sum(1)
        rw, rxq, ryq, rzq = r[6], r[7], r[8], r[9]
        ex = rw * qn[:, 1] - rxq * qn[:, 0] - ryq * qn[:, 3] + rzq * qn[:, 2]
        ey = rw * qn[:, 2] + rxq * qn[:, 3] - ryq * qn[:, 0] - rzq * qn[:, 1]
        ez = rw * qn[:, 3] - rxq * qn[:, 2] + ryq * qn[:, 1] - rzq * qn[:, 0]
        l = l + float(cfg.qerr[0]) * ex ** 2 + float(cfg.qerr[1]) * ey ** 2 + float(cfg.qerr[2]) * ez ** 2 + float(cfg.res_mult) * eta ** 2
        if cfg.state_id:
            for k, i in enumerate(cfg.state_id):
                w = float(np.float32(cfg.state_penalty[k]) * np.float32(cfg.constr_pen))
                l = l + w * (torch.clamp(x[:, i] - float(np.float32(cfg.state_bound[k][1])), min=0) ** 2
                             + torch.clamp(float(np.float32(cfg.state_bound[k][0])) - x[:, i], min=0) ** 2)
        J = J + disc[t] * l
    du = u - T(np.asarray(cfg.uref[:m], np.float64))
    cu = float(cfg.uerr) * du ** 2
    ds = u[1:] - u[:-1]
    sl = float(cfg.u_slew_coeff) * ds ** 2
    if cfg.u_slew_constr is not None:
        lo_b, hi_b = T([b[0] for b in cfg.u_slew_constr]), T([b[1] for b in cfg.u_slew_constr])
        sl = sl + float(cfg.u_slew_constr_coeff) * (torch.clamp(ds - hi_b, min=0) ** 2 + torch.clamp(lo_b - ds, min=0) ** 2)
    cu = torch.cat([cu[:1], cu[1:] + sl], 0)
    return J.mean() + (T(disc[:H])[:, None] * cu).sum()
