"""Exact-arithmetic helpers for the MFMA numerics study: decode tiles, candidate accumulation models (python ints)."""
import numpy as np

MAGIC = 0x4D464D41


def load_cases(path):
    raw = np.fromfile(path, np.uint8)
    hdr = raw[:16].view(np.int32)
    assert hdr[0] == MAGIC
    T, dt = int(hdr[1]), int(hdr[2])
    body = raw[16:].reshape(T, 1024 + 1024 + 4096)
    A = body[:, :1024].copy().view(np.uint16).reshape(T, 32, 16)
    B = body[:, 1024:2048].copy().view(np.uint16).reshape(T, 16, 32)
    C = body[:, 2048:].copy().view(np.uint32).reshape(T, 32, 32)
    return dt, A, B, C


def load_out(path, T):
    return np.fromfile(path, np.uint32).reshape(T, 32, 32)


def dec16(bits, dt):
    """uint16 patterns -> (signed integer mantissa, exponent of its lsb): value = m * 2^e. Sub-normals included."""
    bits = bits.astype(np.int64)
    if dt == 0:
        s = bits >> 15; e = (bits >> 10) & 31; f = bits & 1023
        m = np.where(e == 0, f, f | 1024); ex = np.where(e == 0, -24, e - 25)
    else:
        s = bits >> 15; e = (bits >> 7) & 255; f = bits & 127
        m = np.where(e == 0, f, f | 128); ex = np.where(e == 0, -133, e - 134)
    return np.where(s == 1, -m, m), ex


def dec32(bits):
    bits = bits.astype(np.int64)
    s = bits >> 31; e = (bits >> 23) & 255; f = bits & 0x7FFFFF
    m = np.where(e == 0, f, f | 0x800000); ex = np.where(e == 0, -149, e - 150)
    return np.where(s == 1, -m, m), ex


def f32_round(m, e, mode="rne"):
    """exact value m * 2^e (python ints) -> f32 bit pattern (no overflow handling beyond inf)"""
    if m == 0:
        return 0
    s = 1 if m < 0 else 0
    a = -m if m < 0 else m
    bl = a.bit_length()
    E = e + bl - 1                      # exponent of the leading bit
    lsb = max(E - 23, -149)             # exponent of the result's lsb (sub-normals: fixed at -149)
    sh = lsb - e
    if sh <= 0:
        q = a << (-sh)
    else:
        q = a >> sh
        rem = a & ((1 << sh) - 1)
        half = 1 << (sh - 1)
        if mode == "rne":
            if rem > half or (rem == half and (q & 1)):
                q += 1
        elif mode == "rz":
            pass
        elif mode == "rna":           # ties away
            if rem >= half:
                q += 1
        elif mode == "rdn":           # toward -inf
            if rem and s:
                q += 1
        elif mode == "rup":
            if rem and not s:
                q += 1
        else:
            raise ValueError(mode)
    if q == 0:
        return s << 31
    if q >= (1 << 24):                 # rounding carried out
        q >>= 1; lsb += 1
    if q < (1 << 23):                  # sub-normal
        assert lsb == -149
        return (s << 31) | q
    ebits = lsb + 150
    if ebits >= 255:
        return (s << 31) | 0x7F800000
    return (s << 31) | (ebits << 23) | (q & 0x7FFFFF)


def f32_val(bits):
    """f32 pattern -> (m, e) python ints"""
    s = bits >> 31; e = (bits >> 23) & 255; f = bits & 0x7FFFFF
    m = f if e == 0 else (f | 0x800000)
    ex = -149 if e == 0 else e - 150
    return (-m if s else m), ex


def msb_exp(m, e):
    return e + abs(m).bit_length() - 1


def fixed_sum(terms, W, tmode, ref="max"):
    """Align (m, e) terms to the largest leading-bit exponent, keep W bits below it, add exactly.
    tmode: 'tz' truncate magnitudes, 'floor' two's-complement truncation, 'rne' round each addend to nearest even.
    Returns (m, e) exact value of the fixed-point sum."""
    nz = [(m, e) for (m, e) in terms if m != 0]
    if not nz:
        return 0, 0
    emax = max(msb_exp(m, e) for m, e in nz)
    lsb = emax - W
    acc = 0
    for m, e in nz:
        sh = lsb - e
        if sh <= 0:
            acc += m << (-sh)
        else:
            if tmode == "tz":
                q = (abs(m) >> sh); q = -q if m < 0 else q
            elif tmode == "floor":
                q = m >> sh
            elif tmode == "rne":
                a = abs(m); q = a >> sh; rem = a & ((1 << sh) - 1); half = 1 << (sh - 1)
                if rem > half or (rem == half and (q & 1)):
                    q += 1
                q = -q if m < 0 else q
            else:
                raise ValueError(tmode)
            acc += q
    return acc, lsb


def model_chunks(prods, c, chunks, W, tmode, inter="rne", final="rne", c_first=True):
    """prods: list of 16 (m, e); c: (m, e). chunks: list of lists of k. Each chunk is one fused fixed-point addition of the running
    value and the chunk's products (aligned to the largest, W bits kept), rounded to f32 (inter) before the next chunk."""
    acc = c
    for ci, ch in enumerate(chunks):
        terms = [prods[k] for k in ch]
        if c_first or ci > 0:
            terms = [acc] + terms
        m, e = fixed_sum(terms, W, tmode)
        if not c_first and ci == 0:
            # products first, then C joins exactly
            m2, e2 = acc
            lo = min(e, e2)
            m = (m << (e - lo)) + (m2 << (e2 - lo)); e = lo
        last = ci == len(chunks) - 1
        mode = final if last else inter
        if mode is None:
            acc = (m, e)
        else:
            acc = f32_val(f32_round(m, e, mode))
    return f32_round(acc[0], acc[1], "rne")
