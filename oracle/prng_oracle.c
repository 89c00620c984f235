/*
 * prng_oracle.c — CPU restatement of the key handling and key-derived noise (SPEC.md §7). TEST INFRASTRUCTURE ONLY
 * (same rule as sde_mpc_oracle.c: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load liborc.so).
 *
 * What the reference fixes at this boundary: the MPC node creates `jax.random.PRNGKey(seed)` and splits it three ways
 * (sde4mbrl_px4/mpc_controller/sde_control.py:338-341), passes a key into every `m_reset` / `m_mpc` call and takes a key back
 * (sde_control.py:345-350,400-416,698,706,717). The key algebra is JAX's default PRNG, threefry2x32 (Salmon, Moraes, Dror,
 * Shaw, "Parallel random numbers: as easy as 1, 2, 3", SC'11; JAX's `jax._src.prng`, legacy non-partitionable layout).
 * JAX is an external dependency (absent from /root/reference, version unpinned); the PUBLISHED algorithm is restated here and
 * pinned by public known-answer values (tests/test_prng_cpu.py):
 *   - the three Random123 threefry2x32 vectors (also used by JAX's own test-suite),
 *   - `split(PRNGKey(0))` = [[4146024105, 967050713], [2718843009, 1272950319]] and its second level, as printed in the JAX
 *     documentation,
 *   - `normal(PRNGKey(0), (1,))` = -0.20584226 (and two more), same source, matched to 1 ulp: JAX evaluates
 *     sqrt(2)*erfinv(u) with Giles' single-precision polynomial, restated below; its log1p/sqrt are XLA's, ours are the
 *     bit-reproducible forms of SPEC.md §7.2, hence "1 ulp" and not "bit for bit".
 * How the external package consumes keys INSIDE m_mpc is not in the reference; SPEC.md §7.3 fixes this build's choice.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

static inline float as_f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t as_u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline uint32_t rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

/* SPEC.md §7.1: threefry2x32, 20 rounds */
void orc_threefry2x32(const uint32_t key[2], uint32_t x0, uint32_t x1, uint32_t out[2]) {
    static const int R[2][4] = {{13, 15, 26, 6}, {17, 29, 16, 24}};
    const uint32_t ks[3] = {key[0], key[1], key[0] ^ key[1] ^ 0x1BD11BDAu};
    x0 += ks[0]; x1 += ks[1];
    for (int i = 0; i < 5; ++i) {
        for (int j = 0; j < 4; ++j) { x0 += x1; x1 = rotl(x1, R[i & 1][j]); x1 ^= x0; }
        x0 += ks[(i + 1) % 3];
        x1 += ks[(i + 2) % 3] + (uint32_t)(i + 1);
    }
    out[0] = x0; out[1] = x1;
}

/* n words from one key: counters 0..n-1 (one zero appended when n is odd), first half -> x0, second half -> x1,
 * outputs concatenated (y0 then y1) and truncated to n. */
void orc_random_bits(const uint32_t key[2], size_t n, uint32_t* out) {
    const size_t half = (n + 1) / 2;
    for (size_t k = 0; k < half; ++k) {
        const size_t c1 = k + half;
        uint32_t y[2];
        orc_threefry2x32(key, (uint32_t)k, c1 < n ? (uint32_t)c1 : 0u, y);
        out[k] = y[0];
        if (c1 < n) out[c1] = y[1];
    }
}

void orc_split(const uint32_t key[2], int num, uint32_t* out /*[num][2]*/) { orc_random_bits(key, (size_t)2 * num, out); }

/* SPEC.md §7.2: natural logarithm of a positive normal float (Cephes logf scheme, fma Horner) */
float orc_log(float t) {
    const uint32_t b = as_u(t);
    int e = (int)((b >> 23) & 255u) - 126;
    float m = as_f((b & 0x007FFFFFu) | 0x3F000000u);   /* [0.5, 1) */
    float x;
    if (m < 0.707106781186547524f) { e -= 1; x = (m + m) - 1.0f; } else x = m - 1.0f;
    const float z = x * x;
    float y = 7.0376836292E-2f;
    y = fmaf(y, x, -1.1514610310E-1f);
    y = fmaf(y, x, 1.1676998740E-1f);
    y = fmaf(y, x, -1.2420140846E-1f);
    y = fmaf(y, x, 1.4249322787E-1f);
    y = fmaf(y, x, -1.6668057665E-1f);
    y = fmaf(y, x, 2.0000714765E-1f);
    y = fmaf(y, x, -2.4999993993E-1f);
    y = fmaf(y, x, 3.3333331174E-1f);
    y = (y * x) * z;
    const float fe = (float)e;
    y = fmaf(-2.12194440e-4f, fe, y);
    y = fmaf(-0.5f, z, y);
    float r = x + y;
    r = fmaf(0.693359375f, fe, r);
    return r;
}

static float rsqrt_spec(float a) {   /* SPEC.md §3.2 */
    float y = as_f(0x5F3759DFu - (as_u(a) >> 1));
    const float h = 0.5f * a;
    for (int i = 0; i < 3; ++i) { float t = y * y; t = fmaf(-h, t, 1.5f); y = y * t; }
    return y;
}
/* sqrt for a >= 5: s = a*rsqrt(a), one correction step */
float orc_sqrt(float a) {
    const float y = rsqrt_spec(a);
    float s = a * y;
    const float r = fmaf(-s, s, a);
    return fmaf(r, 0.5f * y, s);
}

/* Giles, "Approximating the erfinv function" (GPU Computing Gems, 2011), single precision */
float orc_erfinv(float u) {
    float w = -orc_log(fmaf(-u, u, 1.0f));
    float p;
    if (w < 5.0f) {
        w = w - 2.5f;
        p = 2.81022636e-08f;
        p = fmaf(p, w, 3.43273939e-07f);
        p = fmaf(p, w, -3.5233877e-06f);
        p = fmaf(p, w, -4.39150654e-06f);
        p = fmaf(p, w, 0.00021858087f);
        p = fmaf(p, w, -0.00125372503f);
        p = fmaf(p, w, -0.00417768164f);
        p = fmaf(p, w, 0.246640727f);
        p = fmaf(p, w, 1.50140941f);
    } else {
        w = orc_sqrt(w) - 3.0f;
        p = -0.000200214257f;
        p = fmaf(p, w, 0.000100950558f);
        p = fmaf(p, w, 0.00134934322f);
        p = fmaf(p, w, -0.00367342844f);
        p = fmaf(p, w, 0.00573950773f);
        p = fmaf(p, w, -0.0076224613f);
        p = fmaf(p, w, 0.00943887047f);
        p = fmaf(p, w, 1.00167406f);
        p = fmaf(p, w, 2.83297682f);
    }
    return p * u;
}

/* 32 random bits -> N(0,1): mantissa trick to [0,1), affine map to (-1,1), sqrt(2)*erfinv */
float orc_bits_to_normal(uint32_t bits) {
    const float lo = -0.99999994f;                       /* nextafter(-1, 0) */
    const float f = as_f((bits >> 9) | 0x3F800000u) - 1.0f;
    float u = fmaf(f, 2.0f, lo);
    if (!(u > lo)) u = lo;
    return 1.41421354f * orc_erfinv(u);
}

void orc_normal(const uint32_t key[2], size_t n, float* out) {
    const size_t half = (n + 1) / 2;
    for (size_t k = 0; k < half; ++k) {
        const size_t c1 = k + half;
        uint32_t y[2];
        orc_threefry2x32(key, (uint32_t)k, c1 < n ? (uint32_t)c1 : 0u, y);
        out[k] = orc_bits_to_normal(y[0]);
        if (c1 < n) out[c1] = orc_bits_to_normal(y[1]);
    }
}

/* SPEC.md §7.3: the noise tensor of one solve, canonical f32[P][H][6] = normal(key, P*H*6) */
void orc_noise_from_key(const uint32_t key[2], int P, int H, float* out) { orc_normal(key, (size_t)P * H * 6, out); }
