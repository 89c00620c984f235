/*
 * sde_mpc_oracle.c — CPU restatement ("oracle") of the MPC inner loop. TEST INFRASTRUCTURE ONLY.
 *
 *   Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's .so.
 *   The product path (sde4mbrl_px4_amd/csrc) never includes, links or calls anything in oracle/.
 *
 * PARITY UNPINNED.  The reference (wuwushrek/sde4mbrl_px4) does not contain this arithmetic: it
 * imports it from the un-vendored, un-pinned JAX package `sde4mbrl` / `sde4mbrlExamples`
 *   sde4mbrl_px4/mpc_controller/sde_control.py:12  from sde4mbrlExamples.rotor_uav.sde_mpc_design import load_mpc_from_cfgfile
 *   sde4mbrl_px4/mpc_controller/sde_control.py:13  from sde4mbrlExamples.rotor_uav.utils import enu2ned
 * which is absent from /root/reference and from this image (no jax either), and the reference holds
 * no tests, golden vectors or fixtures for the path. This file therefore
 * restates the *published structure* of that algorithm (physics-structured rotor neural SDE,
 * Euler–Maruyama particles, accelerated proximal gradient with Armijo backtracking; Djeumou et al.,
 * CoRL 2023) as specified op-by-op in SPEC.md, and is anchored on what the reference does fix:
 *   - call signatures/shapes at sde_control.py:702,706,713,717,345-350,400-416
 *   - the hyper-parameter schema launch/iris_sitl_traj_mpc.yaml:8-85, iris_sitl_posctrl_mpc.yaml:6-101
 *   - telemetry fields sde_control.py:444-450, msg/OptMPCState.msg:6-22
 *   - post-processing sde_control.py:428-432
 *
 * Everything is float32 with an explicit operation order (fmaf where SPEC.md says fma) so that an
 * independent implementation following SPEC.md is reproducible bit for bit. Compile with
 *   gcc -O2 -ffp-contract=off [-mfma] -shared -fPIC
 * Compile with -DORC_DOUBLE for the float64 build (exact libm activations) used by the
 * finite-difference gradient tests; symbols are then prefixed orcd_.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/sdempc.h"
#if !defined(ORC_DOUBLE) && !defined(ORC_VEC)
#include "mfma16_model.c"      /* SPEC.md §9a: the 16-bit-operand matrix instruction, as the checker models it (same translation unit: the group addition inlines) */
#include "transc_model.c"      /* SPEC.md §10a: v_exp_f32 / v_rcp_f32 / v_rsq_f32, as the checker models them (math_mode: fast) */
#define ORC_MFMA16 1
#endif

#ifdef ORC_DOUBLE
typedef double real;
#define FMA(a, b, c) fma((a), (b), (c))
#define FABS(a) fabs(a)
#define NAME(x) orcd_##x
#else
typedef float real;
#define FMA(a, b, c) fmaf((a), (b), (c))
#define FABS(a) fabsf(a)
#define NAME(x) orc_##x
#endif
#define R(x) ((real)(x))

/* Lane blocks. The per-particle functions below (elementary functions, step_fwd, stage_cost, step_vjp) are written on the type `preal`:
 * one particle per call in the two checker builds (preal = real, VL = 1: the bit-exact float32 oracle and the float64 build), and
 * VL = 16 particles per call in the TIMING build (-DORC_VEC: GCC vector extensions, AVX-512 / AVX2 code under -O3 -march=native,
 * exported as orcv_*; bench.py's cpu_baseline leg). The vector build lets the compiler contract a*b+c and is therefore compared with
 * the checker build within a tolerance only (tests/test_oracle_cpu.py); it is never used as the checker. */
#ifdef ORC_VEC
#undef NAME
#define NAME(x) orcv_##x
#define VL 16
typedef float preal __attribute__((vector_size(4 * VL)));
typedef uint32_t puint __attribute__((vector_size(4 * VL)));
typedef int32_t pint __attribute__((vector_size(4 * VL)));
#define PFMA(a, b, c) ((a) * (b) + (c))                 /* contracted to vfmadd by -ffp-contract=fast */
#define PUB static inline                               /* the elementary functions are internal in this build */
static inline preal psel(pint m, preal a, preal b) { return (preal)(((puint)m & (puint)a) | (~(puint)m & (puint)b)); }
static inline preal pclamp(preal x, float lo, float hi) {
    preal l = lo - (preal){0}, h = hi - (preal){0};
    x = psel(x > l, x, l);                              /* NaN -> lo, as in the checker build */
    return psel(x > h, h, x);
}
static inline preal pbroadcast(real v) { return v - (preal){0}; }
#define PLANE(v, l) ((v)[l])
#define PPOS(v) psel((v) < pbroadcast(R(0)), pbroadcast(R(0)), (v))      /* v < 0 ? 0 : v  (a NaN stays) */
#define F16Q(on, v) (v)                                 /* fp16-operand mode is not built for the timing build (ctx_init refuses it) */
#else
#define VL 1
typedef real preal;
typedef uint32_t puint;
#define PFMA(a, b, c) FMA(a, b, c)
#define PUB
#define pbroadcast(v) (v)
#define PLANE(v, l) (v)
#define PPOS(v) ((v) < R(0) ? R(0) : (v))
#define F16Q(on, v) ((on) ? f16_rtz(v) : (v))
#endif

#define NX 13
#define NN 6
#define HID 32
#define MAXM 8
#define NSLOT 4 /* particle-group reduction slots (SPEC.md §6) */

/* ------------------------------------------------------------------------------------------- */
/* SPEC.md §3: elementary functions                                                            */
/* ------------------------------------------------------------------------------------------- */
#ifndef ORC_DOUBLE
static inline preal as_f(puint u) { preal f; memcpy(&f, &u, sizeof f); return f; }
static inline puint as_u(preal f) { puint u; memcpy(&u, &f, sizeof u); return u; }
#ifndef ORC_VEC
#define pclamp(x, lo, hi) fminf(fmaxf((x), (lo)), (hi))
#endif

/* reciprocal of d > 0: magic-constant seed + 3 Newton steps, every step an fma */
PUB preal NAME(rcp)(preal d) {
    preal y = as_f(0x7EF311C7u - as_u(d));
    for (int i = 0; i < 3; ++i) { preal e = PFMA(-d, y, 1.0f); y = PFMA(y, e, y); }
    return y;
}
/* 1/sqrt(a), a > 0: magic seed + 3 Newton steps */
PUB preal NAME(rsqrt)(preal a) {
    preal y = as_f(0x5F3759DFu - (as_u(a) >> 1));
    preal h = 0.5f * a;
    for (int i = 0; i < 3; ++i) { preal t = y * y; t = PFMA(-h, t, 1.5f); y = y * t; }
    return y;
}
/* 2^(x*c) for |x*c| <= 64 (SPEC.md §3.3): t2 = fma(x,c,1.5*2^23) holds n = rne(x*c) in its low
 * mantissa bits; f = fma(x,c,-n) in [-0.5,0.5]; degree-5 polynomial; exponent add by integer shift */
static inline preal exp2_spec(preal x, float c) {
    preal t2 = PFMA(x, c, 12582912.0f);
    preal n = t2 - 12582912.0f;
    preal f = PFMA(x, c, -n);
    preal p = pbroadcast(0.001327647129073739f);
    p = PFMA(p, f, 0.009675540961325169f);
    p = PFMA(p, f, 0.05550713092088699f);
    p = PFMA(p, f, 0.24022120237350464f);
    p = PFMA(p, f, 0.6931469440460205f);
    p = PFMA(p, f, 1.0000001192092896f);
    return as_f(as_u(p) + (as_u(t2) << 23));
}
/* tanh of 4 values sharing ONE reciprocal (batched inversion, SPEC.md §3.4):
 * d_i = 1 + exp(2 x_i); r = 1/(d0 d1 d2 d3); 1/d_i recovered by multiplications; tanh = 1 - 2/d_i */
PUB void NAME(tanh4)(const preal* x, preal* y) {
    preal d[4];
    for (int i = 0; i < 4; ++i) {
        preal xc = pclamp(x[i], -9.0f, 9.0f);
        d[i] = 1.0f + exp2_spec(xc, 2.885390043258667f); /* 2*log2(e) */
    }
    preal p2 = d[0] * d[1], p3 = p2 * d[2], p4 = p3 * d[3];
    preal r = NAME(rcp)(p4);
    preal r3 = r * p3; r = r * d[3];
    preal r2 = r * p2; r = r * d[2];
    preal r1 = r * d[0];
    preal r0 = r * d[1];
    y[0] = PFMA(-2.0f, r0, 1.0f); y[1] = PFMA(-2.0f, r1, 1.0f); y[2] = PFMA(-2.0f, r2, 1.0f); y[3] = PFMA(-2.0f, r3, 1.0f);
}
PUB preal NAME(tanh)(preal x) { preal a[4] = {x, x, x, x}, y[4]; NAME(tanh4)(a, y); return y[0]; }
PUB preal NAME(sigmoid)(preal x) {
    preal xc = pclamp(x, -30.0f, 30.0f);
    preal E = exp2_spec(xc, -1.4426950216293335f); /* -log2(e) */
    return NAME(rcp)(1.0f + E);
}
#else
double NAME(rcp)(double d) { return 1.0 / d; }
double NAME(rsqrt)(double a) { return 1.0 / sqrt(a); }
double NAME(tanh)(double x) { return tanh(x); }
void NAME(tanh4)(const double* x, double* y) { for (int i = 0; i < 4; ++i) y[i] = tanh(x[i]); }
double NAME(sigmoid)(double x) { return 1.0 / (1.0 + exp(-x)); }
#endif

/* ------------------------------------------------------------------------------------------- */
/* model blob (SPEC.md §2)                                                                     */
/* ------------------------------------------------------------------------------------------- */
typedef struct {
    int m;
    int f16;   /* mlp_dtype: 1 = SPEC.md §9 fp16-operand MLP contractions, 2 = §9b three-limb bf16 split of the layer-2 contractions */
    int fast;  /* math_mode fast (SPEC.md §10): tanh / sigmoid / rsqrt through the hardware's transcendental instructions, modelled in transc_model.c */
#ifdef ORC_MFMA16
    /* operands of the matrix instruction, decoded once, in k-slot order: slot k of K-half hf is hidden unit u(hf,k) = rowmap(8 hf + (k & 7), k >> 3) */
    orc_op16 h1[2 * HID][16];            /* f16 mode, layer 1: row r, slots 0..5 = W1z[r][k], the rest zero */
    orc_op16 h2[HID][2][16];             /* f16 mode, layer 2: row i, K-half hf */
    int32_t x2m[2][3][HID][2][16], x2x[2][3][HID][2][16];       /* f32x3 mode: significands / exponents of the bf16 limbs, [0 = W2 rows, 1 = W2^T rows][limb][row][K-half][slot] */
    int32_t x2mT[2][3][2][16][HID], x2xT[2][3][2][16][HID];     /* the same, row index last: sixteen rows per vector in the 16-lane group addition (mfma16_model.c) */
    /* SPEC.md §10e (f32x3 + fast): the three contractions of the MLP's vector-Jacobian products from two binary16 limbs each, behind a per-particle
     * power-of-two scale. Images [0 = (4 W2)^T, 1 = density tile: rows 0..5 = W1z[32 + k][row], 2 = drift tile: rows 0..5 = W1z[k][row], rows 6..6+m-1 = W1u[k][row - 6]][limb][row][K-half][slot] */
    int adjmp, adj_eoff;
    int32_t y2m[4][3][HID][2][16], y2x[4][3][HID][2][16], y2mT[4][3][2][16][HID], y2xT[4][3][2][16][HID];     /* image 3: (-2 W3)^T in k slots 0..5 of K-half 0 (row = layer-2 unit) */
#endif
    real inv_mass, grav, J[3], iJ[3], ct2, ct1, ct0, cm2, cm1;
    real rx[MAXM], ry[MAXM], dir[MAXM];
    real sF[3], sT[3], sigma[NN];
    real W1z[2 * HID][NN], b1[2 * HID], W1u[HID][MAXM], W2[HID][HID], b2[HID], W3[6][HID], b3[6], w3n[HID], b3n;     /* what the forward pass uses */
    real vW1z[2 * HID][NN], vW1u[HID][MAXM], vW2[HID][HID], vW3[6][HID], vw3n[HID];   /* what the vector-Jacobian products use: the same values, except in math_mode fast (SPEC.md §10b) */
} model_t;

static real f16_rtz(real xv);
#ifdef ORC_MFMA16
static inline int slot_unit(int hf, int k) { const int r = 8 * hf + (k & 7), h = k >> 3; return (r & 3) + 8 * (r >> 2) + 4 * h; }
/* binary16 pattern of a value that is exactly representable in binary16 (the outputs of f16_rtz) */
static uint16_t f16_bits(float v) {
    uint32_t u; memcpy(&u, &v, 4);
    const uint32_t sign = (u >> 16) & 0x8000u, mag = u & 0x7FFFFFFFu;
    if (mag >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | ((mag & 0x7FFFFFu) ? 0x200u : 0u));
    if (mag == 0) return (uint16_t)sign;
    const int e = (int)(mag >> 23) - 127;
    if (e >= -14) return (uint16_t)(sign | ((uint32_t)(e + 15) << 10) | ((mag >> 13) & 0x3FFu));
    float a; memcpy(&a, &mag, 4);
    return (uint16_t)(sign | (uint32_t)(a * 16777216.0f));          /* sub-normal: multiples of 2^-24 */
}
/* SPEC.md §9b: x = x1 + x2 + x3 (+ less than 2^-24 |x|), each limb a truncation to bf16, both subtractions exact */
static void bf16_limbs(float x, uint16_t* lb) {
    for (int l = 0; l < 3; ++l) {
        uint32_t u; memcpy(&u, &x, 4);
        const uint32_t hi = u & 0xFFFF0000u;
        lb[l] = (uint16_t)(hi >> 16);
        float h; memcpy(&h, &hi, 4);
        x = x - h;
    }
}
/* SPEC.md §10c: round to nearest even to binary16 (sub-normals kept, overflow to infinity, NaN stays NaN): what v_cvt_pk_f16_f32 returns in the
 * kernels' FP mode; and the value of a binary16 pattern */
static uint16_t f16_rne_bits(float x) {
    uint32_t u; memcpy(&u, &x, 4);
    const uint32_t sign = (u >> 16) & 0x8000u, mag = u & 0x7FFFFFFFu;
    if (mag > 0x7F800000u) return (uint16_t)(sign | 0x7E00u);
    if (mag >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);                  /* >= 65520 rounds to infinity */
    if (mag < 0x33000000u) return (uint16_t)sign;                                /* < 2^-25: zero (2^-25 itself ties to even = 0) */
    const int e = (int)(mag >> 23) - 127;
    uint32_t m = (mag & 0x7FFFFFu) | 0x800000u;                                 /* 24-bit significand */
    int shift = e >= -14 ? 13 : 13 + (-14 - e);                                  /* bits to drop: 13 for normals, more for sub-normals */
    if (shift > 24) { if (mag == 0x33000000u) return (uint16_t)sign; shift = 25; m = m; }
    uint32_t q = shift >= 32 ? 0u : (m >> shift);
    const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q += 1u;
    uint32_t bits;
    if (e >= -14) bits = ((uint32_t)(e + 15) << 10) + (q - 0x400u);             /* a carry out of the significand moves into the exponent by itself */
    else bits = q;                                                              /* sub-normal (q == 0x400: the smallest normal) */
    return (uint16_t)(sign | bits);
}
static float f16_value(uint16_t h) {
    const int s = h >> 15, ef = (h >> 10) & 31, f = h & 1023;
    float v;
    if (ef == 31) v = f ? NAN : INFINITY;
    else if (ef == 0) v = ldexpf((float)f, -24);
    else v = ldexpf((float)(f | 1024), ef - 25);
    return s ? -v : v;
}
uint16_t NAME(f16_rne)(float x) { return f16_rne_bits(x); }       /* test entry */
/* out[i] = c[i] + sum_k W[i][k] v[k] on the matrix pipe. kind 0: the twelve bf16 instructions of SPEC.md §9b (three limbs by truncation, six limb
 * products); kind 1: the eight f16 instructions of §10c (two limbs, round to nearest even, four limb products: math_mode fast's forward contraction,
 * whose activations lie in [0, 1]). Wm / Wx: M->x2m[tr], M->x2x[tr]; WmT / WxT: the row-last copies (or NULL: scalar evaluation only).
 * The 16-lane form of the group addition is taken where the CPU has it; x3_scalar_only forces the scalar (normative) one */
static int x3_scalar_only = 0;
void NAME(x3_force_scalar)(int on) { x3_scalar_only = on; }
static void x3_contract(int kind, const int32_t (*Wm)[HID][2][16], const int32_t (*Wx)[HID][2][16], const int32_t (*WmT)[2][16][HID], const int32_t (*WxT)[2][16][HID],
                        const float* v, const float* c, float* out) {
    static const int WA3[6] = {2, 1, 1, 0, 0, 0}, VB3[6] = {0, 1, 0, 2, 1, 0}, WA2[4] = {1, 1, 0, 0}, VB2[4] = {1, 0, 1, 0};
    const int np = kind ? 4 : 6;
    const int *WA = kind ? WA2 : WA3, *VB = kind ? VB2 : VB3;
    int32_t vm[3][2][16], vx[3][2][16];
    int special = 0;
    for (int hf = 0; hf < 2; ++hf) for (int k = 0; k < 16; ++k) {
        uint16_t lb[3];
        const float x = v[slot_unit(hf, k)];
        if (kind) { lb[0] = f16_rne_bits(x); lb[1] = f16_rne_bits(x - f16_value(lb[0])); lb[2] = 0; }       /* (x - limb is exact; a NaN stays one) */
        else bf16_limbs(x, lb);
        for (int l = 0; l < (kind ? 2 : 3); ++l) { orc_op16 o; orc_mfma16_decode(!kind, lb[l], &o); vm[l][hf][k] = o.m; vx[l][hf][k] = o.ex; special |= o.kind; }
    }
    if (!special && WmT && !x3_scalar_only && orc_mfma16_vec_available()) {
        for (int blk = 0; blk < HID; blk += 16) {
            float acc[16];
            for (int l = 0; l < 16; ++l) acc[l] = c ? c[blk + l] : 0.0f;       /* (a non-finite start value stays as it is: the group addition keeps it) */
            for (int s6 = 0; s6 < np; ++s6) for (int hf = 0; hf < 2; ++hf) for (int g8 = 0; g8 < 16; g8 += 8) {
                if (kind) orc_mfma16_group8_f16_x16(&WmT[WA[s6]][hf][g8][blk], &WxT[WA[s6]][hf][g8][blk], HID, &vm[VB[s6]][hf][g8], &vx[VB[s6]][hf][g8], acc);
                else orc_mfma16_group8_bf16_x16(&WmT[WA[s6]][hf][g8][blk], &WxT[WA[s6]][hf][g8][blk], HID, &vm[VB[s6]][hf][g8], &vx[VB[s6]][hf][g8], acc);
            }
            for (int l = 0; l < 16; ++l) out[blk + l] = acc[l];
        }
        return;
    }
    for (int i = 0; i < HID; ++i) {
        float acc = c ? c[i] : 0.0f;
        if (special) {                        /* a non-finite activation: its second limb is inf - inf = NaN and meets every weight */
            acc = NAN;
        } else if (!isfinite(acc)) {          /* finite products on a non-finite start value leave it as it is */
        } else {
            for (int s6 = 0; s6 < np; ++s6) for (int hf = 0; hf < 2; ++hf) {
                const int32_t *wm = Wm[WA[s6]][i][hf], *wx = Wx[WA[s6]][i][hf], *am = vm[VB[s6]][hf], *ax = vx[VB[s6]][hf];
                if (kind) { acc = orc_mfma16_group8_f16(wm, wx, am, ax, acc); acc = orc_mfma16_group8_f16(wm + 8, wx + 8, am + 8, ax + 8, acc); }
                else { acc = orc_mfma16_group8_bf16(wm, wx, am, ax, acc); acc = orc_mfma16_group8_bf16(wm + 8, wx + 8, am + 8, ax + 8, acc); }
            }
        }
        out[i] = acc;
    }
}
/* Test entry (tests/test_mfma16_model_cpu.py): one 32x32 contraction out[i] = c[i] + sum_k W[i][k] v[k] on caller-supplied operands, in the
 * f32 fma chain of SPEC.md §4 (mode 0) or as the twelve instructions of §9b (mode 2) — the two arithmetics of layer 2, without a model around them */
void NAME(contract32)(int mode, const float* W, const float* v, const float* c, float* out) {
    if (mode == 2 || mode == 3) {          /* 3: the f16 two-limb form of SPEC.md §10c */
        static int32_t Wm[3][HID][2][16], Wx[3][HID][2][16], WmT[3][2][16][HID], WxT[3][2][16][HID];
        for (int i = 0; i < HID; ++i) for (int hf = 0; hf < 2; ++hf) for (int k = 0; k < 16; ++k) {
            uint16_t lb[3];
            const float w = W[i * HID + slot_unit(hf, k)];
            if (mode == 3) { lb[0] = f16_rne_bits(w); lb[1] = f16_rne_bits(w - f16_value(lb[0])); lb[2] = 0; }
            else bf16_limbs(w, lb);
            for (int l = 0; l < 3; ++l) { orc_op16 o; orc_mfma16_decode(mode == 2, lb[l], &o); Wm[l][i][hf][k] = WmT[l][hf][k][i] = o.m; Wx[l][i][hf][k] = WxT[l][hf][k][i] = o.ex; }
        }
        x3_contract(mode == 3, Wm, Wx, WmT, WxT, v, c, out);
        return;
    }
    for (int i = 0; i < HID; ++i) {
        float acc = c ? c[i] : 0.0f;
        for (int r = 0; r < 16; ++r) for (int h = 0; h < 2; ++h) { const int k = (r & 3) + 8 * (r >> 2) + 4 * h; acc = fmaf(W[i * HID + k], v[k], acc); }
        out[i] = acc;
    }
}
#endif

/* derivative of the activation from what the forward pass keeps: h = tanh: 1 - h^2; math_mode fast keeps r (tanh = 1 - 2 r): r - r^2, its factor 4 is in the transposed weights */
#define DACT(M, h) ((M)->fast ? PFMA(-(h), (h), (h)) : PFMA(-(h), (h), R(1)))
/* activations of a step: SPEC.md §3 (default) or, in math_mode fast, §10 on the modelled instructions (float32 checker build only) */
static inline void act_tanh4(const model_t* M, const preal* x, preal* y) {
#ifdef ORC_MFMA16
    if (M->fast) {
        orc_hw_sigm4(x, y);       /* SPEC.md §10b: r_i = v_rcp_f32(1 + v_exp_f32(x_i)), with tanh = 1 - 2 r folded into the weights */
        return;
    }
#endif
    NAME(tanh4)(x, y);
}
static inline preal act_sigmoid(const model_t* M, preal x) {
#ifdef ORC_MFMA16
    if (M->fast) return orc_hw_rcp(1.0f + orc_hw_exp2(x * -1.4426950216293335f));
#endif
    return NAME(sigmoid)(x);
}
static inline preal act_rsqrt(const model_t* M, preal a) {
#ifdef ORC_MFMA16
    if (M->fast) return orc_hw_rsq(a);
#endif
    return NAME(rsqrt)(a);
}

static int parse_blob(const void* blob, model_t* M, int f16, int fast) {
    const int32_t* hd = (const int32_t*)blob;
    if (hd[0] != SDEMPC_BLOB_MAGIC || hd[1] != 1) return -1;
    M->m = hd[2];
    if (M->m < 1 || M->m > MAXM || hd[3] != HID || hd[4] != NN || hd[5] != NN) return -1;
    const float* f = (const float*)(hd + SDEMPC_BLOB_HEADER_INTS);
    M->inv_mass = f[0]; M->grav = f[1];
    for (int i = 0; i < 3; ++i) { M->J[i] = f[2 + i]; M->iJ[i] = f[5 + i]; }
    M->ct2 = f[8]; M->ct1 = f[9]; M->ct0 = f[10]; M->cm2 = f[11]; M->cm1 = f[12];
    f += 16;
    for (int j = 0; j < MAXM; ++j) { M->rx[j] = f[j]; M->ry[j] = f[8 + j]; M->dir[j] = f[16 + j]; }
    f += 24;
    for (int i = 0; i < 3; ++i) { M->sF[i] = f[i]; M->sT[i] = f[3 + i]; }
    f += 8;
    for (int i = 0; i < NN; ++i) M->sigma[i] = f[i];
    f += 8;
    for (int r = 0; r < 2 * HID; ++r) for (int k = 0; k < NN; ++k) M->W1z[r][k] = f[r * NN + k];
    f += 2 * HID * NN;
    for (int r = 0; r < 2 * HID; ++r) M->b1[r] = f[r];
    f += 2 * HID;
    for (int r = 0; r < HID; ++r) for (int j = 0; j < MAXM; ++j) M->W1u[r][j] = f[r * MAXM + j];
    f += HID * MAXM;
    for (int r = 0; r < HID; ++r) for (int k = 0; k < HID; ++k) M->W2[r][k] = f[r * HID + k];
    f += HID * HID;
    for (int r = 0; r < HID; ++r) M->b2[r] = f[r];
    f += HID;
    for (int i = 0; i < 6; ++i) for (int k = 0; k < HID; ++k) M->W3[i][k] = f[i * HID + k];
    f += 8 * HID;
    for (int i = 0; i < 6; ++i) M->b3[i] = f[i];
    f += 8;
    for (int k = 0; k < HID; ++k) M->w3n[k] = f[k];
    f += HID;
    M->b3n = f[0];
    M->f16 = f16;
    if (f16 == 1) { /* layer-1 (state inputs) and layer-2 weights live in fp16, forward and adjoint alike */
        for (int r = 0; r < 2 * HID; ++r) for (int k = 0; k < NN; ++k) M->W1z[r][k] = f16_rtz(M->W1z[r][k]);
        for (int r = 0; r < HID; ++r) for (int k = 0; k < HID; ++k) M->W2[r][k] = f16_rtz(M->W2[r][k]);
    }
    memcpy(M->vW1z, M->W1z, sizeof M->W1z); memcpy(M->vW1u, M->W1u, sizeof M->W1u); memcpy(M->vW2, M->W2, sizeof M->W2);
    memcpy(M->vW3, M->W3, sizeof M->W3); memcpy(M->vw3n, M->w3n, sizeof M->w3n);
    M->fast = 0;
#ifdef ORC_MFMA16
    if (fast) {
        /* SPEC.md §10b (the same float32 statements as sdempc_create's): the hardware tanh is kept as r = rcp(1 + exp2(a')), a' = (2 log2 e) a, tanh = 1 - 2 r.
         * Weights and biases that feed a tanh take the pre-scale (one rounding each); those that consume one take the affine map (factor -2 exact, biases by
         * sequential sums); 1 - tanh^2 = 4 (r - r^2) leaves its 4 in the transposed weights. */
        const float c = 2.885390043258667f;
        M->fast = 1;
        for (int r = 0; r < 2 * HID; ++r) for (int k = 0; k < NN; ++k) { float w = c * M->vW1z[r][k]; M->W1z[r][k] = f16 == 1 ? f16_rtz(w) : w; }
        for (int r = 0; r < 2 * HID; ++r) M->b1[r] = c * M->b1[r];
        for (int r = 0; r < HID; ++r) for (int j = 0; j < MAXM; ++j) M->W1u[r][j] = c * M->vW1u[r][j];
        for (int j = 0; j < HID; ++j) {
            float sum = c * M->b2[j];
            for (int k = 0; k < HID; ++k) {
                float w = c * M->vW2[j][k];
                if (f16 == 1) w = f16_rtz(w);
                sum = sum + w;
                M->W2[j][k] = -2.0f * w;
            }
            M->b2[j] = sum;
        }
        for (int i = 0; i < 6; ++i) {
            float sum = M->b3[i];
            for (int k = 0; k < HID; ++k) { sum = sum + M->vW3[i][k]; M->W3[i][k] = -2.0f * M->vW3[i][k]; }
            M->b3[i] = sum;
        }
        {
            float sum = M->b3n;
            for (int k = 0; k < HID; ++k) { sum = sum + M->vw3n[k]; M->w3n[k] = -2.0f * M->vw3n[k]; }
            M->b3n = sum;
        }
        for (int j = 0; j < HID; ++j) for (int k = 0; k < HID; ++k) M->vW2[j][k] = 4.0f * M->vW2[j][k];
        for (int i = 0; i < 6; ++i) for (int k = 0; k < HID; ++k) M->vW3[i][k] = 4.0f * M->vW3[i][k];
        for (int k = 0; k < HID; ++k) M->vw3n[k] = 4.0f * M->vw3n[k];
        if (f16 == 1) {         /* the forward pass evaluates the re-quantised c * w: the vector-Jacobian products differentiate those weights (SPEC.md §10b) */
            for (int r = 0; r < 2 * HID; ++r) for (int k = 0; k < NN; ++k) M->vW1z[r][k] = M->W1z[r][k] / c;
            for (int j = 0; j < HID; ++j) for (int k = 0; k < HID; ++k) M->vW2[j][k] = (-2.0f * M->W2[j][k]) / c;
        }
    }
    if (f16 == 1) {
        for (int r = 0; r < 2 * HID; ++r) for (int k = 0; k < 16; ++k) orc_mfma16_decode(0, k < NN ? f16_bits(M->W1z[r][k]) : 0, &M->h1[r][k]);
        for (int i = 0; i < HID; ++i) for (int hf = 0; hf < 2; ++hf) for (int k = 0; k < 16; ++k)
            orc_mfma16_decode(0, f16_bits(M->W2[i][slot_unit(hf, k)]), &M->h2[i][hf][k]);
    }
    if (f16 == 2) { /* SPEC.md §9b: three bf16 limbs of every W2 entry by truncation; W2 itself stays f32 */
        for (int tr = 0; tr < 2; ++tr) for (int i = 0; i < HID; ++i) for (int hf = 0; hf < 2; ++hf) for (int k = 0; k < 16; ++k) {
            const int un = slot_unit(hf, k);
            uint16_t lb[3];
            const int h2 = !tr && M->fast;          /* SPEC.md §10c: the forward contraction of math_mode fast takes two f16 limbs (round to nearest even) */
            if (h2) { const float w = M->W2[i][un]; lb[0] = f16_rne_bits(w); lb[1] = f16_rne_bits(w - f16_value(lb[0])); lb[2] = 0; }
            else bf16_limbs(tr ? M->vW2[un][i] : M->W2[i][un], lb);
            for (int l = 0; l < 3; ++l) { orc_op16 o; orc_mfma16_decode(!h2, lb[l], &o); M->x2m[tr][l][i][hf][k] = M->x2mT[tr][l][hf][k][i] = o.m; M->x2x[tr][l][i][hf][k] = M->x2xT[tr][l][hf][k][i] = o.ex; }
        }
    }
    M->adjmp = 0; M->adj_eoff = 10;
    if ((f16 == 2 || f16 == 1) && M->fast) {
        /* SPEC.md §10e (both matrix-pipe contraction modes in math_mode fast): the scale offset (the same float32 statements as sdempc_create's: absolute column sums in ascending index order) */
        M->adjmp = 1;
        float B3 = 0.0f, Bn = 0.0f, C2 = 0.0f;
        for (int k = 0; k < HID; ++k) {
            float s3 = 0.0f, s2 = 0.0f;
            for (int i = 0; i < 6; ++i) s3 = s3 + fabsf(M->W3[i][k]);
            for (int j = 0; j < HID; ++j) s2 = s2 + fabsf(M->vW2[j][k]);
            B3 = fmaxf(B3, s3); C2 = fmaxf(C2, s2); Bn = fmaxf(Bn, fabsf(M->w3n[k]));
        }
        const float bound = fmaxf(fmaxf(B3 * 0.25f, Bn * 0.25f), (C2 * (B3 * 0.25f)) * 0.25f);
        int eb = 0;
        if (bound > 0.0f && bound < INFINITY) { (void)frexpf(bound, &eb); if (14 - eb < M->adj_eoff) M->adj_eoff = 14 - eb; }
        if (M->adj_eoff < -40) M->adj_eoff = -40;
        for (int img = 0; img < 4; ++img) for (int i = 0; i < HID; ++i) for (int hf = 0; hf < 2; ++hf) for (int k = 0; k < 16; ++k) {
            const int un = slot_unit(hf, k);
            float w = 0.0f;
            if (img == 0) w = M->vW2[un][i];                                   /* (4 W2)^T: row i = input unit, contraction over the layer-2 unit un */
            else if (img == 1) w = i < NN ? M->vW1z[HID + un][i] : 0.0f;
            else if (img == 2) w = i < NN ? M->vW1z[un][i] : (i < NN + M->m ? M->vW1u[un][i - NN] : 0.0f);
            else w = (hf == 0 && k < 6) ? M->W3[k][i] : 0.0f;                  /* the six output adjoints sit in k slots 0..5 of K-half 0 */
            uint16_t lb[3] = {f16_rne_bits(w), 0, 0};
            lb[1] = f16_rne_bits(w - f16_value(lb[0]));
            for (int l = 0; l < 3; ++l) { orc_op16 o; orc_mfma16_decode(0, lb[l], &o); M->y2m[img][l][i][hf][k] = M->y2mT[img][l][hf][k][i] = o.m; M->y2x[img][l][i][hf][k] = M->y2xT[img][l][hf][k][i] = o.ex; }
        }
    }
#endif
    return 0;
}

/* SPEC.md §9: round toward zero to the nearest IEEE binary16 value (result returned as real).
 * Finite overflow saturates at 65504, subnormals keep the 2^-24 grid. */
static real f16_rtz(real xv) {
    float x = (float)xv;
    uint32_t u; memcpy(&u, &x, 4);
    uint32_t sign = u & 0x80000000u, mag = u & 0x7FFFFFFFu;
    if (mag >= 0x7F800000u) return xv;                      /* inf / nan: unchanged */
    int e = (int)(mag >> 23) - 127;
    float r;
    if (e > 15) { r = 65504.0f; }
    else if (e >= -14) { uint32_t t = mag & ~((1u << 13) - 1u); memcpy(&r, &t, 4); }
    else { /* subnormal half: multiples of 2^-24 */
        float a; memcpy(&a, &mag, 4);
        r = (float)floor((double)a * 16777216.0) / 16777216.0f;
    }
    uint32_t ru; memcpy(&ru, &r, 4); ru |= sign; memcpy(&r, &ru, 4);
    return (real)r;
}

/* hidden-unit visiting order of SPEC.md §4: k(r,h) = (r&3) + 8*(r>>2) + 4*h, r = 0..15, h = 0..1 */
static inline int rowmap(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

/* ------------------------------------------------------------------------------------------- */
/* per-step control-dependent constants (SPEC.md §5.1)                                         */
/* serves: rollout inside m_mpc (call sites sde_control.py:400-416); body NOT IN REFERENCE      */
/* ------------------------------------------------------------------------------------------- */
typedef struct {
    real c[HID];      /* drift layer-1 pre-activation offset b1 + W1u u_t */
    real Tz, tau[3];  /* total thrust and rotor torques */
    real dT[MAXM], dM[MAXM];
} ustep_t;

static void ustep_eval(const model_t* M, const real* u, ustep_t* U) {
    int m = M->m;
    for (int r = 0; r < HID; ++r) {
        real a = M->b1[r];
        for (int j = 0; j < m; ++j) a = FMA(M->W1u[r][j], u[j], a);
        U->c[r] = a;
    }
    real Tz = 0, t0 = 0, t1 = 0, t2 = 0;
    for (int j = 0; j < m; ++j) {
        real T = FMA(FMA(M->ct2, u[j], M->ct1), u[j], M->ct0);
        real Mq = M->dir[j] * (FMA(M->cm2, u[j], M->cm1) * u[j]);
        Tz = Tz + T;
        t0 = FMA(M->ry[j], T, t0);
        t1 = FMA(-M->rx[j], T, t1);
        t2 = t2 + Mq;
        U->dT[j] = FMA(R(2) * M->ct2, u[j], M->ct1);
        U->dM[j] = M->dir[j] * FMA(R(2) * M->cm2, u[j], M->cm1);
    }
    U->Tz = Tz; U->tau[0] = t0; U->tau[1] = t1; U->tau[2] = t2;
}

/* ------------------------------------------------------------------------------------------- */
/* one Euler–Maruyama step (SPEC.md §5.2) and its vector-Jacobian product (§5.4)               */
/* SURVEY.md §8a A4/A6; knobs launch/iris_sitl_traj_mpc.yaml:44-52; body NOT IN REFERENCE      */
/* ------------------------------------------------------------------------------------------- */
typedef struct { /* values the VJP re-uses; recomputed from x_t in the backward sweep */
    preal Rm[9], vb[3], h1d[HID], h1n[HID], h2[HID], o[6], eta, Fb[3], Jom[3], qt[4], rn, qn[4];
} stepaux_t;

static void step_fwd(const model_t* M, const ustep_t* U, const preal* x, const preal* xi, real dt,
                     const real* sdt, preal* xn, stepaux_t* A) {
    const preal *p = x, *v = x + 3, *q = x + 6, *om = x + 10;
    preal qw = q[0], qx = q[1], qy = q[2], qz = q[3];
    preal xx = qx * qx, yy = qy * qy, zz = qz * qz;
    preal xy = qx * qy, xz = qx * qz, yz = qy * qz, wx = qw * qx, wy = qw * qy, wz = qw * qz;
    preal* Rm = A->Rm;
    Rm[0] = PFMA(R(-2), yy + zz, R(1)); Rm[1] = R(2) * (xy - wz);          Rm[2] = R(2) * (xz + wy);
    Rm[3] = R(2) * (xy + wz);          Rm[4] = PFMA(R(-2), xx + zz, R(1)); Rm[5] = R(2) * (yz - wx);
    Rm[6] = R(2) * (xz - wy);          Rm[7] = R(2) * (yz + wx);          Rm[8] = PFMA(R(-2), xx + yy, R(1));
    /* body-frame velocity vb = R^T v */
    for (int j = 0; j < 3; ++j) A->vb[j] = PFMA(Rm[6 + j], v[2], PFMA(Rm[3 + j], v[1], Rm[j] * v[0]));
    preal z[NN] = {A->vb[0], A->vb[1], A->vb[2], om[0], om[1], om[2]};
#ifndef ORC_VEC
    if (M->f16 == 1) for (int k = 0; k < NN; ++k) z[k] = f16_rtz(z[k]);   /* activations enter the contraction in fp16 */
#endif
    /* layer 1: drift rows 0..31 start from U->c, density rows 32..63 from b1 */
    preal pre_d[HID], pre_n[HID], pre_2[HID];
#ifdef ORC_MFMA16
    if (M->f16 == 1) {      /* SPEC.md §9: ONE v_mfma_f32_32x32x16_f16 per tile: products k = 0..5, ten zero products, C = the start value */
        orc_op16 zb[16];
        int special = 0;
        for (int k = 0; k < 16; ++k) { orc_mfma16_decode(0, k < NN ? f16_bits(z[k]) : 0, &zb[k]); special |= zb[k].kind; }
        for (int r = 0; r < HID; ++r) {
            if (special || !isfinite(U->c[r])) {     /* a diverged rollout: IEEE rules */
                preal a = U->c[r], b = M->b1[HID + r];
                for (int k = 0; k < NN; ++k) { a = PFMA(M->W1z[r][k], z[k], a); b = PFMA(M->W1z[HID + r][k], z[k], b); }
                pre_d[r] = a; pre_n[r] = b;
            } else {
                pre_d[r] = orc_mfma16_group(M->h1[r], zb, 8, U->c[r]);
                pre_n[r] = orc_mfma16_group(M->h1[HID + r], zb, 8, M->b1[HID + r]);
            }
        }
    } else
#endif
    for (int r = 0; r < HID; ++r) {
        preal a = pbroadcast(U->c[r]), b = pbroadcast(M->b1[HID + r]);
        for (int k = 0; k < NN; ++k) { a = PFMA(M->W1z[r][k], z[k], a); b = PFMA(M->W1z[HID + r][k], z[k], b); }
        pre_d[r] = a; pre_n[r] = b;
    }
    for (int r = 0; r < HID; r += 4) { act_tanh4(M, pre_d + r, A->h1d + r); act_tanh4(M, pre_n + r, A->h1n + r); }
    /* layer 2 (drift): k visited in rowmap order */
#ifdef ORC_MFMA16
    if (M->f16 == 1) {      /* SPEC.md §9: TWO chained v_mfma_f32_32x32x16_f16 (hf = 0, 1) from C = b2[i] */
        orc_op16 hb[2][16];
        int special = 0;
        for (int hf = 0; hf < 2; ++hf) for (int k = 0; k < 16; ++k) { orc_mfma16_decode(0, f16_bits(f16_rtz(A->h1d[slot_unit(hf, k)])), &hb[hf][k]); special |= hb[hf][k].kind; }
        for (int i = 0; i < HID; ++i) {
            float acc = M->b2[i];
            if (special || !isfinite(acc)) {
                for (int r = 0; r < 16; ++r) for (int h = 0; h < 2; ++h) { int k = rowmap(r, h); acc = FMA(M->W2[i][k], f16_rtz(A->h1d[k]), acc); }
            } else {
                for (int hf = 0; hf < 2; ++hf) { acc = orc_mfma16_group(M->h2[i][hf], hb[hf], 8, acc); acc = orc_mfma16_group(M->h2[i][hf] + 8, hb[hf] + 8, 8, acc); }
            }
            pre_2[i] = acc;
        }
    } else if (M->f16 == 2) {
        x3_contract(M->fast, M->x2m[0], M->x2x[0], M->x2mT[0], M->x2xT[0], A->h1d, M->b2, pre_2);
    } else
#endif
    for (int i = 0; i < HID; ++i) {
        preal a = pbroadcast(M->b2[i]);
        for (int r = 0; r < 16; ++r) for (int h = 0; h < 2; ++h) { int k = rowmap(r, h); a = PFMA(M->W2[i][k], F16Q(M->f16 == 1, A->h1d[k]), a); }
        pre_2[i] = a;
    }
    for (int r = 0; r < HID; r += 4) act_tanh4(M, pre_2 + r, A->h2 + r);
    /* output layers: two half-sums (h = 0, 1) over r, then (P0 + P1) + bias */
    for (int i = 0; i < 6; ++i) {
        preal P0 = pbroadcast(R(0)), P1 = pbroadcast(R(0));
        for (int r = 0; r < 16; ++r) { P0 = PFMA(M->W3[i][rowmap(r, 0)], A->h2[rowmap(r, 0)], P0); P1 = PFMA(M->W3[i][rowmap(r, 1)], A->h2[rowmap(r, 1)], P1); }
        A->o[i] = (P0 + P1) + M->b3[i];
    }
    {
        preal P0 = pbroadcast(R(0)), P1 = pbroadcast(R(0));
        for (int r = 0; r < 16; ++r) { P0 = PFMA(M->w3n[rowmap(r, 0)], A->h1n[rowmap(r, 0)], P0); P1 = PFMA(M->w3n[rowmap(r, 1)], A->h1n[rowmap(r, 1)], P1); }
        A->eta = act_sigmoid(M, (P0 + P1) + M->b3n);
    }
    /* rigid body */
    A->Fb[0] = M->sF[0] * A->o[0]; A->Fb[1] = M->sF[1] * A->o[1]; A->Fb[2] = PFMA(M->sF[2], A->o[2], U->Tz);
    preal acc[3];
    for (int i = 0; i < 3; ++i) {
        preal Fw = PFMA(Rm[3 * i + 2], A->Fb[2], PFMA(Rm[3 * i + 1], A->Fb[1], Rm[3 * i] * A->Fb[0]));
        acc[i] = Fw * M->inv_mass;
    }
    acc[2] = acc[2] - M->grav;
    preal taub[3];
    for (int i = 0; i < 3; ++i) { taub[i] = PFMA(M->sT[i], A->o[3 + i], U->tau[i]); A->Jom[i] = M->J[i] * om[i]; }
    preal cr[3];
    cr[0] = PFMA(om[1], A->Jom[2], -(om[2] * A->Jom[1]));
    cr[1] = PFMA(om[2], A->Jom[0], -(om[0] * A->Jom[2]));
    cr[2] = PFMA(om[0], A->Jom[1], -(om[1] * A->Jom[0]));
    preal dom[3];
    for (int i = 0; i < 3; ++i) dom[i] = (taub[i] - cr[i]) * M->iJ[i];
    preal dq[4];
    dq[0] = R(-0.5) * PFMA(qz, om[2], PFMA(qy, om[1], qx * om[0]));
    dq[1] = R(0.5) * PFMA(-qz, om[1], PFMA(qy, om[2], qw * om[0]));
    dq[2] = R(0.5) * PFMA(-qx, om[2], PFMA(qz, om[0], qw * om[1]));
    dq[3] = R(0.5) * PFMA(-qy, om[0], PFMA(qx, om[1], qw * om[2]));
    /* Euler–Maruyama update */
    preal se[NN];
    for (int i = 0; i < NN; ++i) se[i] = sdt[i] * A->eta;
    for (int i = 0; i < 3; ++i) {
        xn[i] = PFMA(v[i], dt, p[i]);
        xn[3 + i] = PFMA(se[i], xi[i], PFMA(acc[i], dt, v[i]));
        xn[10 + i] = PFMA(se[3 + i], xi[3 + i], PFMA(dom[i], dt, om[i]));
    }
    for (int i = 0; i < 4; ++i) A->qt[i] = PFMA(dq[i], dt, q[i]);
    preal n2 = PFMA(A->qt[3], A->qt[3], PFMA(A->qt[2], A->qt[2], PFMA(A->qt[1], A->qt[1], A->qt[0] * A->qt[0])));
    A->rn = act_rsqrt(M, n2);
    for (int i = 0; i < 4; ++i) { A->qn[i] = A->qt[i] * A->rn; xn[6 + i] = A->qn[i]; }
}

/* stage state cost at x_{t+1} (SPEC.md §5.3): returns l, optionally the gradient wrt x_{t+1}.
 * SURVEY.md §8a A5; weights = cost_params of launch/iris_sitl_traj_mpc.yaml:32-41 */
static preal stage_cost(const sdempc_cfg* C, const preal* x, const real* xr, preal* gx) {
    preal l = pbroadcast(R(0));
    for (int i = 0; i < 3; ++i) {
        preal e = x[i] - xr[i];
        preal w = (real)C->perr[i] * e;
        l = PFMA(w, e, l);
        if (gx) gx[i] = R(2) * w;
    }
    for (int i = 0; i < 3; ++i) {
        preal e = x[3 + i] - xr[3 + i];
        preal w = (real)C->verr[i] * e;
        l = PFMA(w, e, l);
        if (gx) gx[3 + i] = R(2) * w;
    }
    for (int i = 0; i < 3; ++i) {
        preal e = x[10 + i] - xr[10 + i];
        preal w = (real)C->werr[i] * e;
        l = PFMA(w, e, l);
        if (gx) gx[10 + i] = R(2) * w;
    }
    preal qw = x[6], qx = x[7], qy = x[8], qz = x[9], rw = pbroadcast(xr[6]), rx = pbroadcast(xr[7]), ry = pbroadcast(xr[8]), rz = pbroadcast(xr[9]);
    preal ex = PFMA(rz, qy, PFMA(-ry, qz, PFMA(-rx, qw, rw * qx)));
    preal ey = PFMA(-rz, qx, PFMA(-ry, qw, PFMA(rx, qz, rw * qy)));
    preal ez = PFMA(-rz, qw, PFMA(ry, qx, PFMA(-rx, qy, rw * qz)));
    preal wxe = (real)C->qerr[0] * ex, wye = (real)C->qerr[1] * ey, wze = (real)C->qerr[2] * ez;
    l = PFMA(wxe, ex, l); l = PFMA(wye, ey, l); l = PFMA(wze, ez, l);
    if (gx) {
        preal a = R(2) * wxe, b = R(2) * wye, c = R(2) * wze;
        gx[6] = PFMA(-rz, c, PFMA(-ry, b, -rx * a));
        gx[7] = PFMA(ry, c, PFMA(-rz, b, rw * a));
        gx[8] = PFMA(-rx, c, PFMA(rw, b, rz * a));
        gx[9] = PFMA(rw, c, PFMA(rx, b, -ry * a));
    }
    /* state_constr, penalty form (SPEC.md §5.3; keys launch/iris_sitl_traj_mpc.yaml:16-29): bounded states in ascending index */
    for (int k = 0; k < C->num_state_constr; ++k) {
        int i = C->state_id[k];
        preal hi = x[i] - (real)C->state_hi[k], lo = (real)C->state_lo[k] - x[i];
        hi = PPOS(hi); lo = PPOS(lo);
        l = PFMA((real)C->state_w[k] * hi, hi, l);
        l = PFMA((real)C->state_w[k] * lo, lo, l);
        if (gx) gx[i] = PFMA(R(2) * (real)C->state_w[k], hi - lo, gx[i]);
    }
    return l;
}

/* VJP of step_fwd. L = adjoint wrt x_{t+1}; etabar_cost = direct d(cost)/d(eta).
 * Outputs: lam = adjoint wrt x_t; gu[m] = W1u^T abar1 (drift tile); gT = adjoint of Tz; gtau[3]. */
/* second half of step_vjp: from the adjoint of z (zb) and of the velocity (vbar) to the state adjoint (shared by the arithmetics of the MLP part) */
static inline void step_vjp_tail(const model_t* M, const stepaux_t* A, const preal* v, const preal* om, preal qw, preal qx, preal qy, preal qz, const preal* Rm,
                                 const preal* Fwb, const preal* zb, const preal* vbar, const preal* qtb, const preal* dqb, preal* omb, const preal* Lp, preal* lam) {
    (void)M;
    /* Rbar_ij = Fwb_i Fb_j + v_i zb_j */
    preal Rb[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rb[3 * i + j] = PFMA(v[i], zb[j], Fwb[i] * A->Fb[j]);
    /* dq = 0.5 q (x) (0, om) */
    preal qb[4];
    qb[0] = PFMA(R(0.5), PFMA(dqb[3], om[2], PFMA(dqb[2], om[1], dqb[1] * om[0])), qtb[0]);
    qb[1] = PFMA(R(0.5), PFMA(dqb[3], om[1], PFMA(-dqb[2], om[2], -(dqb[0] * om[0]))), qtb[1]);
    qb[2] = PFMA(R(0.5), PFMA(-dqb[3], om[0], PFMA(dqb[1], om[2], -(dqb[0] * om[1]))), qtb[2]);
    qb[3] = PFMA(R(0.5), PFMA(dqb[2], om[0], PFMA(-dqb[1], om[1], -(dqb[0] * om[2]))), qtb[3]);
    omb[0] = PFMA(R(0.5), PFMA(-dqb[3], qy, PFMA(dqb[2], qz, PFMA(dqb[1], qw, -(dqb[0] * qx)))), omb[0]);
    omb[1] = PFMA(R(0.5), PFMA(dqb[3], qx, PFMA(dqb[2], qw, PFMA(-dqb[1], qz, -(dqb[0] * qy)))), omb[1]);
    omb[2] = PFMA(R(0.5), PFMA(dqb[3], qw, PFMA(-dqb[2], qx, PFMA(dqb[1], qy, -(dqb[0] * qz)))), omb[2]);
    /* R(q) */
    preal s01 = Rb[1] + Rb[3], d10 = Rb[3] - Rb[1];
    preal s02 = Rb[2] + Rb[6], d02 = Rb[2] - Rb[6];
    preal s12 = Rb[5] + Rb[7], d21 = Rb[7] - Rb[5];
    qb[0] = PFMA(R(2), PFMA(qx, d21, PFMA(qy, d02, qz * d10)), qb[0]);
    qb[1] = PFMA(R(2), PFMA(qw, d21, PFMA(qz, s02, qy * s01)), PFMA(R(-4) * qx, Rb[4] + Rb[8], qb[1]));
    qb[2] = PFMA(R(2), PFMA(qz, s12, PFMA(qw, d02, qx * s01)), PFMA(R(-4) * qy, Rb[0] + Rb[8], qb[2]));
    qb[3] = PFMA(R(2), PFMA(qy, s12, PFMA(qx, s02, qw * d10)), PFMA(R(-4) * qz, Rb[0] + Rb[4], qb[3]));
    for (int i = 0; i < 3; ++i) { lam[i] = Lp[i]; lam[3 + i] = vbar[i]; lam[10 + i] = omb[i]; }
    for (int i = 0; i < 4; ++i) lam[6 + i] = qb[i];
}

static void step_vjp(const model_t* M, const preal* x, const preal* xi, real dt, const real* sdt,
                     const stepaux_t* A, const preal* L, preal etabar_cost,
                     preal* lam, preal* gu, preal* gT, preal* gtau) {
    const preal *v = x + 3, *q = x + 6, *om = x + 10;
    const preal *Lp = L, *Lv = L + 3, *Lq = L + 6, *Lo = L + 10;
    const preal* Rm = A->Rm;
    preal qw = q[0], qx = q[1], qy = q[2], qz = q[3];
    /* eta adjoint */
    preal eb = etabar_cost;
    for (int i = 0; i < 3; ++i) eb = PFMA(Lv[i] * sdt[i], xi[i], eb);
    for (int i = 0; i < 3; ++i) eb = PFMA(Lo[i] * sdt[3 + i], xi[3 + i], eb);
    preal ebraw = eb * (A->eta * (R(1) - A->eta));
    /* quaternion normalisation */
    preal dotq = PFMA(A->qn[3], Lq[3], PFMA(A->qn[2], Lq[2], PFMA(A->qn[1], Lq[1], A->qn[0] * Lq[0])));
    preal qtb[4], dqb[4];
    for (int i = 0; i < 4; ++i) { qtb[i] = A->rn * PFMA(-A->qn[i], dotq, Lq[i]); dqb[i] = qtb[i] * dt; }
    /* angular acceleration */
    preal taub_b[3], crb[3];
    for (int i = 0; i < 3; ++i) { taub_b[i] = (Lo[i] * dt) * M->iJ[i]; crb[i] = -taub_b[i]; }
    /* cr = om x Jom : om_bar += Jom x crb ; Jom_bar = crb x om */
    preal omb[3], Jb[3];
    omb[0] = Lo[0] + PFMA(A->Jom[1], crb[2], -(A->Jom[2] * crb[1]));
    omb[1] = Lo[1] + PFMA(A->Jom[2], crb[0], -(A->Jom[0] * crb[2]));
    omb[2] = Lo[2] + PFMA(A->Jom[0], crb[1], -(A->Jom[1] * crb[0]));
    Jb[0] = PFMA(crb[1], om[2], -(crb[2] * om[1]));
    Jb[1] = PFMA(crb[2], om[0], -(crb[0] * om[2]));
    Jb[2] = PFMA(crb[0], om[1], -(crb[1] * om[0]));
    for (int i = 0; i < 3; ++i) omb[i] = PFMA(M->J[i], Jb[i], omb[i]);
    /* linear acceleration: acc = R Fb inv_mass - g e3 */
    preal Fwb[3], Fbb[3];
    for (int i = 0; i < 3; ++i) Fwb[i] = (Lv[i] * dt) * M->inv_mass;
    for (int j = 0; j < 3; ++j) Fbb[j] = PFMA(Rm[6 + j], Fwb[2], PFMA(Rm[3 + j], Fwb[1], Rm[j] * Fwb[0]));
    /* MLP output adjoints */
    preal ob[6];
    for (int i = 0; i < 3; ++i) { ob[i] = M->sF[i] * Fbb[i]; ob[3 + i] = M->sT[i] * taub_b[i]; }
    *gT = Fbb[2];
    for (int i = 0; i < 3; ++i) gtau[i] = taub_b[i];
    /* MLP VJP (SPEC.md §5.4) */
    preal a2b[HID], a1d[HID], a1n[HID];
#ifdef ORC_MFMA16
    float adj_inv = 1.0f;
    if (M->adjmp) {
        /* SPEC.md §10e: one power of two per particle brings the output adjoints to 2^eoff (the largest of the seven in [2^(eoff-1), 2^eoff)): every product
         * below is exact in the scale, and the three contractions can take binary16 limbs. The forward weights (-2 W3, -2 w3n) meet the adjoints times -2 s. */
        float mx = fabsf(ebraw);
        for (int i = 0; i < 6; ++i) mx = fmaxf(mx, fabsf(ob[i]));
        int e = 0;
        if (mx > 0.0f && isfinite(mx)) (void)frexpf(mx, &e);
        if (e < -100) e = -100;
        const float s2 = -2.0f * ldexpf(1.0f, M->adj_eoff - e);
        adj_inv = ldexpf(1.0f, e - M->adj_eoff);
        for (int i = 0; i < 6; ++i) ob[i] = ob[i] * s2;
        ebraw = ebraw * s2;
        float ov[HID], hb3[HID];
        for (int k = 0; k < HID; ++k) ov[k] = 0.0f;
        for (int i = 0; i < 6; ++i) ov[slot_unit(0, i)] = ob[i];               /* k slot i of K-half 0 */
        x3_contract(1, M->y2m[3], M->y2x[3], M->y2mT[3], M->y2xT[3], ov, NULL, hb3);          /* (only K-half 0 holds products: four instructions on the GPU) */
        for (int k = 0; k < HID; ++k) a2b[k] = hb3[k] * DACT(M, A->h2[k]);
        float hb2[HID], zacc[HID], zacc2[HID];
        x3_contract(1, M->y2m[0], M->y2x[0], M->y2mT[0], M->y2xT[0], a2b, NULL, hb2);
        for (int k = 0; k < HID; ++k) { a1d[k] = hb2[k] * DACT(M, A->h1d[k]); a1n[k] = (M->w3n[k] * ebraw) * DACT(M, A->h1n[k]); }
        x3_contract(1, M->y2m[1], M->y2x[1], M->y2mT[1], M->y2xT[1], a1n, NULL, zacc);
        x3_contract(1, M->y2m[2], M->y2x[2], M->y2mT[2], M->y2xT[2], a1d, NULL, zacc2);
        for (int r = 0; r < 8; ++r) zacc2[r] = zacc2[r] + zacc[r];         /* rows 0..7 (accumulator registers 0..3 of either lane half): the density tile's rows, one float32 addition each */
        preal zb2[NN];
        for (int k = 0; k < NN; ++k) zb2[k] = zacc2[k] * adj_inv;
        for (int j = 0; j < M->m; ++j) gu[j] = zacc2[NN + j] * adj_inv;
        for (int i = 0; i < 3; ++i) omb[i] = omb[i] + zb2[3 + i];
        preal vbar2[3];
        for (int i = 0; i < 3; ++i) {
            preal Rvb = PFMA(Rm[3 * i + 2], zb2[2], PFMA(Rm[3 * i + 1], zb2[1], Rm[3 * i] * zb2[0]));
            vbar2[i] = PFMA(Lp[i], dt, Lv[i]) + Rvb;
        }
        step_vjp_tail(M, A, v, om, qw, qx, qy, qz, Rm, Fwb, zb2, vbar2, qtb, dqb, omb, Lp, lam);
        return;
    }
#endif
    for (int k = 0; k < HID; ++k) {
        preal hb = pbroadcast(R(0));
        for (int i = 0; i < 6; ++i) hb = PFMA(M->vW3[i][k], ob[i], hb);
        a2b[k] = hb * DACT(M, A->h2[k]);
    }
#ifdef ORC_MFMA16
    float hb_x3[HID];
    if (M->f16 == 2) x3_contract(0, M->x2m[1], M->x2x[1], M->x2mT[1], M->x2xT[1], a2b, NULL, hb_x3);       /* SPEC.md §9b: W2^T abar2 as the three-limb split */
#endif
    for (int k = 0; k < HID; ++k) {
        preal hb = pbroadcast(R(0));
#ifdef ORC_MFMA16
        if (M->f16 == 2) hb = hb_x3[k]; else
#endif
        for (int r = 0; r < 16; ++r) for (int h = 0; h < 2; ++h) { int i = rowmap(r, h); hb = PFMA(M->vW2[i][k], a2b[i], hb); }
        a1d[k] = hb * DACT(M, A->h1d[k]);
        a1n[k] = (M->vw3n[k] * ebraw) * DACT(M, A->h1n[k]);
    }
    preal zb[NN];
    for (int k = 0; k < NN; ++k) {
        preal P0 = pbroadcast(R(0)), P1 = pbroadcast(R(0));
        for (int r = 0; r < 16; ++r) { P0 = PFMA(M->vW1z[HID + rowmap(r, 0)][k], a1n[rowmap(r, 0)], P0); P1 = PFMA(M->vW1z[HID + rowmap(r, 1)][k], a1n[rowmap(r, 1)], P1); }
        for (int r = 0; r < 16; ++r) { P0 = PFMA(M->vW1z[rowmap(r, 0)][k], a1d[rowmap(r, 0)], P0); P1 = PFMA(M->vW1z[rowmap(r, 1)][k], a1d[rowmap(r, 1)], P1); }
        zb[k] = P0 + P1;
    }
    for (int j = 0; j < M->m; ++j) {
        preal P0 = pbroadcast(R(0)), P1 = pbroadcast(R(0));
        for (int r = 0; r < 16; ++r) { P0 = PFMA(M->vW1u[rowmap(r, 0)][j], a1d[rowmap(r, 0)], P0); P1 = PFMA(M->vW1u[rowmap(r, 1)][j], a1d[rowmap(r, 1)], P1); }
        gu[j] = P0 + P1;
    }
    for (int i = 0; i < 3; ++i) omb[i] = omb[i] + zb[3 + i];
    /* vb = R^T v */
    preal vbar[3];
    for (int i = 0; i < 3; ++i) {
        preal Rvb = PFMA(Rm[3 * i + 2], zb[2], PFMA(Rm[3 * i + 1], zb[1], Rm[3 * i] * zb[0]));
        vbar[i] = PFMA(Lp[i], dt, Lv[i]) + Rvb;
    }
    step_vjp_tail(M, A, v, om, qw, qx, qy, qz, Rm, Fwb, zb, vbar, qtb, dqb, omb, Lp, lam);
}

/* ------------------------------------------------------------------------------------------- */
/* reductions (SPEC.md §6)                                                                     */
/* ------------------------------------------------------------------------------------------- */
/* particle reduction: groups of 32 (xor butterfly 16,8,4,2,1), group g -> slot g%4 (sequential),
 * total = ((S0+S1)+S2)+S3. vals has P entries. */
static real preduce(const real* vals, int P) {
    real S[NSLOT] = {0, 0, 0, 0};
    int G = (P + 31) / 32;
    for (int g = 0; g < G; ++g) {
        real v[32];
        for (int j = 0; j < 32; ++j) v[j] = (g * 32 + j < P) ? vals[g * 32 + j] : R(0);
        for (int s = 16; s >= 1; s >>= 1) {
            real w[32];
            for (int j = 0; j < 32; ++j) w[j] = v[j] + v[j ^ s];
            memcpy(v, w, sizeof v);
        }
        S[g % NSLOT] = S[g % NSLOT] + v[0];
    }
    return ((S[0] + S[1]) + S[2]) + S[3];
}
/* block dot product over N elements with 256 lanes: lane i chains e = i, i+256, ...; xor butterfly
 * 32,16,8,4,2,1 inside each 64-lane wave; total = ((w0+w1)+w2)+w3 */
static real dot256(const real* a, const real* b, int N) {
    real v[256];
    for (int i = 0; i < 256; ++i) {
        real acc = 0;
        for (int e = i; e < N; e += 256) acc = FMA(a[e], b ? b[e] : R(1), acc);
        v[i] = acc;
    }
    for (int w = 0; w < 4; ++w)
        for (int s = 32; s >= 1; s >>= 1) {
            real t[64];
            for (int j = 0; j < 64; ++j) t[j] = v[w * 64 + j] + v[w * 64 + (j ^ s)];
            memcpy(v + w * 64, t, sizeof t);
        }
    return ((v[0] + v[64]) + v[128]) + v[192];
}

/* lanes of one block of VL particles <-> strided scalar storage (n = live lanes of the block; dead lanes load 0) */
static inline preal pload(const real* base, size_t stride, int n) {
#ifdef ORC_VEC
    preal v = pbroadcast(R(0));
    for (int l = 0; l < VL; ++l) if (l < n) v[l] = base[(size_t)l * stride];
    return v;
#else
    (void)stride; (void)n;
    return base[0];
#endif
}
static inline void pstore(real* base, size_t stride, int n, preal v) {
#ifdef ORC_VEC
    for (int l = 0; l < VL; ++l) if (l < n) base[(size_t)l * stride] = v[l];
#else
    (void)stride; (void)n;
    base[0] = v;
#endif
}

/* ------------------------------------------------------------------------------------------- */
/* problem context                                                                             */
/* ------------------------------------------------------------------------------------------- */
typedef struct {
    const sdempc_cfg* C;
    model_t M;
    int H, P, m;
    real* dt;    /* [H] */
    real* sdt;   /* [H][6] sigma_i * sqrt(dt) */
    real* disc;  /* [H+1] */
    real invP;
    /* per-context workspace (no allocation inside rollouts / gradients) */
    void* wsU; real *wsJp, *wsTraj, *wsQ, *wsG, *wsCe, *wsDw, *wsCol, *wsXs;
} ctx_t;

static int ctx_init(ctx_t* X, const sdempc_cfg* C, const void* blob) {
    if (!C || C->struct_size != (int32_t)sizeof(sdempc_cfg)) return SDEMPC_EINVAL;
#ifdef ORC_VEC
    if (C->mlp_dtype != 0) return SDEMPC_EINVAL;   /* the timing build has the f32 arithmetic only */
#endif
    if (parse_blob(blob, &X->M, C->mlp_dtype, C->math_mode != 0)) return SDEMPC_EBLOB;
    if (C->math_mode) {
#ifdef ORC_MFMA16
        if (!orc_transc_ready()) return SDEMPC_EINVAL;      /* tests/orc.py maps the instruction tables (orc_transc_open) before a fast-mode oracle is made */
#else
        return SDEMPC_EINVAL;                               /* the float64 and the timing builds have the SPEC §3 functions only */
#endif
    }
    X->C = C; X->H = C->horizon; X->P = C->num_particles; X->m = C->num_motors;
    if (X->H < 1 || X->P < 1 || X->m != X->M.m) return SDEMPC_EINVAL;
    X->dt = (real*)malloc(sizeof(real) * X->H);
    X->sdt = (real*)malloc(sizeof(real) * X->H * NN);
    X->disc = (real*)malloc(sizeof(real) * (X->H + 1));
    for (int t = 0; t < X->H; ++t) {
        float dtf = C->time_steps[t];
        float sq = sqrtf(dtf);
        X->dt[t] = dtf;
        for (int i = 0; i < NN; ++i) X->sdt[t * NN + i] = (real)((float)X->M.sigma[i] * sq);
    }
    /* stage weights: discount^t / H (cost is the horizon mean, SPEC.md §5.3) */
    float d = 1.0f / (float)X->H;
    for (int t = 0; t <= X->H; ++t) { X->disc[t] = d; d = d * C->discount; }
    X->invP = (real)(1.0f / (float)X->P);
    {
        size_t H = X->H, P = X->P, m = X->m;
        X->wsU = malloc(sizeof(ustep_t) * H);
        X->wsJp = (real*)malloc(sizeof(real) * P);
        X->wsTraj = (real*)malloc(sizeof(real) * P * (H + 1) * NX);
        X->wsQ = (real*)malloc(sizeof(real) * H * (m + 4) * P);
        X->wsG = (real*)malloc(sizeof(real) * H * m);
        X->wsCe = (real*)malloc(sizeof(real) * H * m);
        X->wsDw = (real*)malloc(sizeof(real) * H * m);
        X->wsCol = (real*)malloc(sizeof(real) * P);
        X->wsXs = (real*)malloc(sizeof(real) * P * (H + 1) * NX);
    }
    return 0;
}
static void ctx_free(ctx_t* X) {
    free(X->dt); free(X->sdt); free(X->disc);
    free(X->wsU); free(X->wsJp); free(X->wsTraj); free(X->wsQ); free(X->wsG); free(X->wsCe); free(X->wsDw); free(X->wsCol); free(X->wsXs);
}

/* control cost and its gradient (SPEC.md §5.5): element e = t*m + j.
 * keys uref/uerr/u_slew_coeff/u_slew_constr(_coeff): launch/iris_sitl_posctrl_mpc.yaml:30-41 */
static real ucost(const ctx_t* X, const real* u, real* gcu) {
    const sdempc_cfg* C = X->C;
    int H = X->H, m = X->m, N = H * m;
    real* ce = X->wsCe;
    real* dw = X->wsDw;
    for (int t = 0; t < H; ++t) for (int j = 0; j < m; ++j) {
        int e = t * m + j;
        real du = u[e] - (real)C->uref[j];
        real a = ((real)C->uerr * du) * du;
        dw[e] = 0;
        if (t >= 1) {
            real ds = u[e] - u[e - m];
            a = FMA((real)C->u_slew_coeff * ds, ds, a);
            real d = (R(2) * (real)C->u_slew_coeff) * ds;
            if (C->has_slew_constr) {
                real hi = ds - (real)C->u_slew_hi[j]; if (hi < 0) hi = 0;
                real lo = (real)C->u_slew_lo[j] - ds; if (lo < 0) lo = 0;
                a = FMA((real)C->u_slew_constr_coeff * hi, hi, a);
                a = FMA((real)C->u_slew_constr_coeff * lo, lo, a);
                d = FMA(R(2) * (real)C->u_slew_constr_coeff, hi - lo, d);
            }
            dw[e] = d;
        }
        ce[e] = X->disc[t] * a;
    }
    if (gcu)
        for (int t = 0; t < H; ++t) for (int j = 0; j < m; ++j) {
            int e = t * m + j;
            real du = u[e] - (real)C->uref[j];
            real g = X->disc[t] * FMA(R(2) * (real)C->uerr, du, dw[e]);
            if (t + 1 < H) g = FMA(-X->disc[t + 1], dw[e + m], g);
            gcu[e] = g;
        }
    return dot256(ce, NULL, N);
}

/* Particles are independent between the reductions (each writes its own rows of traj / Jp / Q; the sums of SPEC.md §6 are formed afterwards in their
 * fixed order), so the particle loops below may be spread over threads without changing a bit: NAME(set_threads)(n), default 1 (bench.py's verifier
 * runs one solve per thread instead; the golden generators of the long configurations and a lone solve use n = cores). Needs -fopenmp (oracle/Makefile);
 * without it the pragmas are ignored and the loops are serial. */
static int g_threads = 1;
void NAME(set_threads)(int n) { g_threads = n < 1 ? 1 : n; }

/* rollout: expected cost; optional traj [P][H+1][13], xmean [H+1][13] (SPEC.md §5.3, §7) */
static real rollout(const ctx_t* X, const real* x0, const real* u, const real* xref, const real* noise,
                    real* traj, real* xmean) {
    int H = X->H, P = X->P, m = X->m;
    ustep_t* U = (ustep_t*)X->wsU;
    for (int t = 0; t < H; ++t) ustep_eval(&X->M, u + t * m, &U[t]);
    real* Jp = X->wsJp;
    real* xs = xmean ? X->wsXs : NULL;
    const size_t ps = (size_t)(H + 1) * NX;                 /* particle stride of traj / xs */
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
    for (int p = 0; p < P; p += VL) {                        /* one block of VL particles (VL = 1 in the checker builds) */
        const int n = P - p < VL ? P - p : VL;
        stepaux_t A;
        preal x[NX], xn[NX], xi[NN];
        for (int i = 0; i < NX; ++i) x[i] = pbroadcast(x0[i]);
        real* tp = traj ? traj + (size_t)p * ps : NULL;
        real* sp = xs ? xs + (size_t)p * ps : NULL;
        if (tp) for (int i = 0; i < NX; ++i) pstore(tp + i, ps, n, x[i]);
        if (sp) for (int i = 0; i < NX; ++i) pstore(sp + i, ps, n, x[i]);
        preal J = pbroadcast(R(0));
        for (int t = 0; t < H; ++t) {
            for (int i = 0; i < NN; ++i) xi[i] = pload(noise + ((size_t)p * H + t) * NN + i, (size_t)H * NN, n);
            step_fwd(&X->M, &U[t], x, xi, X->dt[t], X->sdt + t * NN, xn, &A);
            preal l = stage_cost(X->C, xn, xref + (t + 1) * NX, NULL);
            l = PFMA((real)X->C->res_mult * A.eta, A.eta, l);
            J = PFMA(X->disc[t], l, J);
            memcpy(x, xn, sizeof x);
            if (tp) for (int i = 0; i < NX; ++i) pstore(tp + (t + 1) * NX + i, ps, n, x[i]);
            if (sp) for (int i = 0; i < NX; ++i) pstore(sp + (t + 1) * NX + i, ps, n, x[i]);
        }
        pstore(Jp + p, 1, n, J);
    }
    real tot = preduce(Jp, P);
    real cu = ucost(X, u, NULL);
    if (xmean) {
        real* col = X->wsCol;
        for (int t = 0; t <= H; ++t) for (int i = 0; i < NX; ++i) {
            for (int p = 0; p < P; ++p) col[p] = xs[((size_t)p * (H + 1) + t) * NX + i];
            xmean[t * NX + i] = preduce(col, P) * X->invP;
        }
    }
    return FMA(tot, X->invP, cu);
}

/* cost + gradient wrt u by the adjoint sweep (SPEC.md §5.4, §6) */
static real cost_grad(const ctx_t* X, const real* x0, const real* u, const real* xref, const real* noise, real* grad) {
    int H = X->H, P = X->P, m = X->m;
    const model_t* M = &X->M;
    ustep_t* U = (ustep_t*)X->wsU;
    for (int t = 0; t < H; ++t) ustep_eval(M, u + t * m, &U[t]);
    real* traj = X->wsTraj;
    real* Jp = X->wsJp;
    /* per-particle, per-step adjoint outputs: [H][m+4][P] */
    int nq = m + 4;
    real* Q = X->wsQ;
    const size_t ps = (size_t)(H + 1) * NX;                 /* particle stride of traj */
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
    for (int p = 0; p < P; p += VL) {                        /* one block of VL particles (VL = 1 in the checker builds) */
        const int n = P - p < VL ? P - p : VL;
        stepaux_t A;
        real* tp = traj + (size_t)p * ps;
        preal x[NX], xn[NX], xi[NN];
        for (int i = 0; i < NX; ++i) { x[i] = pbroadcast(x0[i]); pstore(tp + i, ps, n, x[i]); }
        preal J = pbroadcast(R(0));
        for (int t = 0; t < H; ++t) {
            for (int i = 0; i < NN; ++i) xi[i] = pload(noise + ((size_t)p * H + t) * NN + i, (size_t)H * NN, n);
            step_fwd(M, &U[t], x, xi, X->dt[t], X->sdt + t * NN, xn, &A);
            preal l = stage_cost(X->C, xn, xref + (t + 1) * NX, NULL);
            l = PFMA((real)X->C->res_mult * A.eta, A.eta, l);
            J = PFMA(X->disc[t], l, J);
            memcpy(x, xn, sizeof x);
            for (int i = 0; i < NX; ++i) pstore(tp + (t + 1) * NX + i, ps, n, x[i]);
        }
        pstore(Jp + p, 1, n, J);
        preal lam[NX];
        for (int i = 0; i < NX; ++i) lam[i] = pbroadcast(R(0));
        for (int t = H - 1; t >= 0; --t) {
            preal xt[NX], x1[NX], gx[NX], L[NX], lamn[NX], gu[MAXM], gT, gtau[3];
            for (int i = 0; i < NX; ++i) { xt[i] = pload(tp + t * NX + i, ps, n); x1[i] = pload(tp + (t + 1) * NX + i, ps, n); }
            for (int i = 0; i < NN; ++i) xi[i] = pload(noise + ((size_t)p * H + t) * NN + i, (size_t)H * NN, n);
            step_fwd(M, &U[t], xt, xi, X->dt[t], X->sdt + t * NN, xn, &A);
            stage_cost(X->C, x1, xref + (t + 1) * NX, gx);
            for (int i = 0; i < NX; ++i) L[i] = PFMA(X->disc[t], gx[i], lam[i]);
            preal ebc = X->disc[t] * ((R(2) * (real)X->C->res_mult) * A.eta);
            step_vjp(M, xt, xi, X->dt[t], X->sdt + t * NN, &A, L, ebc, lamn, gu, &gT, gtau);
            memcpy(lam, lamn, sizeof lam);
            for (int j = 0; j < m; ++j) pstore(Q + ((size_t)t * nq + j) * P + p, 1, n, gu[j]);
            pstore(Q + ((size_t)t * nq + m) * P + p, 1, n, gT);
            for (int i = 0; i < 3; ++i) pstore(Q + ((size_t)t * nq + m + 1 + i) * P + p, 1, n, gtau[i]);
        }
    }
    real tot = preduce(Jp, P);
    real* gcu = X->wsG;
    real cu = ucost(X, u, gcu);
    for (int t = 0; t < H; ++t) {
        real S[MAXM + 4];
        for (int k = 0; k < nq; ++k) S[k] = preduce(Q + ((size_t)t * nq + k) * P, P);
        for (int j = 0; j < m; ++j) {
            real a = S[j];
            a = FMA(S[m], U[t].dT[j], a);
            a = FMA(S[m + 1], M->ry[j] * U[t].dT[j], a);
            a = FMA(S[m + 2], -(M->rx[j] * U[t].dT[j]), a);
            a = FMA(S[m + 3], U[t].dM[j], a);
            grad[t * m + j] = FMA(a, X->invP, gcu[t * m + j]);
        }
    }
    return FMA(tot, X->invP, cu);
}

/* ------------------------------------------------------------------------------------------- */
/* accelerated proximal gradient with Armijo backtracking (SPEC.md §8)                         */
/* = m_mpc (sde_control.py:713-719,400-416); knobs apg_mpc: launch/iris_sitl_traj_mpc.yaml:55-85; */
/* telemetry read at sde_control.py:444-450                                                    */
/* ------------------------------------------------------------------------------------------- */
/* SPEC.md §3.6: median of three = fmin(fmax(v,lo),hi); a NaN maps to lo and -0 to +0 when lo = +0, as v_med3_f32 does */
static inline real clampr(real v, real lo, real hi) { return !(v > lo) ? lo : (v > hi ? hi : v); }

static void solve(const ctx_t* X, const real* x0, const real* xref, const real* noise,
                  const real* u_init, real s_in, real* uopt, real* xevol, real* info, real* trace, int trace_cap) {
    const sdempc_cfg* C = X->C;
    int H = X->H, m = X->m, N = H * m;
    real *xk = malloc(sizeof(real) * N), *yk = malloc(sizeof(real) * N), *xn = malloc(sizeof(real) * N),
         *g = malloc(sizeof(real) * N), *ub = malloc(sizeof(real) * N), *d1 = malloc(sizeof(real) * N), *d2 = malloc(sizeof(real) * N);
    /* momentum table beta[kr] (host float32 arithmetic) */
    int nb = C->max_iter + 2;
    real* beta = malloc(sizeof(real) * nb);
    for (int i = 0; i < nb; ++i) {
        float b = (i == 0) ? C->beta_init : (float)(i + 1) / (float)(i + 4);
        if (i > 0 && C->use_moment_scale) b = C->moment_scale * b;
        beta[i] = b;
    }
    for (int e = 0; e < N; ++e) { xk[e] = clampr(u_init[e], (real)C->u_lo[e % m], (real)C->u_hi[e % m]); yk[e] = xk[e]; ub[e] = xk[e]; }
    real c_init = rollout(X, x0, xk, xref, noise, NULL, NULL);
    real c_x = c_init, s = s_in, gsq = 0, sum_ls = 0, sum_s = 0;
    int kr = 0, noimp = 0, nit = 0, nls_tot = 0, plain = 1; /* plain: yk == xk (no momentum in yk) */
    for (int k = 0; k < C->max_iter; ++k) {
        real c_y = cost_grad(X, x0, yk, xref, noise, g);
        gsq = dot256(g, g, N);
        if (!(gsq < (real)INFINITY)) break; /* SPEC.md §8 non-finite guard: no step is taken on a NaN/inf gradient */
        real c_n = 0;
        int nls = 0;
        if (C->ls_maxls > 0) {
            if (k > 0 && C->ls_reset_option == 1) s = s * (real)C->ls_increase_factor;
            if (s > (real)C->ls_max_stepsize) s = (real)C->ls_max_stepsize;
            for (int j = 0; j < C->ls_maxls; ++j) {
                for (int e = 0; e < N; ++e) { xn[e] = clampr(FMA(-s, g[e], yk[e]), (real)C->u_lo[e % m], (real)C->u_hi[e % m]); d1[e] = xn[e] - yk[e]; }
                c_n = rollout(X, x0, xn, xref, noise, NULL, NULL);
                real gd = dot256(g, d1, N);
                nls = j + 1;
                if (c_n <= FMA((real)C->ls_coef, gd, c_y)) break;
                if (j < C->ls_maxls - 1) s = s * (real)C->ls_decrease_factor;
            }
        } else {
            s = (real)C->stepsize;
            for (int e = 0; e < N; ++e) xn[e] = clampr(FMA(-s, g[e], yk[e]), (real)C->u_lo[e % m], (real)C->u_hi[e % m]);
            c_n = rollout(X, x0, xn, xref, noise, NULL, NULL);
            nls = 1;
        }
        sum_ls = sum_ls + (real)nls; sum_s = sum_s + s; nit = k + 1; nls_tot += nls;
        if (trace && k < trace_cap) { trace[4 * k] = c_y; trace[4 * k + 1] = c_n; trace[4 * k + 2] = s; trace[4 * k + 3] = (real)nls; }
        int stop = (FABS(c_n - c_x) <= FMA((real)C->rtol, FABS(c_x), (real)C->atol));
        if (c_n < c_x) {
            /* monotone acceptance: the new iterate lowers the cost of the current one */
            for (int e = 0; e < N; ++e) { d1[e] = yk[e] - xn[e]; d2[e] = xn[e] - xk[e]; }
            real rs = dot256(d1, d2, N);
            if (rs > 0) { /* adaptive restart, gradient scheme */
                kr = 0; plain = 1;
                for (int e = 0; e < N; ++e) yk[e] = xn[e];
            } else {
                real b = beta[kr];
                for (int e = 0; e < N; ++e) yk[e] = clampr(FMA(b, d2[e], xn[e]), (real)C->u_lo[e % m], (real)C->u_hi[e % m]);
                kr = kr + 1; plain = 0;
            }
            c_x = c_n; noimp = 0;
            memcpy(xk, xn, sizeof(real) * N);
        } else {
            /* no decrease: drop the momentum and retry from xk with the (already shrunk) step size;
             * only a failed plain gradient step may signal convergence */
            if (!plain) stop = 0;
            kr = 0; plain = 1;
            memcpy(yk, xk, sizeof(real) * N);
            noimp = noimp + 1;
        }
        if (noimp >= C->max_no_improvement_iter) stop = 1;
        if (stop) break;
    }
    real c_best = c_x;
    memcpy(ub, xk, sizeof(real) * N);
    memcpy(uopt, ub, sizeof(real) * N);
    rollout(X, x0, ub, xref, noise, NULL, xevol);
    info[0] = nit ? (real)((float)sum_ls / (float)nit) : 0;
    info[1] = s;
    info[2] = (real)nit;
    info[3] = gsq;
    info[4] = nit ? (real)((float)sum_s / (float)nit) : 0;
    info[5] = c_init;
    info[6] = c_best;
    info[7] = (real)nls_tot;
    free(xk); free(yk); free(xn); free(g); free(ub); free(d1); free(d2); free(beta);
}

/* ------------------------------------------------------------------------------------------- */
/* exported API (float I/O in both builds)                                                     */
/* ------------------------------------------------------------------------------------------- */
static real* to_real(const float* a, size_t n) { real* r = malloc(sizeof(real) * n); for (size_t i = 0; i < n; ++i) r[i] = a[i]; return r; }

int NAME(rollout)(const sdempc_cfg* C, const void* blob, const float* x0, const float* u, const float* xref,
                  const float* noise, double* cost, float* traj, float* xmean) {
    ctx_t X; int rc = ctx_init(&X, C, blob); if (rc) return rc;
    int H = X.H, P = X.P, m = X.m;
    real *rx0 = to_real(x0, NX), *ru = to_real(u, H * m), *rxr = to_real(xref, (H + 1) * NX), *rn = to_real(noise, (size_t)P * H * NN);
    real* rt = traj ? malloc(sizeof(real) * (size_t)P * (H + 1) * NX) : NULL;
    real* rm = xmean ? malloc(sizeof(real) * (H + 1) * NX) : NULL;
    *cost = (double)rollout(&X, rx0, ru, rxr, rn, rt, rm);
    if (rt) { for (size_t i = 0; i < (size_t)P * (H + 1) * NX; ++i) traj[i] = (float)rt[i]; free(rt); }
    if (rm) { for (int i = 0; i < (H + 1) * NX; ++i) xmean[i] = (float)rm[i]; free(rm); }
    free(rx0); free(ru); free(rxr); free(rn); ctx_free(&X);
    return 0;
}

int NAME(grad)(const sdempc_cfg* C, const void* blob, const float* x0, const float* u, const float* xref,
               const float* noise, double* cost, double* grad) {
    ctx_t X; int rc = ctx_init(&X, C, blob); if (rc) return rc;
    int H = X.H, P = X.P, m = X.m;
    real *rx0 = to_real(x0, NX), *ru = to_real(u, H * m), *rxr = to_real(xref, (H + 1) * NX), *rn = to_real(noise, (size_t)P * H * NN);
    real* g = malloc(sizeof(real) * H * m);
    *cost = (double)cost_grad(&X, rx0, ru, rxr, rn, g);
    for (int i = 0; i < H * m; ++i) grad[i] = (double)g[i];
    free(g); free(rx0); free(ru); free(rxr); free(rn); ctx_free(&X);
    return 0;
}

#ifndef ORC_VEC
/* cost as a function of double-precision u (finite-difference tests, ORC_DOUBLE build only keeps the
 * precision; in the float build u is rounded to float) */
int NAME(cost_du)(const sdempc_cfg* C, const void* blob, const float* x0, const double* u, const float* xref,
                  const float* noise, double* cost) {
    ctx_t X; int rc = ctx_init(&X, C, blob); if (rc) return rc;
    int H = X.H, P = X.P, m = X.m;
    real *rx0 = to_real(x0, NX), *rxr = to_real(xref, (H + 1) * NX), *rn = to_real(noise, (size_t)P * H * NN);
    real* ru = malloc(sizeof(real) * H * m);
    for (int i = 0; i < H * m; ++i) ru[i] = (real)u[i];
    *cost = (double)rollout(&X, rx0, ru, rxr, rn, NULL, NULL);
    free(ru); free(rx0); free(rxr); free(rn); ctx_free(&X);
    return 0;
}

#endif /* !ORC_VEC */

int NAME(solve)(const sdempc_cfg* C, const void* blob, const float* x0, const float* xref, const float* noise,
                const float* u_init, float stepsize_in, float* uopt, float* xevol, float* info8,
                float* trace /* [trace_cap][4] or NULL */, int trace_cap) {
    ctx_t X; int rc = ctx_init(&X, C, blob); if (rc) return rc;
    int H = X.H, P = X.P, m = X.m;
    real *rx0 = to_real(x0, NX), *rxr = to_real(xref, (H + 1) * NX), *rn = to_real(noise, (size_t)P * H * NN), *rui = to_real(u_init, H * m);
    real *ruo = malloc(sizeof(real) * H * m), *rxe = malloc(sizeof(real) * (H + 1) * NX), rinfo[8];
    real* rtr = trace ? calloc((size_t)trace_cap * 4, sizeof(real)) : NULL;
    solve(&X, rx0, rxr, rn, rui, (real)stepsize_in, ruo, rxe, rinfo, rtr, trace_cap);
    for (int i = 0; i < H * m; ++i) uopt[i] = (float)ruo[i];
    for (int i = 0; i < (H + 1) * NX; ++i) xevol[i] = (float)rxe[i];
    for (int i = 0; i < 8; ++i) info8[i] = (float)rinfo[i];
    if (trace) { for (int i = 0; i < trace_cap * 4; ++i) trace[i] = (float)rtr[i]; free(rtr); }
    free(rx0); free(rxr); free(rn); free(rui); free(ruo); free(rxe); ctx_free(&X);
    return 0;
}

/* batch of independent solves, one after another on the calling thread (cpu_baseline timing) */
int NAME(solve_batch)(const sdempc_cfg* C, const void* blob, int B, const float* x0, const float* xref, const float* noise,
                      const float* u_init, const float* stepsize_in, float* uopt, float* xevol, float* info8) {
    int H = C->horizon, P = C->num_particles, m = C->num_motors;
    for (int b = 0; b < B; ++b) {
        int rc = NAME(solve)(C, blob, x0 + (size_t)b * NX, xref + (size_t)b * (H + 1) * NX, noise + (size_t)b * P * H * NN,
                             u_init + (size_t)b * H * m, stepsize_in[b], uopt + (size_t)b * H * m, xevol + (size_t)b * (H + 1) * NX,
                             info8 + (size_t)b * 8, NULL, 0);
        if (rc) return rc;
    }
    return 0;
}

#ifndef ORC_VEC
/* fp16 round-toward-zero quantiser, exposed for unit tests */
double NAME(f16_rtz_value)(double x) { return (double)f16_rtz((real)x); }

/* single EM step and its VJP, exposed for unit tests */
int NAME(step)(const sdempc_cfg* C, const void* blob, const float* x, const float* u, const float* xi, int t, float* xn, float* eta) {
    ctx_t X; int rc = ctx_init(&X, C, blob); if (rc) return rc;
    real *rx = to_real(x, NX), *ru = to_real(u, X.m), *rxi = to_real(xi, NN), out[NX];
    ustep_t U; stepaux_t A;
    ustep_eval(&X.M, ru, &U);
    step_fwd(&X.M, &U, rx, rxi, X.dt[t], X.sdt + t * NN, out, &A);
    for (int i = 0; i < NX; ++i) xn[i] = (float)out[i];
    *eta = (float)A.eta;
    free(rx); free(ru); free(rxi); ctx_free(&X);
    return 0;
}
#endif /* !ORC_VEC */
