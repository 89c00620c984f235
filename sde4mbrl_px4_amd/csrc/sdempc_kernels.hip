// sdempc_kernels.hip — hand-written gfx950 (CDNA4) kernels for the MPC inner loop.
//
// Path replaced (reference): the body of m_mpc / m_reset that sde4mbrl_px4 obtains from
// load_mpc_from_cfgfile (sde4mbrl_px4/mpc_controller/sde_control.py:685) and calls per tick
// (sde_control.py:345-350,400-416). Arithmetic spec: SPEC.md (this repo); rows A3-A7 of SURVEY.md §8a.
//
// Mapping to the hardware
//   * one workgroup (4 wave64 = 256 threads, one wave per SIMD) per MPC problem instance; the whole
//     accelerated-proximal-gradient loop runs inside the kernel (no host round trip per iteration).
//   * a wave integrates 32 SDE particles at a time in the v_mfma_f32_32x32x2_f32 accumulator layout:
//     lane l <-> particle column j = l&31, lane half h = l>>5; accumulator register r holds hidden
//     unit rowmap(r,h) = (r&3) + 8*(r>>2) + 4*h of that particle.
//   * MLP layers are chained MFMAs: the layer-1 accumulator registers ARE the B operands of layer 2
//     (k-step r <-> register r), so activations never leave registers. Layer-1 A operands stay in VGPRs,
//     the W2 / W2^T A operands are read from LDS in lane order right before their MFMA chain.
//     f32 MFMA is bit-exactly a k-ordered fmaf chain (tools/mfma_probe.hip), which is what lets the CPU
//     oracle reproduce these kernels bit for bit. On gfx950 f32 MFMA and f32 VALU share the vector
//     datapath (tools/mfma_valu_overlap.hip): the cost model is VALU cycles + 64 cycles per MFMA.
//   * activations / rigid-body physics / cost run on the VALU with the explicit operation order of
//     SPEC.md; tanh uses the batched-reciprocal form (4 values share one Newton reciprocal).
//   * HBM streams, all in 128-byte rows: noise [instance][group][t][6][32]; particle x horizon tensor
//     [instance][group][t][13][32] written by the forward sweep of a gradient evaluation and read back by
//     the adjoint sweep; activation checkpoint [instance][group][t][1280] (layer-2 activations + 5 step
//     scalars) so that the adjoint recomputes only layer 1. The adjoint's loads are issued one step ahead.
//   * per-step control-dependent terms (W1u u_t + b1, rotor thrust/torques) are computed once per
//     rollout per instance into LDS and enter the MFMA as its C operand.
//   * reductions over particles: v_permlane16/32_swap + DPP adds inside the 32-lane groups (bitwise equal
//     to the xor butterflies of SPEC.md §6), fixed slot order across waves; no atomics anywhere, results are
//     run-to-run deterministic and independent of the batch slot.
//   * template <bool F16>: optional fp16-operand contractions on v_mfma_f32_32x32x16_f16 (SPEC.md §9).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "sdempc_kernels.h"

// This file is compiled twice (Makefile): SDEMPC_FAST=0 -> namespace sdempc::exact, the bit-reproducible arithmetic of SPEC.md
// §3 (the default and the path every parity claim is about); SDEMPC_FAST=1 -> namespace sdempc::fastm, the same kernels with the
// hardware transcendentals v_exp_f32 / v_rcp_f32 / v_rsq_f32 in tanh, sigmoid and the quaternion normalisation (SPEC.md §10,
// `math_mode: fast`): not reproducible on a CPU, checked against the oracle within tolerances.
#ifndef SDEMPC_FAST
#define SDEMPC_FAST 0
#endif

namespace sdempc {
#if SDEMPC_FAST
namespace fastm {
#else
namespace exact {
#endif
constexpr bool FAST = SDEMPC_FAST != 0;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
#define FMA(a, b, c) __builtin_fmaf((a), (b), (c))
#define DI __device__ __forceinline__
// phase fences for the instruction scheduler (SDEMPC_SB=0 lets hipcc interleave freely)
#ifndef SDEMPC_SB
#define SDEMPC_SB 1
#endif
#if SDEMPC_SB
#define SCHED_PHASE() __builtin_amdgcn_sched_barrier(0)
#else
#define SCHED_PHASE() ((void)0)
#endif

constexpr int NX = 13, NN = 6, HID = 32;

// A "team" is the set of waves that owns one MPC instance.
//   TeamBlock: the whole workgroup (4 waves, up to 4 particle groups in flight)           -- P > 32
//   TeamWave : one wave per instance, 4 instances per workgroup sharing the LDS weights  -- P <= 32
// Reduction semantics (SPEC.md §6) are identical: dot256 walks its 256 virtual lanes in 256/NT passes.
struct TeamBlock {
    static constexpr int NT = 256, NWAVES = 4, IPB = 1, BNT = 256;   // BNT: threads per workgroup
    DI static int tid() { return threadIdx.x; }
    DI static int team() { return 0; }
    DI static void sync() { __syncthreads(); }
};
// Eight waves (two per SIMD) on one instance: single-instance latency when it has more than four particle groups
// (any wave may take any group: the per-group rows of SPEC.md §6.1 live in global memory, see group_ordered_sum)
struct TeamBlock8 {
    static constexpr int NT = 512, NWAVES = 8, IPB = 1, BNT = 512;
    DI static int tid() { return threadIdx.x; }
    DI static int team() { return 0; }
    DI static void sync() { __syncthreads(); }
};
struct TeamWave {
    static constexpr int NT = 64, NWAVES = 1, IPB = 4, BNT = 256;
    DI static int tid() { return threadIdx.x & 63; }
    DI static int team() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }   // wave-uniform -> SGPR
    // LDS operations of one wave execute in order; the fence keeps the compiler from moving them and drains lgkmcnt
    DI static void sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier(); }
};

// ------------------------------------------------------------------------------------------------
// SPEC.md §3: elementary functions (bit-reproducible: only fma / mul / add / integer ops)
// ------------------------------------------------------------------------------------------------
DI float rcp_spec(float d) {
    float y = __uint_as_float(0x7EF311C7u - __float_as_uint(d));
#pragma unroll
    for (int i = 0; i < 3; ++i) { float e = FMA(-d, y, 1.0f); y = FMA(y, e, y); }
    return y;
}
DI float rsqrt_spec(float a) {
    float y = __uint_as_float(0x5F3759DFu - (__float_as_uint(a) >> 1));
    float h = 0.5f * a;
#pragma unroll
    for (int i = 0; i < 3; ++i) { float t = y * y; t = FMA(-h, t, 1.5f); y = y * t; }
    return y;
}
DI float exp2_spec(float x, float c) {
    float t2 = FMA(x, c, 12582912.0f);
    float n = t2 - 12582912.0f;
    float f = FMA(x, c, -n);
    float p = 0.001327647129073739f;
    p = FMA(p, f, 0.009675540961325169f);
    p = FMA(p, f, 0.05550713092088699f);
    p = FMA(p, f, 0.24022120237350464f);
    p = FMA(p, f, 0.6931469440460205f);
    p = FMA(p, f, 1.0000001192092896f);
    return __uint_as_float(__float_as_uint(p) + (__float_as_uint(t2) << 23));
}
DI float clampf(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
// tanh of 4 values with one shared reciprocal (SPEC.md §3.4)
DI void tanh4(float& a0, float& a1, float& a2, float& a3) {
    float d0 = 1.0f + exp2_spec(clampf(a0, -9.0f, 9.0f), 2.885390043258667f);
    float d1 = 1.0f + exp2_spec(clampf(a1, -9.0f, 9.0f), 2.885390043258667f);
    float d2 = 1.0f + exp2_spec(clampf(a2, -9.0f, 9.0f), 2.885390043258667f);
    float d3 = 1.0f + exp2_spec(clampf(a3, -9.0f, 9.0f), 2.885390043258667f);
    float p2 = d0 * d1, p3 = p2 * d2, p4 = p3 * d3;
    float r = rcp_spec(p4);
    float r3 = r * p3; r = r * d3;
    float r2 = r * p2; r = r * d2;
    float r1 = r * d0;
    float r0 = r * d1;
    a0 = FMA(-2.0f, r0, 1.0f); a1 = FMA(-2.0f, r1, 1.0f); a2 = FMA(-2.0f, r2, 1.0f); a3 = FMA(-2.0f, r3, 1.0f);
}
DI void tanh16(f32x16& v) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float a = v[4 * q], b = v[4 * q + 1], c = v[4 * q + 2], d = v[4 * q + 3];
        tanh4(a, b, c, d);
        v[4 * q] = a; v[4 * q + 1] = b; v[4 * q + 2] = c; v[4 * q + 3] = d;
    }
}
// Packed form of the same arithmetic (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two values per instruction, identical IEEE
// operations per value, so results are bit-identical to tanh4). A v_pk instruction costs two issue slots of the vector
// datapath, so it gains nothing once a SIMD is shared by 2+ waves (tools/tanh_probe.hip: 772 vs 710 cycles per tile at two
// waves per SIMD) but a lone wave per SIMD is issue-bound and gets 1.5x (816 vs 1237 cycles): used by the small-batch
// (latency) instantiation of the solve kernel only.
typedef float f32x2 __attribute__((ext_vector_type(2)));
DI f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
DI f32x2 splat2(float x) { return f32x2{x, x}; }
DI f32x2 exp2d_pk(f32x2 x) {   // 1 + 2^(x*c) of two clamped values
    const f32x2 c = splat2(2.885390043258667f), mg = splat2(12582912.0f);
    f32x2 t2 = pk_fma(x, c, mg);
    f32x2 n = t2 - mg;
    f32x2 f = pk_fma(x, c, -n);
    f32x2 p = splat2(0.001327647129073739f);
    p = pk_fma(p, f, splat2(0.009675540961325169f));
    p = pk_fma(p, f, splat2(0.05550713092088699f));
    p = pk_fma(p, f, splat2(0.24022120237350464f));
    p = pk_fma(p, f, splat2(0.6931469440460205f));
    p = pk_fma(p, f, splat2(1.0000001192092896f));
    f32x2 e;
    e[0] = __uint_as_float(__float_as_uint(p[0]) + (__float_as_uint(t2[0]) << 23));
    e[1] = __uint_as_float(__float_as_uint(p[1]) + (__float_as_uint(t2[1]) << 23));
    return e + splat2(1.0f);
}
// two tanh4 groups at once: group a in element 0 of every pair, group b in element 1 (the two reciprocals share the Newton steps)
DI void tanh8_pk(float* a, float* b) {
    f32x2 a01 = f32x2{clampf(a[0], -9.0f, 9.0f), clampf(a[1], -9.0f, 9.0f)}, a23 = f32x2{clampf(a[2], -9.0f, 9.0f), clampf(a[3], -9.0f, 9.0f)};
    f32x2 b01 = f32x2{clampf(b[0], -9.0f, 9.0f), clampf(b[1], -9.0f, 9.0f)}, b23 = f32x2{clampf(b[2], -9.0f, 9.0f), clampf(b[3], -9.0f, 9.0f)};
    f32x2 da01 = exp2d_pk(a01), da23 = exp2d_pk(a23), db01 = exp2d_pk(b01), db23 = exp2d_pk(b23);
    f32x2 d0 = f32x2{da01[0], db01[0]}, d1 = f32x2{da01[1], db01[1]}, d2 = f32x2{da23[0], db23[0]}, d3 = f32x2{da23[1], db23[1]};
    f32x2 p2 = d0 * d1, p3 = p2 * d2, p4 = p3 * d3;
    f32x2 y;
    y[0] = __uint_as_float(0x7EF311C7u - __float_as_uint(p4[0]));
    y[1] = __uint_as_float(0x7EF311C7u - __float_as_uint(p4[1]));
#pragma unroll
    for (int i = 0; i < 3; ++i) { f32x2 e = pk_fma(-p4, y, splat2(1.0f)); y = pk_fma(y, e, y); }
    f32x2 r = y;
    f32x2 r3 = r * p3; r = r * d3;
    f32x2 r2 = r * p2; r = r * d2;
    f32x2 r1 = r * d0;
    f32x2 r0 = r * d1;
    const f32x2 m2 = splat2(-2.0f), one = splat2(1.0f);
    f32x2 t0 = pk_fma(m2, r0, one), t1 = pk_fma(m2, r1, one), t2 = pk_fma(m2, r2, one), t3 = pk_fma(m2, r3, one);
    a[0] = t0[0]; a[1] = t1[0]; a[2] = t2[0]; a[3] = t3[0];
    b[0] = t0[1]; b[1] = t1[1]; b[2] = t2[1]; b[3] = t3[1];
}
DI void tanh16_pk(f32x16& v) {
#pragma unroll
    for (int q = 0; q < 4; q += 2) {
        float a[4] = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
        float b[4] = {v[4 * q + 4], v[4 * q + 5], v[4 * q + 6], v[4 * q + 7]};
        tanh8_pk(a, b);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[4 * q + i] = a[i]; v[4 * q + 4 + i] = b[i]; }
    }
}
// math_mode fast (SPEC.md §10): 1 - 2 / (1 + 2^(x * 2 log2 e)) on the transcendental unit; saturates through inf / 0 without a clamp
DI void tanh16_hw(f32x16& v) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(v[r] * 2.885390043258667f);
        v[r] = FMA(-2.0f, __builtin_amdgcn_rcpf(1.0f + e), 1.0f);
    }
}
template <bool PK>
DI void tanh_tile(f32x16& v) {
    if constexpr (FAST) tanh16_hw(v);
    else if constexpr (PK) tanh16_pk(v);
    else tanh16(v);
}
DI float sigmoid_spec(float x) {
    if constexpr (FAST) return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950216293335f));
    float E = exp2_spec(clampf(x, -30.0f, 30.0f), -1.4426950216293335f);
    return rcp_spec(1.0f + E);
}

// ------------------------------------------------------------------------------------------------
// shared-memory carve (floats). One instance per workgroup.
// ------------------------------------------------------------------------------------------------
struct Smem {
    float *W3, *w3n, *b1n, *b2, *b1d, *W1zT, *W1uT;  // weights, row-major in hidden-unit index
    float *A2, *A2T;                                   // MFMA A operands of W2 / W2^T: [q][lane][4]
    float *A2h;                                        // f16 mode: A operands of W2, [half][lane][8 x fp16]
    float *ust;                                        // [H][36]: c[32], Tz, tau[3]
    float *xref;                                       // [H+1][13]
    float *dt, *sdt, *disc;                            // [H], [H][6], [H+1]
    float *red;                                        // [16] block-reduction scratch
    float *tot;                                        // cooperative path only: [H*12] particle sums of the adjoint outputs
    float *v[6];                                       // N-vectors: 0 xk, 1 yk, 2 xn, 3 g, 4 d1, 5 ucur
};
constexpr int UST = 36;

DI Smem carve(float* base, int H, int m, int team, bool coop = false) {
    Smem s;
    float* p = base;
    // ---- shared by every team of the workgroup ----
    s.W3 = p; p += 6 * HID;
    s.w3n = p; p += HID;
    s.b1n = p; p += HID;
    s.b2 = p; p += HID;
    s.b1d = p; p += HID;
    s.W1zT = p; p += NN * 2 * HID;
    s.W1uT = p; p += 8 * HID;
    s.A2 = p; p += HID * HID;
    s.A2T = p; p += HID * HID;
    s.A2h = p; p += 512;
    s.dt = p; p += (H + 3) & ~3;
    s.sdt = p; p += (H * NN + 3) & ~3;
    s.disc = p; p += (H + 1 + 3) & ~3;
    // ---- per team ----
    const int nv = (H * m + 3) & ~3;
    const int per_team = H * UST + (((H + 1) * NX + 3) & ~3) + 16 + 6 * nv;
    p += team * per_team;
    s.ust = p; p += H * UST;
    s.xref = p; p += ((H + 1) * NX + 3) & ~3;
    s.red = p; p += 16;
    for (int i = 0; i < 6; ++i) { s.v[i] = p; p += nv; }
    s.tot = p;                       // only the cooperative kernel (one team per workgroup) reserves it: see smem_bytes
    (void)coop;
    return s;
}
size_t smem_bytes(int H, int m, int ipb, bool coop = false) {
    size_t shared = 6 * HID + 4 * HID + NN * 2 * HID + 8 * HID + 2 * HID * HID + 512 + ((H + 3) & ~3) + ((H * NN + 3) & ~3) + ((H + 1 + 3) & ~3);
    size_t per_team = (size_t)H * UST + (((H + 1) * NX + 3) & ~3) + 16 + 6 * (size_t)((H * m + 3) & ~3);
    return (shared + ipb * per_team + (coop ? (size_t)((H * 12 + 3) & ~3) : 0)) * sizeof(float);
}

// blob float payload offsets (SPEC.md §2)
constexpr int OFF_W1Z = 56, OFF_B1 = OFF_W1Z + 384, OFF_W1U = OFF_B1 + 64, OFF_W2 = OFF_W1U + 256, OFF_B2 = OFF_W2 + 1024,
              OFF_W3 = OFF_B2 + 32, OFF_B3 = OFF_W3 + 256, OFF_W3N = OFF_B3 + 8, OFF_B3N = OFF_W3N + 32;

// MFMA A operands kept in registers for the whole kernel
struct WaveW {
    float w1d[3], w1n[3];   // f32 mode: layer-1 A operands (3 k-steps per tile)
    half8 h1d, h1n;         // f16 mode: layer-1 A operands (k slots 0..5 of lanes 0..31, rest zero)
};

DI int opaque_s(int v) { asm volatile("" : "+s"(v)); return v; }
DI int opaque_v(int v) { asm volatile("" : "+v"(v)); return v; }
DI int rowmap(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// cooperative (all BNT threads of the workgroup); caller issues __syncthreads() afterwards
DI void load_weights(const KArgs& a, const Smem& sm, WaveW& ww, int tid, int BNT) {
    const float* w = a.wts;
    const int lane = tid & 63, j = lane & 31, h = lane >> 5;
    for (int i = tid; i < 6 * HID; i += BNT) sm.W3[i] = w[OFF_W3 + i];
    for (int i = tid; i < HID; i += BNT) {
        sm.w3n[i] = w[OFF_W3N + i];
        sm.b1d[i] = w[OFF_B1 + i];
        sm.b1n[i] = w[OFF_B1 + HID + i];
        sm.b2[i] = w[OFF_B2 + i];
    }
    for (int i = tid; i < NN * 2 * HID; i += BNT) { int k = i / (2 * HID), r = i % (2 * HID); sm.W1zT[i] = w[OFF_W1Z + r * NN + k]; }
    for (int i = tid; i < 8 * HID; i += BNT) { int j = i / HID, r = i % HID; sm.W1uT[i] = w[OFF_W1U + r * 8 + j]; }
    for (int i = tid; i < a.H; i += BNT) sm.dt[i] = a.dt[i];
    for (int i = tid; i < a.H * NN; i += BNT) sm.sdt[i] = a.sdt[i];
    for (int i = tid; i <= a.H; i += BNT) sm.disc[i] = a.disc[i];
#pragma unroll
    for (int s = 0; s < 3; ++s) { ww.w1d[s] = w[OFF_W1Z + j * NN + 2 * s + h]; ww.w1n[s] = w[OFF_W1Z + (HID + j) * NN + 2 * s + h]; }
    // A operand of k-step r for lane l: W2[j][rowmap(r,h)] (forward) / W2[rowmap(r,h)][j] (transpose)
    for (int i = tid; i < HID * HID; i += BNT) {
        int c = i & 3, l = (i >> 2) & 63, q = i >> 8, jj = l & 31, hh = l >> 5, r = 4 * q + c;
        sm.A2[i] = w[OFF_W2 + jj * HID + rowmap(r, hh)];
        sm.A2T[i] = w[OFF_W2 + rowmap(r, hh) * HID + jj];
    }
    if (a.f16) {   // weights are already fp16-representable (quantised on the host): the casts are exact
        _Float16* ah = reinterpret_cast<_Float16*>(sm.A2h);
        for (int i = tid; i < 2 * 64 * 8; i += BNT) {
            int e = i & 7, l = (i >> 3) & 63, hf = i >> 9, jj = l & 31, hh = l >> 5;
            ah[i] = (_Float16)w[OFF_W2 + jj * HID + rowmap(8 * hf + e, hh)];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            ww.h1d[e] = (e < NN && h == 0) ? (_Float16)w[OFF_W1Z + j * NN + e] : (_Float16)0.0f;
            ww.h1n[e] = (e < NN && h == 0) ? (_Float16)w[OFF_W1Z + (HID + j) * NN + e] : (_Float16)0.0f;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// block-wide helpers (all 256 threads call; the result is identical in every thread)
// ------------------------------------------------------------------------------------------------
// Cross-lane sums without LDS traffic. Each stage adds the value of the xor-partner lane; because
// float addition is commutative and every stage leaves partner lanes bitwise equal, rotations inside
// already-periodic rows reproduce the xor butterfly of SPEC.md §6 exactly:
//   xor 32: v_permlane32_swap(v,v) -> {lo,lo},{hi,hi};  xor 16: v_permlane16_swap(v,v);
//   xor 8 / 4: DPP row_ror:8 / row_ror:4 (rows are 8-periodic after the xor-8 stage);
//   xor 2 / 1: DPP quad_perm [2,3,0,1] / [1,0,3,2].
template <int CTRL>
DI float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
// NB: the swap instructions exchange halves/rows BETWEEN two registers; given the same register twice
// they alias (tools/lane_probe.hip), so the second operand is forced into its own VGPR.
DI float xor32_sum(float v) {
    unsigned a = __builtin_bit_cast(unsigned, v), b;
    asm("v_mov_b32 %0, %1" : "=v"(b) : "v"(a));   // opaque copy (a tied "+v" copy gets folded away by LLVM); not volatile: may be DCEd
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);   // r[0] = {lo,lo}, r[1] = {hi,hi}
    const unsigned r0 = r[0], r1 = r[1];   // (bit_cast straight from r[1] reads element 0: keep the temporaries)
    return __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
}
DI float xor16_sum(float v) {
    unsigned a = __builtin_bit_cast(unsigned, v), b;
    asm("v_mov_b32 %0, %1" : "=v"(b) : "v"(a));
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);   // r[0] = {r0,r0,r2,r2}, r[1] = {r1,r1,r3,r3}
    const unsigned r0 = r[0], r1 = r[1];
    return __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
}
DI float group_bfly32(float v) {
    v = xor16_sum(v);
    v = v + dpp_f<0x128>(v);  // row_ror:8
    v = v + dpp_f<0x124>(v);  // row_ror:4
    v = v + dpp_f<0x4E>(v);   // quad_perm [2,3,0,1]
    v = v + dpp_f<0xB1>(v);   // quad_perm [1,0,3,2]
    return v;
}
DI float wave_bfly64(float v) { return group_bfly32(xor32_sum(v)); }
// SPEC.md §6.2 dot256: virtual lane i < 256 chains e = i, i+256, ...; butterflies inside each 64-lane virtual wave;
// ((w0+w1)+w2)+w3. A team of NT threads walks the 256 virtual lanes in 256/NT passes.
template <class Team, class F>
DI float team_reduce256(const Smem& sm, int N, int tid, F&& elem) {
    if constexpr (Team::NT >= 256) {      // the first four waves are the 256 virtual lanes; further waves only take part in the barriers
        float acc = 0.0f;
        if (tid < 256)
            for (int e = tid; e < N; e += 256) acc = elem(e, acc);
        acc = wave_bfly64(acc);
        Team::sync();
        if ((tid & 63) == 0 && tid < 256) sm.red[tid >> 6] = acc;
        Team::sync();
        return ((sm.red[0] + sm.red[1]) + sm.red[2]) + sm.red[3];
    } else {
        float w[4];
#pragma unroll
        for (int vw = 0; vw < 4; ++vw) {
            float acc = 0.0f;
            for (int e = vw * 64 + tid; e < N; e += 256) acc = elem(e, acc);
            w[vw] = wave_bfly64(acc);
        }
        return ((w[0] + w[1]) + w[2]) + w[3];
    }
}
template <class Team>
DI float block_dot(const Smem& sm, const float* x, const float* y, int N, int tid) {
    return team_reduce256<Team>(sm, N, tid, [&](int e, float acc) { return FMA(x[e], y ? y[e] : 1.0f, acc); });
}

// SPEC.md §5.5: control cost element and d/du pieces
DI float slew_dw(const KArgs& a, const float* u, int t, int j, int m, float& cterm) {
    // returns dw(t,j) and the (undiscounted) slew cost term for t >= 1
    float ds = u[t * m + j] - u[(t - 1) * m + j];
    float c = (a.C.slew * ds) * ds;
    float d = (2.0f * a.C.slew) * ds;
    if (a.C.has_sc) {
        float hi = ds - a.C.slew_hi[j]; hi = hi < 0.0f ? 0.0f : hi;
        float lo = a.C.slew_lo[j] - ds; lo = lo < 0.0f ? 0.0f : lo;
        c = FMA(a.C.slew_cc * hi, hi, c);
        c = FMA(a.C.slew_cc * lo, lo, c);
        d = FMA(2.0f * a.C.slew_cc, hi - lo, d);
    }
    cterm = c;
    return d;
}
DI float ucost_elem(const KArgs& a, const Smem& sm, const float* u, int e, int m) {
    int t = e / m, j = e - t * m;
    float du = u[e] - a.C.uref[j];
    float c = (a.C.uerr * du) * du;
    if (t >= 1) {
        float ds = u[e] - u[e - m];
        c = FMA(a.C.slew * ds, ds, c);
        if (a.C.has_sc) {
            float hi = ds - a.C.slew_hi[j]; hi = hi < 0.0f ? 0.0f : hi;
            float lo = a.C.slew_lo[j] - ds; lo = lo < 0.0f ? 0.0f : lo;
            c = FMA(a.C.slew_cc * hi, hi, c);
            c = FMA(a.C.slew_cc * lo, lo, c);
        }
    }
    return sm.disc[t] * c;
}
template <class Team>
DI float block_ucost(const KArgs& a, const Smem& sm, const float* u, int tid) {
    return team_reduce256<Team>(sm, a.H * a.m, tid, [&](int e, float acc) { return FMA(ucost_elem(a, sm, u, e, a.m), 1.0f, acc); });
}

// SPEC.md §5.1: per-step control-dependent constants into sm.ust
template <class Team>
DI void block_prepass(const KArgs& a, const Smem& sm, const float* u, int tid) {
    const int H = a.H, m = a.m;
    for (int e = tid; e < H * HID; e += Team::NT) {
        int t = e >> 5, r = e & 31;
        float c = sm.b1d[r];
        for (int j = 0; j < m; ++j) c = FMA(sm.W1uT[j * HID + r], u[t * m + j], c);
        sm.ust[t * UST + r] = c;
    }
    for (int t = tid; t < H; t += Team::NT) {
        float Tz = 0.0f, t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;
        for (int j = 0; j < m; ++j) {
            float uj = u[t * m + j];
            float T = FMA(FMA(a.M.ct2, uj, a.M.ct1), uj, a.M.ct0);
            float Mq = a.M.dir[j] * (FMA(a.M.cm2, uj, a.M.cm1) * uj);
            Tz = Tz + T;
            t0 = FMA(a.M.ry[j], T, t0);
            t1 = FMA(-a.M.rx[j], T, t1);
            t2 = t2 + Mq;
        }
        sm.ust[t * UST + 32] = Tz; sm.ust[t * UST + 33] = t0; sm.ust[t * UST + 34] = t1; sm.ust[t * UST + 35] = t2;
    }
}

// ------------------------------------------------------------------------------------------------
// one Euler–Maruyama step for the wave's 32 particles (SPEC.md §5.2)
// ------------------------------------------------------------------------------------------------
struct StepAux {
    float Rm[9];
    f32x16 h1d, h1n, h2;
    float eta, Fb[3], Jom[3], rn, qn[4];
};


// ---- uniform head of a step: rotation matrix and the MLP inputs z = (R^T v, omega) ----
DI void fwd_head(const float* x, float* Rm, float* z) {
    const float qw = x[6], qx = x[7], qy = x[8], qz = x[9];
    const float xx = qx * qx, yy = qy * qy, zz = qz * qz;
    const float xy = qx * qy, xz = qx * qz, yz = qy * qz, wx = qw * qx, wy = qw * qy, wz = qw * qz;
    Rm[0] = FMA(-2.0f, yy + zz, 1.0f); Rm[1] = 2.0f * (xy - wz);          Rm[2] = 2.0f * (xz + wy);
    Rm[3] = 2.0f * (xy + wz);          Rm[4] = FMA(-2.0f, xx + zz, 1.0f); Rm[5] = 2.0f * (yz - wx);
    Rm[6] = 2.0f * (xz - wy);          Rm[7] = 2.0f * (yz + wx);          Rm[8] = FMA(-2.0f, xx + yy, 1.0f);
#pragma unroll
    for (int j = 0; j < 3; ++j) z[j] = FMA(Rm[6 + j], x[5], FMA(Rm[3 + j], x[4], Rm[j] * x[3]));
    z[3] = x[10]; z[4] = x[11]; z[5] = x[12];
}

// ---- MLPs of a step in the MFMA tile layout (32 particles per wave): outputs o[6] and eta per particle ----
template <bool F16, bool PK>
DI void fwd_mlp_tiles(const KArgs& a, const Smem& sm, const WaveW& ww, const float* ust, int h, int lane, const float* z, StepAux& A, float* o, float& eta_out) {
    // layer 1: C operand = per-step offsets (drift) / bias (density); K = 6 -> 3 MFMAs per tile
    f32x16 accD, accN;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float4 c4 = *reinterpret_cast<const float4*>(ust + 8 * q + 4 * h);
        float4 n4 = *reinterpret_cast<const float4*>(sm.b1n + 8 * q + 4 * h);
        accD[4 * q] = c4.x; accD[4 * q + 1] = c4.y; accD[4 * q + 2] = c4.z; accD[4 * q + 3] = c4.w;
        accN[4 * q] = n4.x; accN[4 * q + 1] = n4.y; accN[4 * q + 2] = n4.z; accN[4 * q + 3] = n4.w;
    }
    if constexpr (F16) {
        // fp16 operands (round toward zero), f32 accumulate: one v_mfma_f32_32x32x16_f16 per tile, k slots 0..5 live in lanes 0..31
        half8 bv;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            auto pk = __builtin_amdgcn_cvt_pkrtz(h ? 0.0f : z[2 * e], h ? 0.0f : z[2 * e + 1]);
            bv[2 * e] = (_Float16)pk[0]; bv[2 * e + 1] = (_Float16)pk[1];
        }
        bv[6] = (_Float16)0.0f; bv[7] = (_Float16)0.0f;
        accD = __builtin_amdgcn_mfma_f32_32x32x16_f16(ww.h1d, bv, accD, 0, 0, 0);
        accN = __builtin_amdgcn_mfma_f32_32x32x16_f16(ww.h1n, bv, accN, 0, 0, 0);
    } else {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            float b = h ? z[2 * s + 1] : z[2 * s];
            accD = __builtin_amdgcn_mfma_f32_32x32x2f32(ww.w1d[s], b, accD, 0, 0, 0);
            accN = __builtin_amdgcn_mfma_f32_32x32x2f32(ww.w1n[s], b, accN, 0, 0, 0);
        }
    }
    SCHED_PHASE();

    tanh_tile<PK>(accD);
    SCHED_PHASE();

    tanh_tile<PK>(accN);
    A.h1d = accD; A.h1n = accN;
    SCHED_PHASE();

    // layer 2 (drift): B operand of k-step r is accumulator register r of layer 1
    f32x16 acc2;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float4 b4 = *reinterpret_cast<const float4*>(sm.b2 + 8 * q + 4 * h);
        acc2[4 * q] = b4.x; acc2[4 * q + 1] = b4.y; acc2[4 * q + 2] = b4.z; acc2[4 * q + 3] = b4.w;
    }
    if constexpr (F16) {
        // two K=16 MFMAs: k slot e of lane half h <-> accumulator register 8*hf + e, i.e. hidden unit rowmap(8*hf + e, h)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            half8 av = *reinterpret_cast<const half8*>(sm.A2h + (hf * 64 + lane) * 4);
            half8 bv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                auto pk = __builtin_amdgcn_cvt_pkrtz(accD[8 * hf + 2 * e], accD[8 * hf + 2 * e + 1]);
                bv[2 * e] = (_Float16)pk[0]; bv[2 * e + 1] = (_Float16)pk[1];
            }
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc2, 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 w4 = *reinterpret_cast<const float4*>(sm.A2 + (q * 64 + lane) * 4);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.x, accD[4 * q], acc2, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.y, accD[4 * q + 1], acc2, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.z, accD[4 * q + 2], acc2, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.w, accD[4 * q + 3], acc2, 0, 0, 0);
        }
    }
    SCHED_PHASE();

    tanh_tile<PK>(acc2);
    A.h2 = acc2;
    SCHED_PHASE();

    // output layers on the VALU: per-half partial chains, then (P0 + P1) + bias
    {
        float Po[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) Po[i] = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                float4 w4 = *reinterpret_cast<const float4*>(sm.W3 + i * HID + 8 * q + 4 * h);
                Po[i] = FMA(w4.x, acc2[4 * q], Po[i]); Po[i] = FMA(w4.y, acc2[4 * q + 1], Po[i]); Po[i] = FMA(w4.z, acc2[4 * q + 2], Po[i]); Po[i] = FMA(w4.w, acc2[4 * q + 3], Po[i]);
            }
            SCHED_PHASE();
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) o[i] = xor32_sum(Po[i]) + a.M.b3[i];
    }
    float eta;
    {
        float P = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 w4 = *reinterpret_cast<const float4*>(sm.w3n + 8 * q + 4 * h);
            P = FMA(w4.x, accN[4 * q], P); P = FMA(w4.y, accN[4 * q + 1], P); P = FMA(w4.z, accN[4 * q + 2], P); P = FMA(w4.w, accN[4 * q + 3], P);
        }
        eta = sigmoid_spec(xor32_sum(P) + a.M.b3n);
    }
    eta_out = eta;
    SCHED_PHASE();
}

// ---- uniform tail of a step: rigid body, Euler-Maruyama update, quaternion renormalisation ----
DI void fwd_tail(const KArgs& a, const Smem& sm, const float* ust, int t, const float* x, const float* xi, const float* Rm, const float* o, float eta, float* xn, StepAux& A) {
    const float dt = sm.dt[t];
    const float qw = x[6], qx = x[7], qy = x[8], qz = x[9];
    A.eta = eta;
    // rigid body
    A.Fb[0] = a.M.sF[0] * o[0]; A.Fb[1] = a.M.sF[1] * o[1]; A.Fb[2] = FMA(a.M.sF[2], o[2], ust[32]);
    float acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float Fw = FMA(Rm[3 * i + 2], A.Fb[2], FMA(Rm[3 * i + 1], A.Fb[1], Rm[3 * i] * A.Fb[0]));
        acc[i] = Fw * a.M.inv_mass;
    }
    acc[2] = acc[2] - a.M.grav;
    float taub[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { taub[i] = FMA(a.M.sT[i], o[3 + i], ust[33 + i]); A.Jom[i] = a.M.J[i] * x[10 + i]; }
    float cr[3];
    cr[0] = FMA(x[11], A.Jom[2], -(x[12] * A.Jom[1]));
    cr[1] = FMA(x[12], A.Jom[0], -(x[10] * A.Jom[2]));
    cr[2] = FMA(x[10], A.Jom[1], -(x[11] * A.Jom[0]));
    float dom[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) dom[i] = (taub[i] - cr[i]) * a.M.iJ[i];
    float dq[4];
    dq[0] = -0.5f * FMA(qz, x[12], FMA(qy, x[11], qx * x[10]));
    dq[1] = 0.5f * FMA(-qz, x[11], FMA(qy, x[12], qw * x[10]));
    dq[2] = 0.5f * FMA(-qx, x[12], FMA(qz, x[10], qw * x[11]));
    dq[3] = 0.5f * FMA(-qy, x[10], FMA(qx, x[11], qw * x[12]));
    const float* sdt = sm.sdt + t * NN;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        xn[i] = FMA(x[3 + i], dt, x[i]);
        xn[3 + i] = FMA(sdt[i] * eta, xi[i], FMA(acc[i], dt, x[3 + i]));
        xn[10 + i] = FMA(sdt[3 + i] * eta, xi[3 + i], FMA(dom[i], dt, x[10 + i]));
    }
    float qt[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) qt[i] = FMA(dq[i], dt, x[6 + i]);
    float n2 = FMA(qt[3], qt[3], FMA(qt[2], qt[2], FMA(qt[1], qt[1], qt[0] * qt[0])));
    if constexpr (FAST) A.rn = __builtin_amdgcn_rsqf(n2); else A.rn = rsqrt_spec(n2);
#pragma unroll
    for (int i = 0; i < 4; ++i) { A.qn[i] = qt[i] * A.rn; xn[6 + i] = A.qn[i]; }
}

template <bool F16, bool PK = false>
DI void step_fwd(const KArgs& a, const Smem& sm, const WaveW& ww, int t, int h, int lane, const float* x, const float* xi, float* xn, StepAux& A) {
    const float* ust = sm.ust + t * UST;
    float z[NN];
    fwd_head(x, A.Rm, z);
    SCHED_PHASE();
    float o[6], eta;
    fwd_mlp_tiles<F16, PK>(a, sm, ww, ust, h, lane, z, A, o, eta);
    fwd_tail(a, sm, ust, t, x, xi, A.Rm, o, eta, xn, A);
}

// rotation matrix of q (same expressions as in step_fwd)
DI void rot_from_q(const float* x, float* Rm) {
    const float qw = x[6], qx = x[7], qy = x[8], qz = x[9];
    const float xx = qx * qx, yy = qy * qy, zz = qz * qz;
    const float xy = qx * qy, xz = qx * qz, yz = qy * qz, wx = qw * qx, wy = qw * qy, wz = qw * qz;
    Rm[0] = FMA(-2.0f, yy + zz, 1.0f); Rm[1] = 2.0f * (xy - wz);          Rm[2] = 2.0f * (xz + wy);
    Rm[3] = 2.0f * (xy + wz);          Rm[4] = FMA(-2.0f, xx + zz, 1.0f); Rm[5] = 2.0f * (yz - wx);
    Rm[6] = 2.0f * (xz - wy);          Rm[7] = 2.0f * (yz + wx);          Rm[8] = FMA(-2.0f, xx + yy, 1.0f);
}

// SPEC.md §5.3 stage cost at x_{t+1}; GX: also the gradient
template <bool GX>
DI float stage_cost(const KArgs& a, const float* x, const float* xr, float* gx) {
    float l = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) { float e = x[i] - xr[i]; float w = a.C.perr[i] * e; l = FMA(w, e, l); if (GX) gx[i] = 2.0f * w; }
#pragma unroll
    for (int i = 0; i < 3; ++i) { float e = x[3 + i] - xr[3 + i]; float w = a.C.verr[i] * e; l = FMA(w, e, l); if (GX) gx[3 + i] = 2.0f * w; }
#pragma unroll
    for (int i = 0; i < 3; ++i) { float e = x[10 + i] - xr[10 + i]; float w = a.C.werr[i] * e; l = FMA(w, e, l); if (GX) gx[10 + i] = 2.0f * w; }
    float qw = x[6], qx = x[7], qy = x[8], qz = x[9], rw = xr[6], rx = xr[7], ry = xr[8], rz = xr[9];
    float ex = FMA(rz, qy, FMA(-ry, qz, FMA(-rx, qw, rw * qx)));
    float ey = FMA(-rz, qx, FMA(-ry, qw, FMA(rx, qz, rw * qy)));
    float ez = FMA(-rz, qw, FMA(ry, qx, FMA(-rx, qy, rw * qz)));
    float wxe = a.C.qerr[0] * ex, wye = a.C.qerr[1] * ey, wze = a.C.qerr[2] * ez;
    l = FMA(wxe, ex, l); l = FMA(wye, ey, l); l = FMA(wze, ez, l);
    if (GX) {
        float ga = 2.0f * wxe, gb = 2.0f * wye, gc = 2.0f * wze;
        gx[6] = FMA(-rz, gc, FMA(-ry, gb, -rx * ga));
        gx[7] = FMA(ry, gc, FMA(-rz, gb, rw * ga));
        gx[8] = FMA(-rx, gc, FMA(rw, gb, rz * ga));
        gx[9] = FMA(rw, gc, FMA(rx, gb, -ry * ga));
    }
    return l;
}

// ------------------------------------------------------------------------------------------------
// vector-Jacobian product of one step (SPEC.md §5.4). gq[0..m-1] = W1u^T abar1, gq[m] = Tz adjoint,
// gq[m+1..m+3] = rotor-torque adjoint
// ------------------------------------------------------------------------------------------------
// Uniform (per particle) quantities that the three parts of the step's vector-Jacobian product share
struct VjpTmp {
    float ebraw, qtb[4], dqb[4], omb[3], Fwb[3], ob[6];
};

// ---- head: everything upstream of the MLPs (per particle); gq[M..M+3] = thrust / rotor-torque adjoints ----
template <int M>
DI void vjp_head(const KArgs& a, const Smem& sm, int t, const float* x, const float* xi, const StepAux& A, const float* L, float etabar_cost, VjpTmp& T, float* gq) {
    const float dt = sm.dt[t];
    const float* sdt = sm.sdt + t * NN;
    const float* Rm = A.Rm;
    const float* om = x + 10;
    float eb = etabar_cost;
#pragma unroll
    for (int i = 0; i < 3; ++i) eb = FMA(L[3 + i] * sdt[i], xi[i], eb);
#pragma unroll
    for (int i = 0; i < 3; ++i) eb = FMA(L[10 + i] * sdt[3 + i], xi[3 + i], eb);
    T.ebraw = eb * (A.eta * (1.0f - A.eta));
    float dotq = FMA(A.qn[3], L[9], FMA(A.qn[2], L[8], FMA(A.qn[1], L[7], A.qn[0] * L[6])));
    float* qtb = T.qtb; float* dqb = T.dqb;
#pragma unroll
    for (int i = 0; i < 4; ++i) { qtb[i] = A.rn * FMA(-A.qn[i], dotq, L[6 + i]); dqb[i] = qtb[i] * dt; }
    float taub_b[3], crb[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { taub_b[i] = (L[10 + i] * dt) * a.M.iJ[i]; crb[i] = -taub_b[i]; }
    float* omb = T.omb; float Jb[3];
    omb[0] = L[10] + FMA(A.Jom[1], crb[2], -(A.Jom[2] * crb[1]));
    omb[1] = L[11] + FMA(A.Jom[2], crb[0], -(A.Jom[0] * crb[2]));
    omb[2] = L[12] + FMA(A.Jom[0], crb[1], -(A.Jom[1] * crb[0]));
    Jb[0] = FMA(crb[1], om[2], -(crb[2] * om[1]));
    Jb[1] = FMA(crb[2], om[0], -(crb[0] * om[2]));
    Jb[2] = FMA(crb[0], om[1], -(crb[1] * om[0]));
#pragma unroll
    for (int i = 0; i < 3; ++i) omb[i] = FMA(a.M.J[i], Jb[i], omb[i]);
    float* Fwb = T.Fwb; float Fbb[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) Fwb[i] = (L[3 + i] * dt) * a.M.inv_mass;
#pragma unroll
    for (int j = 0; j < 3; ++j) Fbb[j] = FMA(Rm[6 + j], Fwb[2], FMA(Rm[3 + j], Fwb[1], Rm[j] * Fwb[0]));
    float* ob = T.ob;
#pragma unroll
    for (int i = 0; i < 3; ++i) { ob[i] = a.M.sF[i] * Fbb[i]; ob[3 + i] = a.M.sT[i] * taub_b[i]; }
    gq[M] = Fbb[2];
    gq[M + 1] = taub_b[0]; gq[M + 2] = taub_b[1]; gq[M + 3] = taub_b[2];
}

// ---- MLP part in the MFMA tile layout: zb[6] = adjoint of z, gq[0..M-1] = W1u^T abar1 (per particle) ----
template <int M>
DI void vjp_mlp_tiles(const Smem& sm, int h, int lane, const StepAux& A, const VjpTmp& T, float* zb, float* gq) {
    const float ebraw = T.ebraw;
    const float* ob = T.ob;
    // MLP VJP. Order chosen to keep few tiles live: density tile first (frees h1n), then the drift
    // tile: abar2 on the VALU, W2^T abar2 by MFMA in the accumulator layout.
    {
        float Pz[NN], Pu[M];
#pragma unroll
        for (int k = 0; k < NN; ++k) Pz[k] = 0.0f;
#pragma unroll
        for (int jj = 0; jj < M; ++jj) Pu[jj] = 0.0f;
        // density net: abar1n = (w3n * etaraw_bar) * (1 - h1n^2); zbar += W1z[32:64]^T abar1n
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 wn4 = *reinterpret_cast<const float4*>(sm.w3n + 8 * q + 4 * h);
            float an0 = (wn4.x * ebraw) * FMA(-A.h1n[4 * q], A.h1n[4 * q], 1.0f);
            float an1 = (wn4.y * ebraw) * FMA(-A.h1n[4 * q + 1], A.h1n[4 * q + 1], 1.0f);
            float an2 = (wn4.z * ebraw) * FMA(-A.h1n[4 * q + 2], A.h1n[4 * q + 2], 1.0f);
            float an3 = (wn4.w * ebraw) * FMA(-A.h1n[4 * q + 3], A.h1n[4 * q + 3], 1.0f);
#pragma unroll
            for (int k = 0; k < NN; ++k) {
                float4 w4 = *reinterpret_cast<const float4*>(sm.W1zT + k * 2 * HID + HID + 8 * q + 4 * h);
                Pz[k] = FMA(w4.x, an0, Pz[k]); Pz[k] = FMA(w4.y, an1, Pz[k]); Pz[k] = FMA(w4.z, an2, Pz[k]); Pz[k] = FMA(w4.w, an3, Pz[k]);
            }
            SCHED_PHASE();
        }
        // drift net
        f32x16 a2b;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float hb0 = 0.0f, hb1 = 0.0f, hb2 = 0.0f, hb3 = 0.0f;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                float4 w4 = *reinterpret_cast<const float4*>(sm.W3 + i * HID + 8 * q + 4 * h);
                hb0 = FMA(w4.x, ob[i], hb0); hb1 = FMA(w4.y, ob[i], hb1); hb2 = FMA(w4.z, ob[i], hb2); hb3 = FMA(w4.w, ob[i], hb3);
            }
            a2b[4 * q] = hb0 * FMA(-A.h2[4 * q], A.h2[4 * q], 1.0f);
            a2b[4 * q + 1] = hb1 * FMA(-A.h2[4 * q + 1], A.h2[4 * q + 1], 1.0f);
            a2b[4 * q + 2] = hb2 * FMA(-A.h2[4 * q + 2], A.h2[4 * q + 2], 1.0f);
            a2b[4 * q + 3] = hb3 * FMA(-A.h2[4 * q + 3], A.h2[4 * q + 3], 1.0f);
            SCHED_PHASE();
        }
        f32x16 accB;
#pragma unroll
        for (int r = 0; r < 16; ++r) accB[r] = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 w4 = *reinterpret_cast<const float4*>(sm.A2T + (q * 64 + lane) * 4);
            accB = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.x, a2b[4 * q], accB, 0, 0, 0);
            accB = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.y, a2b[4 * q + 1], accB, 0, 0, 0);
            accB = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.z, a2b[4 * q + 2], accB, 0, 0, 0);
            accB = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.w, a2b[4 * q + 3], accB, 0, 0, 0);
        }
        SCHED_PHASE();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float ad0 = accB[4 * q] * FMA(-A.h1d[4 * q], A.h1d[4 * q], 1.0f);
            float ad1 = accB[4 * q + 1] * FMA(-A.h1d[4 * q + 1], A.h1d[4 * q + 1], 1.0f);
            float ad2 = accB[4 * q + 2] * FMA(-A.h1d[4 * q + 2], A.h1d[4 * q + 2], 1.0f);
            float ad3 = accB[4 * q + 3] * FMA(-A.h1d[4 * q + 3], A.h1d[4 * q + 3], 1.0f);
#pragma unroll
            for (int k = 0; k < NN; ++k) {
                float4 w4 = *reinterpret_cast<const float4*>(sm.W1zT + k * 2 * HID + 8 * q + 4 * h);
                Pz[k] = FMA(w4.x, ad0, Pz[k]); Pz[k] = FMA(w4.y, ad1, Pz[k]); Pz[k] = FMA(w4.z, ad2, Pz[k]); Pz[k] = FMA(w4.w, ad3, Pz[k]);
            }
#pragma unroll
            for (int jj = 0; jj < M; ++jj) {
                float4 w4 = *reinterpret_cast<const float4*>(sm.W1uT + jj * HID + 8 * q + 4 * h);
                Pu[jj] = FMA(w4.x, ad0, Pu[jj]); Pu[jj] = FMA(w4.y, ad1, Pu[jj]); Pu[jj] = FMA(w4.z, ad2, Pu[jj]); Pu[jj] = FMA(w4.w, ad3, Pu[jj]);
            }
            SCHED_PHASE();
        }
#pragma unroll
        for (int k = 0; k < NN; ++k) zb[k] = xor32_sum(Pz[k]);
#pragma unroll
        for (int jj = 0; jj < M; ++jj) gq[jj] = xor32_sum(Pu[jj]);
    }
    SCHED_PHASE();
}

// ---- tail: adjoint of the state (per particle) ----
DI void vjp_tail(const Smem& sm, int t, const float* x, const StepAux& A, const float* L, const VjpTmp& T, const float* zb, float* lam) {
    const float dt = sm.dt[t];
    const float* Rm = A.Rm;
    const float qw = x[6], qx = x[7], qy = x[8], qz = x[9];
    const float* v = x + 3;
    const float* om = x + 10;
    const float* qtb = T.qtb; const float* dqb = T.dqb; const float* Fwb = T.Fwb;
    float omb[3] = {T.omb[0], T.omb[1], T.omb[2]};
#pragma unroll
    for (int i = 0; i < 3; ++i) omb[i] = omb[i] + zb[3 + i];
    float vbar[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float Rvb = FMA(Rm[3 * i + 2], zb[2], FMA(Rm[3 * i + 1], zb[1], Rm[3 * i] * zb[0]));
        vbar[i] = FMA(L[i], dt, L[3 + i]) + Rvb;
    }
    float Rb[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) Rb[3 * i + j] = FMA(v[i], zb[j], Fwb[i] * A.Fb[j]);
    float qb[4];
    qb[0] = FMA(0.5f, FMA(dqb[3], om[2], FMA(dqb[2], om[1], dqb[1] * om[0])), qtb[0]);
    qb[1] = FMA(0.5f, FMA(dqb[3], om[1], FMA(-dqb[2], om[2], -(dqb[0] * om[0]))), qtb[1]);
    qb[2] = FMA(0.5f, FMA(-dqb[3], om[0], FMA(dqb[1], om[2], -(dqb[0] * om[1]))), qtb[2]);
    qb[3] = FMA(0.5f, FMA(dqb[2], om[0], FMA(-dqb[1], om[1], -(dqb[0] * om[2]))), qtb[3]);
    omb[0] = FMA(0.5f, FMA(-dqb[3], qy, FMA(dqb[2], qz, FMA(dqb[1], qw, -(dqb[0] * qx)))), omb[0]);
    omb[1] = FMA(0.5f, FMA(dqb[3], qx, FMA(dqb[2], qw, FMA(-dqb[1], qz, -(dqb[0] * qy)))), omb[1]);
    omb[2] = FMA(0.5f, FMA(dqb[3], qw, FMA(-dqb[2], qx, FMA(dqb[1], qy, -(dqb[0] * qz)))), omb[2]);
    float s01 = Rb[1] + Rb[3], d10 = Rb[3] - Rb[1];
    float s02 = Rb[2] + Rb[6], d02 = Rb[2] - Rb[6];
    float s12 = Rb[5] + Rb[7], d21 = Rb[7] - Rb[5];
    qb[0] = FMA(2.0f, FMA(qx, d21, FMA(qy, d02, qz * d10)), qb[0]);
    qb[1] = FMA(2.0f, FMA(qw, d21, FMA(qz, s02, qy * s01)), FMA(-4.0f * qx, Rb[4] + Rb[8], qb[1]));
    qb[2] = FMA(2.0f, FMA(qz, s12, FMA(qw, d02, qx * s01)), FMA(-4.0f * qy, Rb[0] + Rb[8], qb[2]));
    qb[3] = FMA(2.0f, FMA(qy, s12, FMA(qx, s02, qw * d10)), FMA(-4.0f * qz, Rb[0] + Rb[4], qb[3]));
#pragma unroll
    for (int i = 0; i < 3; ++i) { lam[i] = L[i]; lam[3 + i] = vbar[i]; lam[10 + i] = omb[i]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) lam[6 + i] = qb[i];
}

template <int M>
DI void step_vjp(const KArgs& a, const Smem& sm, const WaveW& ww, int t, int h, int lane, const float* x, const float* xi, const StepAux& A,
                 const float* L, float etabar_cost, float* lam, float* gq) {
    VjpTmp T;
    vjp_head<M>(a, sm, t, x, xi, A, L, etabar_cost, T, gq);
    SCHED_PHASE();
    float zb[NN];
    vjp_mlp_tiles<M>(sm, h, lane, A, T, zb, gq);
    vjp_tail(sm, t, x, A, L, T, zb, lam);
}

// SPEC.md §6.1 over per-group totals kept in global memory (a.part, one row of part_stride(H) floats per particle group):
// slot s = g mod 4 accumulates S_s <- S_s + T_g in ascending g starting from 0, total ((S0+S1)+S2)+S3. The per-group rows
// live in HBM/L2 instead of four LDS slot arrays so that a workgroup's LDS footprint does not grow with 4*(H+1)*13 floats:
// long horizons keep two or three workgroups per CU, and any wave may process any group (no slot ownership).
DI float group_ordered_sum(const float* rows, int G, int PS, int i) {
    float S0 = 0.0f, S1 = 0.0f, S2 = 0.0f, S3 = 0.0f;
    for (int g = 0; g < G; g += 4) {
        S0 = S0 + rows[(size_t)g * PS + i];
        if (g + 1 < G) S1 = S1 + rows[(size_t)(g + 1) * PS + i];
        if (g + 2 < G) S2 = S2 + rows[(size_t)(g + 2) * PS + i];
        if (g + 3 < G) S3 = S3 + rows[(size_t)(g + 3) * PS + i];
    }
    return ((S0 + S1) + S2) + S3;
}

// ================================================================================================
// Single-particle path (P == 1 — every MPC YAML the reference ships: launch/*_mpc.yaml `num_particles: 1`).
// With one particle the 32-column MFMA tiles would carry 31 idle columns, so the MLPs are laid out "one hidden unit per
// lane" instead: lanes 0..31 hold drift-net unit k = lane, lanes 32..63 density-net unit k = lane - 32; weights live in VGPRs.
//   layer 1      : 6 fma per lane (chain k = 0..5 from the C operand, as the MFMA does)
//   tanh         : SPEC.md §3.4 groups units 4g..4g+3 = one DPP quad: the four (1 + exp) values are exchanged with quad_perm
//                  broadcasts, every lane forms the shared reciprocal and keeps its own quotient (36 instructions per layer
//                  instead of 272 per tile)
//   layer 2      : 32 x (v_readlane of unit k, fma with this lane's W2 row), visiting k in the SPEC.md §4 order
//   output layers: the 14 half-chains (6 outputs + density, halves h = 0/1) run on 14 lanes at once; ds_bpermute gathers the
//                  unit each chain needs at step r, `row_ror:8` adds the two halves
//   adjoint      : the same three patterns transposed (readlane chain for W2^T, 12 + 2m half-chains for z-bar / gu-bar)
// Every value is produced by the same operation sequence as in the tile layout, so results are bit-identical to it and to
// the oracle. State, rigid body and cost are wave-uniform and reuse fwd_head / fwd_tail / vjp_head / vjp_tail.
// ================================================================================================
DI int koff(int r) { return (r & 3) + 8 * (r >> 2); }   // rowmap(r, 0)
DI float readlane_f(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
DI float bperm_f(int src_lane, float v) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane << 2, __builtin_bit_cast(int, v))); }

struct LaneW {
    float w1[NN];        // W1z[32*hh + k][0..5]
    float c1n;           // b1[32 + k] (C operand of the density rows; drift rows take c_t[k] from LDS)
    float b2k;           // b2[k]
    float w2row[HID];    // W2[k][0..31]   (layer 2, output unit k)
    float w2col[HID];    // W2[0..31][k]   (adjoint, input unit k)
    float w3col[6];      // W3[0..5][k]
    float w3nk;          // w3n[k]
    float wo[16];        // this lane's output half-chain: W3[c][koff(r) + 4 hs] (lanes c + 8 hs, c < 6), w3n[..] (c == 6), else 0
    float wz[32];        // this lane's adjoint half-chain: positions 0..15 density units, 16..31 drift units
    int obase, zbase;    // first source lane of the chains (4 hs, +32 for the density output chain)
    bool is_u;           // gu-bar chain (lanes 16..31): skips the density positions
};

DI void load_lane_weights(const KArgs& a, LaneW& W, int lane) {
    const float* w = a.wts;
    const int k = lane & 31, hh = lane >> 5, row = 32 * hh + k;
#pragma unroll
    for (int j = 0; j < NN; ++j) W.w1[j] = w[OFF_W1Z + row * NN + j];
    W.c1n = w[OFF_B1 + HID + k];
    W.b2k = w[OFF_B2 + k];
#pragma unroll
    for (int i = 0; i < HID; ++i) { W.w2row[i] = w[OFF_W2 + k * HID + i]; W.w2col[i] = w[OFF_W2 + i * HID + k]; }
#pragma unroll
    for (int i = 0; i < 6; ++i) W.w3col[i] = w[OFF_W3 + i * HID + k];
    W.w3nk = w[OFF_W3N + k];
    const int c = lane & 7, hs = (lane >> 3) & 1;
    const bool row0 = lane < 16, row1 = lane >= 16 && lane < 32;
    W.obase = (row0 && c == 6) ? 32 + 4 * hs : 4 * hs;
    W.zbase = 4 * hs;
    W.is_u = row1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int unit = koff(r) + 4 * hs;
        float v = 0.0f;
        if (row0 && c < 6) v = w[OFF_W3 + c * HID + unit];
        if (row0 && c == 6) v = w[OFF_W3N + unit];
        W.wo[r] = v;
        float zd = 0.0f, zf = 0.0f;
        if (row0 && c < 6) { zd = w[OFF_W1Z + (HID + unit) * NN + c]; zf = w[OFF_W1Z + unit * NN + c]; }
        if (row1 && c < a.m) zf = w[OFF_W1U + unit * 8 + c];
        W.wz[r] = zd; W.wz[16 + r] = zf;
    }
}

// SPEC.md §3.4 tanh4 with the four values of a group in the four lanes of a DPP quad
DI float lane_tanh(float av, int lane) {
    const float d = 1.0f + exp2_spec(clampf(av, -9.0f, 9.0f), 2.885390043258667f);
    const float d0 = dpp_f<0x00>(d), d1 = dpp_f<0x55>(d), d2 = dpp_f<0xAA>(d), d3 = dpp_f<0xFF>(d);
    const float p2 = d0 * d1, p3 = p2 * d2, p4 = p3 * d3;
    float r = rcp_spec(p4);
    const float r3 = r * p3; r = r * d3;
    const float r2 = r * p2; r = r * d2;
    const float r1 = r * d0;
    const float r0 = r * d1;
    const int q = lane & 3;
    const float rq = (q & 2) ? ((q & 1) ? r3 : r2) : ((q & 1) ? r1 : r0);
    return FMA(-2.0f, rq, 1.0f);
}

// forward MLPs of one step; h1: drift (lanes 0..31) / density (32..63) hidden unit, h2: layer-2 unit (both halves)
DI void lane_fwd_mlp(const KArgs& a, const LaneW& W, const float* ust, int lane, const float* z, float& h1, float& h2, float* o, float& eta) {
    const int k = lane & 31, hh = lane >> 5;
    float a1 = hh ? W.c1n : ust[k];
#pragma unroll
    for (int j = 0; j < NN; ++j) a1 = FMA(W.w1[j], z[j], a1);
    h1 = lane_tanh(a1, lane);
    float a2 = W.b2k;
#pragma unroll
    for (int r = 0; r < 16; r += 4) {      // eight lanes are read ahead of their fma chain (a read right before its use costs a wait state)
        float sv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) sv[e] = readlane_f(h1, rowmap(r + (e >> 1), e & 1));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 8; ++e) a2 = FMA(W.w2row[rowmap(r + (e >> 1), e & 1)], sv[e], a2);
        __builtin_amdgcn_sched_barrier(0);
    }
    h2 = lane_tanh(a2, lane);
    const float Mreg = hh ? h1 : h2;     // lanes 0..31: layer-2 activations, lanes 32..63: density hidden units
    float P = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) P = FMA(W.wo[r], bperm_f(W.obase + koff(r), Mreg), P);
    const float Pc = P + dpp_f<0x128>(P);   // row_ror:8 -> lane c: P_0 + P_1
#pragma unroll
    for (int i = 0; i < 6; ++i) o[i] = readlane_f(Pc, i) + a.M.b3[i];
    eta = sigmoid_spec(readlane_f(Pc, 6) + a.M.b3n);
}

// adjoint of the MLPs: zb[6], gq[0..M-1]
template <int M>
DI void lane_vjp_mlp(const LaneW& W, int lane, float h1, float h2, const VjpTmp& T, float* zb, float* gq) {
    const int hh = lane >> 5;
    float hb = 0.0f;
#pragma unroll
    for (int i = 0; i < 6; ++i) hb = FMA(W.w3col[i], T.ob[i], hb);
    const float a2b = hb * FMA(-h2, h2, 1.0f);
    float accB = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; r += 4) {
        float sv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) sv[e] = readlane_f(a2b, rowmap(r + (e >> 1), e & 1));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 8; ++e) accB = FMA(W.w2col[rowmap(r + (e >> 1), e & 1)], sv[e], accB);
        __builtin_amdgcn_sched_barrier(0);
    }
    const float g1 = FMA(-h1, h1, 1.0f);
    const float ad = accB * g1;
    const float an = (W.w3nk * T.ebraw) * g1;
    const float Abar = hh ? an : ad;
    float Pz = 0.0f;
#pragma unroll
    for (int p = 0; p < 16; ++p) {       // density units first (z-bar chains only)
        const float src = bperm_f(32 + W.zbase + koff(p), Abar);
        const float nv = FMA(W.wz[p], src, Pz);
        Pz = W.is_u ? Pz : nv;
    }
#pragma unroll
    for (int p = 0; p < 16; ++p) Pz = FMA(W.wz[16 + p], bperm_f(W.zbase + koff(p), Abar), Pz);
    const float Pc = Pz + dpp_f<0x128>(Pz);
#pragma unroll
    for (int kk = 0; kk < NN; ++kk) zb[kk] = readlane_f(Pc, kk);
#pragma unroll
    for (int jj = 0; jj < M; ++jj) gq[jj] = readlane_f(Pc, 16 + jj);
}

constexpr int LANE_ACT_H1 = 0, LANE_ACT_H2 = 64, LANE_ACT_SC = 128, LANE_ACT_X = 136;   // offsets inside one checkpoint row
constexpr int COOP_ROW = 160;      // floats per (particle, step) checkpoint row of the cooperative path: h1[64] h2[64] scalars[8] x_t[13] pad

// Where one particle's streams live (the same device functions serve the P == 1 team and the cooperative multi-workgroup path)
struct LaneIO {
    const float* x0;          // [13]
    const float* nz;          // noise: element (t, i) at nz[(t*6 + i) * 32]
    float* xs; int xs_t, xs_i;   // x_t kept for the adjoint / traj output: element (t, i) at xs[t*xs_t + i*xs_i]
    float* ck; int ck_t;      // checkpoint rows: row t at ck + t*ck_t
    float* out; int os;       // per-particle outputs: quantity q at out[q*os]  (q: t*12+k adjoint sums, t*13+i states, PS-1 cost)
    bool add0;                // P == 1: store v + 0.0f (what the SPEC.md §6.1 butterfly over 31 zero lanes leaves)
};
// Cooperative path: the handed-off values are written and read with agent-scope (sc1) accesses, so the grid barrier needs no
// L2 write-back / invalidate (the per-XCD L2s are not coherent with each other; a release fence would flush every dirty line of
// the checkpoint stream as well). SDEMPC_COOP_FENCE=1 builds the fence-based variant instead (A/B).
#ifndef SDEMPC_COOP_FENCE
#define SDEMPC_COOP_FENCE 0
#endif
DI void out_store(const LaneIO& io, size_t q, float v) {
    if (io.add0) io.out[q] = v + 0.0f;                 // P == 1 team (os == 1)
    else if (SDEMPC_COOP_FENCE) io.out[q * io.os] = v;
    else __hip_atomic_store(io.out + q * io.os, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
DI float coop_load(const float* p) {
    if (SDEMPC_COOP_FENCE) return *p;
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one particle: rollout and cost; store_x: stream x_t to io.xs, want_mean: x_t to io.out
DI void lane_particle_rollout(const KArgs& a, const Smem& sm, const LaneW& W, const LaneIO& io, int lane, bool store_x, bool want_mean) {
    const int H = a.H, PS = part_stride(H);
    float x[NX], xn[NX], xi[NN];
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = io.x0[i];
#pragma unroll
    for (int i = 0; i < NN; ++i) xi[i] = io.nz[i * 32];
    if (lane == 0) {
        if (store_x) {
#pragma unroll
            for (int i = 0; i < NX; ++i) io.xs[i * io.xs_i] = x[i];
        }
        if (want_mean) {
#pragma unroll
            for (int i = 0; i < NX; ++i) out_store(io, i, x[i]);
        }
    }
    float J = 0.0f;
    StepAux A;
    for (int t = 0; t < H; ++t) {
        float xin[NN];
        if (t + 1 < H) {
#pragma unroll
            for (int i = 0; i < NN; ++i) xin[i] = io.nz[((t + 1) * NN + i) * 32];
        }
        const float* ust = sm.ust + t * UST;
        float z[NN], h1, h2, o[6], eta;
        fwd_head(x, A.Rm, z);
        lane_fwd_mlp(a, W, ust, lane, z, h1, h2, o, eta);
        fwd_tail(a, sm, ust, t, x, xi, A.Rm, o, eta, xn, A);
        float l = stage_cost<false>(a, xn, sm.xref + (t + 1) * NX, nullptr);
        l = FMA(a.C.res_mult * A.eta, A.eta, l);
        J = FMA(sm.disc[t], l, J);
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = xn[i];
        if (t + 1 < H) {
#pragma unroll
            for (int i = 0; i < NN; ++i) xi[i] = xin[i];
        }
        if (lane == 0) {
            if (store_x) {
                float* tp = io.xs + (size_t)(t + 1) * io.xs_t;
#pragma unroll
                for (int i = 0; i < NX; ++i) tp[i * io.xs_i] = x[i];
            }
            if (want_mean) {
#pragma unroll
                for (int i = 0; i < NX; ++i) out_store(io, (t + 1) * NX + i, x[i]);
            }
        }
    }
    if (lane == 0) out_store(io, PS - 1, J);
}

// one particle: cost, forward sweep with checkpoint, adjoint sweep; per-step adjoint outputs gq[0..M+3] -> io.out
template <int M>
DI void lane_particle_grad(const KArgs& a, const Smem& sm, const LaneW& W, const LaneIO& io, int lane) {
    const int H = a.H, PS = part_stride(H);
    constexpr int nq = M + 4;
    float x[NX], xn[NX], xi[NN];
    StepAux A;
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = io.x0[i];
#pragma unroll
    for (int i = 0; i < NN; ++i) xi[i] = io.nz[i * 32];
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NX; ++i) io.xs[i * io.xs_i] = x[i];
    }
    float J = 0.0f;
    for (int t = 0; t < H; ++t) {
        float xin[NN];
        if (t + 1 < H) {
#pragma unroll
            for (int i = 0; i < NN; ++i) xin[i] = io.nz[((t + 1) * NN + i) * 32];
        }
        const float* ust = sm.ust + t * UST;
        float z[NN], h1, h2, o[6], eta;
        fwd_head(x, A.Rm, z);
        lane_fwd_mlp(a, W, ust, lane, z, h1, h2, o, eta);
        fwd_tail(a, sm, ust, t, x, xi, A.Rm, o, eta, xn, A);
        {
            float* ap = io.ck + (size_t)t * io.ck_t;
            ap[LANE_ACT_H1 + lane] = h1;
            ap[LANE_ACT_H2 + lane] = h2;
            if (lane == 0) {
                *reinterpret_cast<float4*>(ap + LANE_ACT_SC) = make_float4(A.eta, A.Fb[0], A.Fb[1], A.Fb[2]);
                ap[LANE_ACT_SC + 4] = A.rn;
            }
        }
        float l = stage_cost<false>(a, xn, sm.xref + (t + 1) * NX, nullptr);
        l = FMA(a.C.res_mult * A.eta, A.eta, l);
        J = FMA(sm.disc[t], l, J);
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = xn[i];
        if (t + 1 < H) {
#pragma unroll
            for (int i = 0; i < NN; ++i) xi[i] = xin[i];
        }
        if (lane == 0) {
            float* tp = io.xs + (size_t)(t + 1) * io.xs_t;
#pragma unroll
            for (int i = 0; i < NX; ++i) tp[i * io.xs_i] = x[i];
        }
    }
    if (lane == 0) out_store(io, PS - 1, J);
    // ---- adjoint sweep (x holds x_H); loads of step t-1 are in flight while step t is processed ----
    float lam[NX], xt[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) lam[i] = 0.0f;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    float nh1, nh2, nrn, nxt[NX], nxi[NN];
    float4 ns4;
    auto issue_loads = [&](int t) {
        const float* ap = io.ck + (size_t)t * io.ck_t;
        nh1 = ap[LANE_ACT_H1 + lane];
        nh2 = ap[LANE_ACT_H2 + lane];
        ns4 = *reinterpret_cast<const float4*>(ap + LANE_ACT_SC);
        nrn = ap[LANE_ACT_SC + 4];
        const float* tp = io.xs + (size_t)t * io.xs_t;
#pragma unroll
        for (int i = 0; i < NX; ++i) nxt[i] = tp[i * io.xs_i];
#pragma unroll
        for (int i = 0; i < NN; ++i) nxi[i] = io.nz[(t * NN + i) * 32];
    };
    issue_loads(H - 1);
    for (int t = H - 1; t >= 0; --t) {
        const float h1 = nh1, h2 = nh2;
        A.eta = ns4.x; A.Fb[0] = ns4.y; A.Fb[1] = ns4.z; A.Fb[2] = ns4.w; A.rn = nrn;
#pragma unroll
        for (int i = 0; i < NX; ++i) xt[i] = nxt[i];
#pragma unroll
        for (int i = 0; i < NN; ++i) xi[i] = nxi[i];
        if (t > 0) issue_loads(t - 1);
        const float dsc = sm.disc[t];
        {
            float gx[NX];
            stage_cost<true>(a, x, sm.xref + (t + 1) * NX, gx);
#pragma unroll
            for (int i = 0; i < NX; ++i) lam[i] = FMA(dsc, gx[i], lam[i]);
        }
        float zdummy[NN];
        fwd_head(xt, A.Rm, zdummy);
#pragma unroll
        for (int i = 0; i < 3; ++i) A.Jom[i] = a.M.J[i] * xt[10 + i];
#pragma unroll
        for (int i = 0; i < 4; ++i) A.qn[i] = x[6 + i];
        const float ebc = dsc * ((2.0f * a.C.res_mult) * A.eta);
        float lamn[NX], gq[12], zb[NN];
        VjpTmp T;
        vjp_head<M>(a, sm, t, xt, xi, A, lam, ebc, T, gq);
        lane_vjp_mlp<M>(W, lane, h1, h2, T, zb, gq);
        vjp_tail(sm, t, xt, A, lam, T, zb, lamn);
#pragma unroll
        for (int i = 0; i < NX; ++i) { lam[i] = lamn[i]; x[i] = xt[i]; }
        if (lane == 0) {
#pragma unroll
            for (int kq = 0; kq < nq; ++kq) out_store(io, t * 12 + kq, gq[kq]);
        }
    }
}

// SPEC.md §5.5 gradient assembly from the particle sums S(t, k) (shared by the lane and cooperative teams)
template <class Team, int M, class SumF>
DI void assemble_gradient(const KArgs& a, const Smem& sm, const float* y, float* gout, int tid, SumF&& Ssum) {
    const int H = a.H, m = a.m, N = H * m;
    for (int e = tid; e < N; e += Team::NT) {
        int t = e / m, jj = e - t * m;
        float S[5];
        int idx[5] = {jj, M, M + 1, M + 2, M + 3};
#pragma unroll
        for (int kq = 0; kq < 5; ++kq) S[kq] = Ssum(t * 12 + idx[kq]);
        float uj = y[e];
        float dT = FMA(2.0f * a.M.ct2, uj, a.M.ct1);
        float dM = a.M.dir[jj] * FMA(2.0f * a.M.cm2, uj, a.M.cm1);
        float acc = S[0];
        acc = FMA(S[1], dT, acc);
        acc = FMA(S[2], a.M.ry[jj] * dT, acc);
        acc = FMA(S[3], -(a.M.rx[jj] * dT), acc);
        acc = FMA(S[4], dM, acc);
        float du = uj - a.C.uref[jj];
        float dw = 0.0f, ctmp;
        if (t >= 1) dw = slew_dw(a, y, t, jj, m, ctmp);
        float gcu = sm.disc[t] * FMA(2.0f * a.C.uerr, du, dw);
        if (t + 1 < H) { float dwn = slew_dw(a, y, t + 1, jj, m, ctmp); gcu = FMA(-sm.disc[t + 1], dwn, gcu); }
        gout[e] = FMA(acc, a.invP, gcu);
    }
}

// ---- P == 1 team: one wave per instance ----
DI LaneIO lane_io_p1(const KArgs& a, int b) {
    const int H = a.H, PS = part_stride(H);
    LaneIO io;
    io.x0 = a.x0 + (size_t)b * NX;
    io.nz = a.noise + ((size_t)b * H) * NN * 32;              // particle 0 sits in column 0 of the 32-wide rows
    io.xs = a.traj + ((size_t)b * (H + 1)) * NX * 32; io.xs_t = NX * 32; io.xs_i = 32;
    io.ck = a.act + ((size_t)b * H) * ACT_STRIDE; io.ck_t = ACT_STRIDE;
    io.out = a.part + (size_t)b * PS; io.os = 1;
    io.add0 = true;
    return io;
}
template <class Team>
DI float lane_rollout(const KArgs& a, const Smem& sm, const LaneW& W, const float* u, int b, int tid, bool store_traj, float* xmean_out) {
    b = opaque_s(b); tid = opaque_v(tid);
    const int H = a.H, lane = tid & 63, PS = part_stride(H);
    const bool want_mean = xmean_out != nullptr;
    Team::sync();
    block_prepass<Team>(a, sm, u, tid);
    float cu = block_ucost<Team>(a, sm, u, tid);
    const LaneIO io = lane_io_p1(a, b);
    lane_particle_rollout(a, sm, W, io, lane, store_traj, want_mean);
    Team::sync();
    const float tot = group_ordered_sum(io.out, 1, PS, PS - 1);
    if (want_mean)
        for (int i = tid; i < (H + 1) * NX; i += Team::NT) xmean_out[i] = group_ordered_sum(io.out, 1, PS, i) * a.invP;
    return FMA(tot, a.invP, cu);
}
template <class Team, int M>
DI float lane_cost_grad(const KArgs& a, const Smem& sm, const LaneW& W, const float* y, float* gout, int b, int tid) {
    b = opaque_s(b); tid = opaque_v(tid);
    const int H = a.H, lane = tid & 63, PS = part_stride(H);
    Team::sync();
    block_prepass<Team>(a, sm, y, tid);
    float cu = block_ucost<Team>(a, sm, y, tid);
    const LaneIO io = lane_io_p1(a, b);
    lane_particle_grad<M>(a, sm, W, io, lane);
    Team::sync();
    const float tot = group_ordered_sum(io.out, 1, PS, PS - 1);
    assemble_gradient<Team, M>(a, sm, y, gout, tid, [&](int q) { return group_ordered_sum(io.out, 1, PS, q); });
    Team::sync();
    return FMA(tot, a.invP, cu);
}

// ================================================================================================
// Cooperative latency path: ONE instance spread over ceil(P/4) workgroups, one particle per wave in the lane layout, so a
// step costs ~560 instructions per wave instead of ~1,300 + 22 MFMAs for a 32-particle tile. Every workgroup runs the
// optimiser redundantly on identical data (deterministic, so all copies agree and take the same branches); the only
// exchange is the per-particle outputs of a rollout, written particle-minor to a global array, followed by one grid barrier
// per rollout; each workgroup then applies the SPEC.md §6.1 butterflies and slot order itself (bit-identical to the tile path).
// Launched only when all workgroups of the batch are co-resident (B * ceil(P/4) <= number of CUs); every spin is bounded.
// ================================================================================================
struct CoopCtx {
    int nwg, wgi, Ppad;
    unsigned* bar;          // this instance's arrival counter (zeroed by the host before the launch), bar[1] = error flag
    unsigned epoch;
    float* pp;              // [2][PS][Ppad] per-particle outputs, double-buffered by rollout parity
    float* ck;              // [P][H+1][COOP_ROW] checkpoint rows
};
constexpr unsigned COOP_SPIN_LIMIT = 8u * 1000u * 1000u;     // polls of one barrier before giving up (several seconds)

DI void coop_barrier(CoopCtx& C, int tid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's (sc1) stores of the handed-off values have completed
    __syncthreads();
    C.epoch += 1;
    if (tid == 0) {
        if (SDEMPC_COOP_FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(C.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = C.epoch * (unsigned)C.nwg;
        unsigned spins = 0;
        while (__hip_atomic_load(C.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (__hip_atomic_load(C.bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;      // another workgroup gave up
            __builtin_amdgcn_s_sleep(1);
            if (++spins > COOP_SPIN_LIMIT) { __hip_atomic_store(C.bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
        if (SDEMPC_COOP_FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// total of quantity q over the particles: SPEC.md §6.1 (32-particle butterflies, slots g mod 4 in ascending g, ((S0+S1)+S2)+S3);
// a wave reduces two groups per pass (lanes 0..31 group g, lanes 32..63 group g + 1). Result valid in every lane.
DI float coop_total(const float* pq, int P, int G, int lane) {
    const int hh = lane >> 5, j = lane & 31;
    float Sa = 0.0f, Sb = 0.0f;                       // lower half: slots 0 / 2, upper half: slots 1 / 3
    for (int g0 = 0; g0 < G; g0 += 4) {               // four groups per chunk: both loads are in flight together
        const int pa = 32 * (g0 + hh) + j, pb = 32 * (g0 + 2 + hh) + j;
        const float va = ((g0 + hh < G) && pa < P) ? coop_load(pq + pa) : 0.0f;
        const float vb = ((g0 + 2 + hh < G) && pb < P) ? coop_load(pq + pb) : 0.0f;
        Sa = Sa + group_bfly32(va);
        if (g0 + 2 < G) Sb = Sb + group_bfly32(vb);
    }
    const float S0 = readlane_f(Sa, 0), S1 = readlane_f(Sa, 32), S2 = readlane_f(Sb, 0), S3 = readlane_f(Sb, 32);
    return ((S0 + S1) + S2) + S3;
}

DI LaneIO lane_io_coop(const KArgs& a, const CoopCtx& C, int b, int p) {
    const int H = a.H;
    LaneIO io;
    io.x0 = a.x0 + (size_t)b * NX;
    io.nz = a.noise + ((size_t)(b * a.G + (p >> 5)) * H) * NN * 32 + (p & 31);
    io.ck = C.ck + (size_t)p * (H + 1) * COOP_ROW; io.ck_t = COOP_ROW;     // H + 1 rows: row t also carries x_t, t = 0..H
    io.xs = io.ck + LANE_ACT_X; io.xs_t = COOP_ROW; io.xs_i = 1;
    io.out = C.pp + (size_t)(C.epoch & 1u) * part_stride(H) * C.Ppad + p; io.os = C.Ppad;
    io.add0 = false;
    return io;
}

template <class Team>
DI float coop_rollout(const KArgs& a, const Smem& sm, const LaneW& W, CoopCtx& C, const float* u, int b, int tid, float* xmean_out) {
    b = opaque_s(b); tid = opaque_v(tid);
    const int H = a.H, P = a.P, G = a.G, lane = tid & 63, wave = tid >> 6, PS = part_stride(H);
    const bool want_mean = xmean_out != nullptr;
    Team::sync();
    block_prepass<Team>(a, sm, u, tid);
    float cu = block_ucost<Team>(a, sm, u, tid);
    const int p = C.wgi * 4 + wave;
    const float* pbuf = C.pp + (size_t)(C.epoch & 1u) * PS * C.Ppad;
    if (p < P) {
        const LaneIO io = lane_io_coop(a, C, b, p);
        lane_particle_rollout(a, sm, W, io, lane, false, want_mean);
    }
    coop_barrier(C, tid);
    if (wave == 0) { const float t0 = coop_total(pbuf + (size_t)(PS - 1) * C.Ppad, P, G, lane); if (lane == 0) sm.red[12] = t0; }
    if (want_mean && C.wgi == 0) {      // the mean trajectory is an output only: one workgroup writes it
        for (int q = wave; q < (H + 1) * NX; q += 4) {
            const float s = coop_total(pbuf + (size_t)q * C.Ppad, P, G, lane);
            if (lane == 0) xmean_out[q] = s * a.invP;
        }
    }
    Team::sync();
    return FMA(sm.red[12], a.invP, cu);
}

template <class Team, int M>
DI float coop_cost_grad(const KArgs& a, const Smem& sm, const LaneW& W, CoopCtx& C, const float* y, float* gout, int b, int tid) {
    b = opaque_s(b); tid = opaque_v(tid);
    const int H = a.H, P = a.P, G = a.G, lane = tid & 63, wave = tid >> 6, PS = part_stride(H);
    constexpr int nq = M + 4;
    Team::sync();
    block_prepass<Team>(a, sm, y, tid);
    float cu = block_ucost<Team>(a, sm, y, tid);
    const int p = C.wgi * 4 + wave;
    const float* pbuf = C.pp + (size_t)(C.epoch & 1u) * PS * C.Ppad;
    if (p < P) {
        const LaneIO io = lane_io_coop(a, C, b, p);
        lane_particle_grad<M>(a, sm, W, io, lane);
    }
    coop_barrier(C, tid);
    // particle sums of the nq adjoint outputs of every step -> LDS (wave w takes the steps t = w, w + 4, ...)
    if (wave == 0) { const float t0 = coop_total(pbuf + (size_t)(PS - 1) * C.Ppad, P, G, lane); if (lane == 0) sm.red[12] = t0; }
    // all nq sums of a step are reduced together: the 2 x nq loads of a pass are independent and in flight at once (a load that
    // crosses XCDs takes about a microsecond; one at a time they would dominate the gradient evaluation)
    {
        const int hh = lane >> 5, j = lane & 31;
        for (int t = wave; t < H; t += 4) {
            float Sa[nq], Sb[nq];
#pragma unroll
            for (int kq = 0; kq < nq; ++kq) { Sa[kq] = 0.0f; Sb[kq] = 0.0f; }
            for (int g0 = 0; g0 < G; g0 += 4) {          // two passes (four groups) per chunk
                float v0[nq], v1[nq];
                const int pa = 32 * (g0 + hh) + j, pb = 32 * (g0 + 2 + hh) + j;
                const bool oka = (g0 + hh < G) && pa < P, okb = (g0 + 2 + hh < G) && pb < P;
#pragma unroll
                for (int kq = 0; kq < nq; ++kq) {
                    const float* pq = pbuf + (size_t)(t * 12 + kq) * C.Ppad;
                    v0[kq] = oka ? coop_load(pq + pa) : 0.0f;
                    v1[kq] = okb ? coop_load(pq + pb) : 0.0f;
                }
#pragma unroll
                for (int kq = 0; kq < nq; ++kq) {
                    Sa[kq] = Sa[kq] + group_bfly32(v0[kq]);                     // groups g0, g0+1 -> slots 0 / 1
                    if (g0 + 2 < G) Sb[kq] = Sb[kq] + group_bfly32(v1[kq]);     // groups g0+2, g0+3 -> slots 2 / 3
                }
            }
#pragma unroll
            for (int kq = 0; kq < nq; ++kq) {
                const float S0 = readlane_f(Sa[kq], 0), S1 = readlane_f(Sa[kq], 32), S2 = readlane_f(Sb[kq], 0), S3 = readlane_f(Sb[kq], 32);
                if (lane == 0) sm.tot[t * 12 + kq] = ((S0 + S1) + S2) + S3;
            }
        }
    }
    Team::sync();
    const float tot = sm.red[12];
    assemble_gradient<Team, M>(a, sm, y, gout, tid, [&](int q) { return sm.tot[q]; });
    Team::sync();
    return FMA(tot, a.invP, cu);
}

// ------------------------------------------------------------------------------------------------
// block-level rollout: expected cost of control sequence u (LDS). SPEC.md §5.3/§6/§7
//   store_traj: stream x_t to a.traj; want_mean: particle mean trajectory -> xmean_out (global)
// ------------------------------------------------------------------------------------------------
template <class Team, bool F16, bool PK = false>
DI float block_rollout(const KArgs& a, const Smem& sm, const WaveW& ww, const float* u, int b, int tid, bool store_traj, float* xmean_out) {
    b = opaque_s(b); tid = opaque_v(tid);
    const int H = a.H, G = a.G, P = a.P;
    const int lane = tid & 63, j = lane & 31, h = lane >> 5, wave = tid >> 6;   // wave index inside the team
    const bool want_mean = xmean_out != nullptr;
    Team::sync();
    block_prepass<Team>(a, sm, u, tid);
    const int PS = part_stride(H);
    float* prows = a.part + (size_t)b * G * PS;
    float cu = block_ucost<Team>(a, sm, u, tid);  // contains barriers: prepass results visible afterwards
    float x0r[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) x0r[i] = a.x0[b * NX + i];
    for (int g = wave; g < G; g += Team::NWAVES) {
        const bool valid = (g * 32 + j) < P;
        const float* nz = a.noise + ((size_t)(b * G + g) * H) * NN * 32 + j;
        float* tj = a.traj + ((size_t)(b * G + g) * (H + 1)) * NX * 32 + j;
        float* xm = prows + (size_t)g * PS;           // this group's row of per-step particle sums (SPEC.md §6.1/§6.3)
        float x[NX], xn[NX], xi[NN];
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = x0r[i];
#pragma unroll
        for (int i = 0; i < NN; ++i) xi[i] = nz[i * 32];
        if (store_traj) {
#pragma unroll
            for (int c = 0; c < 7; ++c) { if (c + 7 * h < NX) tj[(c + 7 * h) * 32] = h ? x[(c + 7) % NX] : x[c]; }
        }
        if (want_mean) {
#pragma unroll
            for (int i = 0; i < NX; ++i) { float s = group_bfly32(valid ? x[i] : 0.0f); if (lane == 0) xm[i] = s; }
        }
        float J = 0.0f;
        StepAux A;
        for (int t = 0; t < H; ++t) {
            float xin[NN];
            if (t + 1 < H) {
#pragma unroll
                for (int i = 0; i < NN; ++i) xin[i] = nz[((t + 1) * NN + i) * 32];
            }
            step_fwd<F16, PK>(a, sm, ww, t, h, lane, x, xi, xn, A);
            float l = stage_cost<false>(a, xn, sm.xref + (t + 1) * NX, nullptr);
            l = FMA(a.C.res_mult * A.eta, A.eta, l);
            J = FMA(sm.disc[t], l, J);
#pragma unroll
            for (int i = 0; i < NX; ++i) x[i] = xn[i];
            if (t + 1 < H) {
#pragma unroll
                for (int i = 0; i < NN; ++i) xi[i] = xin[i];
            }
            if (store_traj) {
                float* tp = tj + (size_t)(t + 1) * NX * 32;
#pragma unroll
                for (int c = 0; c < 7; ++c) { if (c + 7 * h < NX) tp[(c + 7 * h) * 32] = h ? x[(c + 7) % NX] : x[c]; }
            }
            if (want_mean) {
#pragma unroll
                for (int i = 0; i < NX; ++i) { float s = group_bfly32(valid ? x[i] : 0.0f); if (lane == 0) xm[(t + 1) * NX + i] = s; }
            }
        }
        float T = group_bfly32(valid ? J : 0.0f);
        if (lane == 0) xm[PS - 1] = T;                // group total of the particle costs (last element of the group's row)
    }
    Team::sync();
    const float tot = group_ordered_sum(prows, G, PS, PS - 1);
    if (want_mean) {
        for (int i = tid; i < (H + 1) * NX; i += Team::NT) xmean_out[i] = group_ordered_sum(prows, G, PS, i) * a.invP;
    }
    return FMA(tot, a.invP, cu);
}

// ------------------------------------------------------------------------------------------------
// block-level cost + gradient (forward sweep with trajectory store, adjoint sweep). SPEC.md §5.4/§6
//   y: control sequence in LDS; gout: gradient [H*m] in LDS
// ------------------------------------------------------------------------------------------------
// PREF: software-prefetch the adjoint sweep's loads one step ahead through a register double buffer (40 VGPRs). Needed when a
// SIMD holds one or two waves; the throughput instantiation drops it to fit three waves per SIMD, which hide the latency instead.
template <class Team, int M, bool F16, bool PK = false, bool PREF = true>
DI float block_cost_grad(const KArgs& a, const Smem& sm, const WaveW& ww, const float* y, float* gout, int b, int tid) {
    b = opaque_s(b); tid = opaque_v(tid);
    const int H = a.H, G = a.G, P = a.P, m = a.m;
    constexpr int nq = M + 4;
    const int lane = tid & 63, j = lane & 31, h = lane >> 5, wave = tid >> 6;   // wave index inside the team
    Team::sync();
    block_prepass<Team>(a, sm, y, tid);
    const int PS = part_stride(H);
    float* prows = a.part + (size_t)b * G * PS;
    float cu = block_ucost<Team>(a, sm, y, tid);
    float x0r[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) x0r[i] = a.x0[b * NX + i];
    for (int g = wave; g < G; g += Team::NWAVES) {
        const bool valid = (g * 32 + j) < P;
        const float* nz = a.noise + ((size_t)(b * G + g) * H) * NN * 32 + j;
        float* tj = a.traj + ((size_t)(b * G + g) * (H + 1)) * NX * 32 + j;
        float* ac = a.act + ((size_t)(b * G + g) * H) * ACT_STRIDE;
        float* Sq = prows + (size_t)g * PS;          // this group's row of per-step adjoint sums (SPEC.md §6.1)
        float x[NX], xn[NX], xi[NN];
        StepAux A;
        // ---- forward sweep, x_t and the hidden activations streamed to HBM ----
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = x0r[i];
#pragma unroll
        for (int i = 0; i < NN; ++i) xi[i] = nz[i * 32];
#pragma unroll
        for (int c = 0; c < 7; ++c) { if (c + 7 * h < NX) tj[(c + 7 * h) * 32] = h ? x[(c + 7) % NX] : x[c]; }
        float J = 0.0f;
        for (int t = 0; t < H; ++t) {
            float xin[NN];
            if (t + 1 < H) {
#pragma unroll
                for (int i = 0; i < NN; ++i) xin[i] = nz[((t + 1) * NN + i) * 32];
            }
            step_fwd<F16, PK>(a, sm, ww, t, h, lane, x, xi, xn, A);
            {   // activation checkpoint: second hidden layer (4 x 16-byte stores per lane) + step scalars once per particle;
                // the adjoint sweep recomputes only layer 1 from x_t (balance between HBM traffic and vector work)
                float* ap = ac + (size_t)t * ACT_STRIDE;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(ap + (q * 64 + lane) * 4) = make_float4(A.h2[4 * q], A.h2[4 * q + 1], A.h2[4 * q + 2], A.h2[4 * q + 3]);
                if (h == 0) {
                    *reinterpret_cast<float4*>(ap + 1024 + j * 8) = make_float4(A.eta, A.Fb[0], A.Fb[1], A.Fb[2]);
                    ap[1024 + j * 8 + 4] = A.rn;
                }
#if SDEMPC_CKPT1 >= 1
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(ap + 1280 + (q * 64 + lane) * 4) = make_float4(A.h1d[4 * q], A.h1d[4 * q + 1], A.h1d[4 * q + 2], A.h1d[4 * q + 3]);
#endif
#if SDEMPC_CKPT1 >= 2
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(ap + 2304 + (q * 64 + lane) * 4) = make_float4(A.h1n[4 * q], A.h1n[4 * q + 1], A.h1n[4 * q + 2], A.h1n[4 * q + 3]);
#endif
            }
            float l = stage_cost<false>(a, xn, sm.xref + (t + 1) * NX, nullptr);
            l = FMA(a.C.res_mult * A.eta, A.eta, l);
            J = FMA(sm.disc[t], l, J);
#pragma unroll
            for (int i = 0; i < NX; ++i) x[i] = xn[i];
            if (t + 1 < H) {
#pragma unroll
                for (int i = 0; i < NN; ++i) xi[i] = xin[i];
            }
            float* tp = tj + (size_t)(t + 1) * NX * 32;
#pragma unroll
            for (int c = 0; c < 7; ++c) { if (c + 7 * h < NX) tp[(c + 7 * h) * 32] = h ? x[(c + 7) % NX] : x[c]; }
        }
        { const float T = group_bfly32(valid ? J : 0.0f); if (lane == 0) Sq[PS - 1] = T; }
        // ---- adjoint sweep: x (registers) currently holds x_H ----
        float lam[NX], xt[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) lam[i] = 0.0f;
        // this wave's own stores of x_t / activations must be visible to its loads (same CU: workgroup scope)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        // software pipeline: the loads of step t-1 are issued while step t is being processed
        float4 nh[4], ns4;
#if SDEMPC_CKPT1 >= 1
        float4 nh1[4];
#endif
#if SDEMPC_CKPT1 >= 2
        float4 nh1n[4];
#endif
        float nrn, nxt[NX], nxi[NN];
        auto issue_loads = [&](int t) {
            const float* ap = ac + (size_t)t * ACT_STRIDE;
#pragma unroll
            for (int q = 0; q < 4; ++q) nh[q] = *reinterpret_cast<const float4*>(ap + (q * 64 + lane) * 4);
#if SDEMPC_CKPT1 >= 1
#pragma unroll
            for (int q = 0; q < 4; ++q) nh1[q] = *reinterpret_cast<const float4*>(ap + 1280 + (q * 64 + lane) * 4);
#endif
#if SDEMPC_CKPT1 >= 2
#pragma unroll
            for (int q = 0; q < 4; ++q) nh1n[q] = *reinterpret_cast<const float4*>(ap + 2304 + (q * 64 + lane) * 4);
#endif
            ns4 = *reinterpret_cast<const float4*>(ap + 1024 + j * 8);
            nrn = ap[1024 + j * 8 + 4];
            const float* tp = tj + (size_t)t * NX * 32;
#pragma unroll
            for (int i = 0; i < NX; ++i) nxt[i] = tp[i * 32];
#pragma unroll
            for (int i = 0; i < NN; ++i) nxi[i] = nz[(t * NN + i) * 32];
        };
        if constexpr (PREF) issue_loads(H - 1);
        for (int t = H - 1; t >= 0; --t) {
            if constexpr (!PREF) issue_loads(t);
            // take ownership of the prefetched step
            f32x16 h2l;
#pragma unroll
            for (int q = 0; q < 4; ++q) { h2l[4 * q] = nh[q].x; h2l[4 * q + 1] = nh[q].y; h2l[4 * q + 2] = nh[q].z; h2l[4 * q + 3] = nh[q].w; }
#if SDEMPC_CKPT1 >= 1
            f32x16 h1dl;
#pragma unroll
            for (int q = 0; q < 4; ++q) { h1dl[4 * q] = nh1[q].x; h1dl[4 * q + 1] = nh1[q].y; h1dl[4 * q + 2] = nh1[q].z; h1dl[4 * q + 3] = nh1[q].w; }
#endif
#if SDEMPC_CKPT1 >= 2
            f32x16 h1nl;
#pragma unroll
            for (int q = 0; q < 4; ++q) { h1nl[4 * q] = nh1n[q].x; h1nl[4 * q + 1] = nh1n[q].y; h1nl[4 * q + 2] = nh1n[q].z; h1nl[4 * q + 3] = nh1n[q].w; }
#endif
            const float eta_l = ns4.x, fb0 = ns4.y, fb1 = ns4.z, fb2 = ns4.w, rn_l = nrn;
#pragma unroll
            for (int i = 0; i < NX; ++i) xt[i] = nxt[i];
#pragma unroll
            for (int i = 0; i < NN; ++i) xi[i] = nxi[i];
            if constexpr (PREF) { if (t > 0) issue_loads(t - 1); }
            // x = x_{t+1}: fold the stage-cost gradient into the incoming adjoint
            const float dsc = sm.disc[t];
            {
                float gx[NX];
                stage_cost<true>(a, x, sm.xref + (t + 1) * NX, gx);
#pragma unroll
                for (int i = 0; i < NX; ++i) lam[i] = FMA(dsc, gx[i], lam[i]);
            }
            SCHED_PHASE();
            float lamn[NX], gq[12];
            // recompute layer 1 only (R, v_body, 6 MFMAs, 32 tanh); everything downstream of it comes from the checkpoint
            // (the unused remainder of step_fwd is dead code and is removed by the compiler)
            step_fwd<F16, PK>(a, sm, ww, t, h, lane, xt, xi, xn, A);
#if SDEMPC_CKPT1 >= 1
            A.h1d = h1dl;      // checkpointed: the drift tile's MFMAs and tanh in step_fwd above become dead code
#endif
#if SDEMPC_CKPT1 >= 2
            A.h1n = h1nl;
#endif
            A.h2 = h2l; A.eta = eta_l; A.Fb[0] = fb0; A.Fb[1] = fb1; A.Fb[2] = fb2; A.rn = rn_l;
#pragma unroll
            for (int i = 0; i < 3; ++i) A.Jom[i] = a.M.J[i] * xt[10 + i];
#pragma unroll
            for (int i = 0; i < 4; ++i) A.qn[i] = x[6 + i];   // q_{t+1}
            float ebc = dsc * ((2.0f * a.C.res_mult) * A.eta);
            step_vjp<M>(a, sm, ww, t, h, lane, xt, xi, A, lam, ebc, lamn, gq);
#pragma unroll
            for (int i = 0; i < NX; ++i) { lam[i] = lamn[i]; x[i] = xt[i]; }
            // particle sums of the nq per-step adjoint outputs: both lane halves hold the same values, so the lower
            // half reduces value k and the upper half value k + nq/2 in one butterfly (halves never mix below xor 32)
            {
                constexpr int half = (nq + 1) / 2;
#pragma unroll
                for (int k = 0; k < half; ++k) {
                    const float lo = gq[k], hi = (k + half < nq) ? gq[k + half] : 0.0f;
                    float s = group_bfly32(valid ? (h ? hi : lo) : 0.0f);
                    if (j == 0 && (h == 0 || k + half < nq)) {
                        Sq[t * 12 + k + h * half] = s;
                    }
                }
            }
        }
    }
    Team::sync();
    const float tot = group_ordered_sum(prows, G, PS, PS - 1);
    // gradient assembly (SPEC.md §6.3)
    const int N = H * m;
    for (int e = tid; e < N; e += Team::NT) {
        int t = e / m, jj = e - t * m;
        float S[5];
        int idx[5] = {jj, M, M + 1, M + 2, M + 3};
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            S[k] = group_ordered_sum(prows, G, PS, t * 12 + idx[k]);
        }
        float uj = y[e];
        float dT = FMA(2.0f * a.M.ct2, uj, a.M.ct1);
        float dM = a.M.dir[jj] * FMA(2.0f * a.M.cm2, uj, a.M.cm1);
        float acc = S[0];
        acc = FMA(S[1], dT, acc);
        acc = FMA(S[2], a.M.ry[jj] * dT, acc);
        acc = FMA(S[3], -(a.M.rx[jj] * dT), acc);
        acc = FMA(S[4], dM, acc);
        // control-cost gradient
        float du = uj - a.C.uref[jj];
        float dw = 0.0f, ctmp;
        if (t >= 1) dw = slew_dw(a, y, t, jj, m, ctmp);
        float gcu = sm.disc[t] * FMA(2.0f * a.C.uerr, du, dw);
        if (t + 1 < H) { float dwn = slew_dw(a, y, t + 1, jj, m, ctmp); gcu = FMA(-sm.disc[t + 1], dwn, gcu); }
        gout[e] = FMA(acc, a.invP, gcu);
    }
    Team::sync();
    return FMA(tot, a.invP, cu);
}

// MODE 0: tile layout (32 particles per wave); 1: single-particle lane layout (P == 1); 2: cooperative lane layout (one particle per
// wave, one instance over several workgroups)
template <class Team, bool F16, bool PK, int MODE>
DI float team_rollout(const KArgs& a, const Smem& sm, const WaveW& ww, const LaneW& LW, CoopCtx& CC, const float* u, int b, int tid, bool store_traj, float* xmean_out) {
    if constexpr (MODE == 2) return coop_rollout<Team>(a, sm, LW, CC, u, b, tid, xmean_out);
    else if constexpr (MODE == 1) return lane_rollout<Team>(a, sm, LW, u, b, tid, store_traj, xmean_out);
    else return block_rollout<Team, F16, PK>(a, sm, ww, u, b, tid, store_traj, xmean_out);
}
template <class Team, int M, bool F16, bool PK, bool PREF, int MODE>
DI float team_cost_grad(const KArgs& a, const Smem& sm, const WaveW& ww, const LaneW& LW, CoopCtx& CC, const float* y, float* gout, int b, int tid) {
    if constexpr (MODE == 2) return coop_cost_grad<Team, M>(a, sm, LW, CC, y, gout, b, tid);
    else if constexpr (MODE == 1) return lane_cost_grad<Team, M>(a, sm, LW, y, gout, b, tid);
    else return block_cost_grad<Team, M, F16, PK, PREF>(a, sm, ww, y, gout, b, tid);
}

template <class Team>
DI void load_common(const KArgs& a, const Smem& sm, int b, int tid) {
    for (int i = tid; i < (a.H + 1) * NX; i += Team::NT) sm.xref[i] = a.xref[(size_t)b * (a.H + 1) * NX + i];
}

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
// Common kernel prologue: carve LDS, stage weights (whole workgroup), then each team takes its instance.
#define SDEMPC_KERNEL_PROLOGUE()                                                     \
    extern __shared__ __attribute__((aligned(16))) float smem[];                     \
    LaneW LW;                                                                        \
    CoopCtx CC;                                                                      \
    const int tid = Team::tid();                                                     \
    int b_;                                                                          \
    if constexpr (MODE == 2) {                                                       \
        b_ = blockIdx.x / a.coop_nwg;                                                \
        CC.nwg = a.coop_nwg; CC.wgi = blockIdx.x - b_ * a.coop_nwg; CC.Ppad = a.G * 32; CC.epoch = 0u;            \
        CC.bar = a.coop_bar + 2 * b_;                                                \
        CC.pp = a.coop_pp + (size_t)b_ * 2 * part_stride(a.H) * CC.Ppad;             \
        CC.ck = a.coop_ck + (size_t)b_ * a.P * (a.H + 1) * COOP_ROW;                 \
    } else {                                                                         \
        b_ = blockIdx.x * Team::IPB + Team::team();                                  \
    }                                                                                \
    const int b = __builtin_amdgcn_readfirstlane(b_);                                \
    Smem sm = carve(smem, a.H, a.m, Team::team(), MODE == 2);                        \
    WaveW ww;                                                                        \
    load_weights(a, sm, ww, threadIdx.x, Team::BNT);                                 \
    __syncthreads();                                                                 \
    if (b >= a.B) return; /* no workgroup-wide barrier below this line in TeamWave */ \
    if constexpr (MODE != 0) load_lane_weights(a, LW, threadIdx.x & 63);             \
    load_common<Team>(a, sm, b, tid);

template <class Team, bool F16, int MODE = 0>
__global__ void __launch_bounds__(Team::BNT, (MODE ? 2 : 3)) sdempc_rollout_kernel(KArgs a) {
    SDEMPC_KERNEL_PROLOGUE();
    const int N = a.H * a.m;
    for (int e = tid; e < N; e += Team::NT) sm.v[5][e] = a.u[(size_t)b * N + e];
    float c = team_rollout<Team, F16, false, MODE>(a, sm, ww, LW, CC, sm.v[5], b, tid, a.store_traj != 0, a.xmean ? a.xmean + (size_t)b * (a.H + 1) * NX : nullptr);
    if (tid == 0) a.cost[b] = c;
}

template <class Team, int M, bool F16, int MODE = 0>
__global__ void __launch_bounds__(Team::BNT, (MODE ? 2 : 3)) sdempc_grad_kernel(KArgs a) {
    SDEMPC_KERNEL_PROLOGUE();
    const int N = a.H * a.m;
    for (int e = tid; e < N; e += Team::NT) sm.v[5][e] = a.u[(size_t)b * N + e];
    float c = team_cost_grad<Team, M, F16, false, false, MODE>(a, sm, ww, LW, CC, sm.v[5], sm.v[3], b, tid);   // tiles: three waves per SIMD, no prefetch buffer
    if (tid == 0) a.cost[b] = c;
    for (int e = tid; e < N; e += Team::NT) a.grad[(size_t)b * N + e] = sm.v[3][e];
}

// SPEC.md §8: monotone accelerated proximal gradient with Armijo backtracking, one instance per block
// PK: packed-f32 tanh, for launches that leave one wave per SIMD (see tanh8_pk); results are bit-identical either way
// Occupancy: the throughput instantiation of the workgroup-wide team is built for three waves per SIMD (168 VGPRs: the hot loops
// fit, the compiler spills only solver state around them; +6 % at C2 over two waves per SIMD with the prefetch buffer). The
// latency instantiation (PK) and the one-wave teams (LDS allows two workgroups per CU anyway) keep two.
template <class Team, bool PK> constexpr int solve_waves_per_simd() { return PK ? 2 : 3; }
// MODE 1 / 2: lane layouts (weights in VGPRs, two waves per SIMD; MODE 2 with PK: one workgroup per CU, spills go to AGPRs)
template <class Team, int M, bool F16, bool PK = false, int MODE = 0>
__global__ void __launch_bounds__(Team::BNT, (MODE == 2 && PK ? 1 : MODE ? 2 : solve_waves_per_simd<Team, PK>())) sdempc_solve_kernel(KArgs a) {
    SDEMPC_KERNEL_PROLOGUE();
    const int m = a.m, N = a.H * m;
    float *xk = sm.v[0], *yk = sm.v[1], *xn = sm.v[2], *g = sm.v[3], *d1 = sm.v[4], *d2 = sm.v[5];
    for (int e = tid; e < N; e += Team::NT) {
        int jj = e % m;
        float v = clampf(a.u[(size_t)b * N + e], a.C.ulo[jj], a.C.uhi[jj]);
        xk[e] = v; yk[e] = v;
    }
    const float c_init = team_rollout<Team, F16, PK, MODE>(a, sm, ww, LW, CC, xk, b, tid, false, nullptr);
    float c_x = c_init, s = a.stepsize_in[b], gsq = 0.0f, sum_ls = 0.0f, sum_s = 0.0f;
    int kr = 0, noimp = 0, nit = 0, nls_tot = 0, plain = 1;
    for (int k = 0; k < a.A.max_iter; ++k) {
        const float c_y = team_cost_grad<Team, M, F16, PK, solve_waves_per_simd<Team, PK>() == 2, MODE>(a, sm, ww, LW, CC, yk, g, b, tid);
        gsq = block_dot<Team>(sm, g, g, N, tid);
        if (!(gsq < __builtin_inff())) break;   // SPEC.md §8 non-finite guard (team-uniform): keep xk, report gsq
        float c_n = 0.0f;
        int nls = 0;
        if (a.A.maxls > 0) {
            if (k > 0 && a.A.reset_inc) s = s * a.A.inc;
            if (s > a.A.smax) s = a.A.smax;
            for (int jl = 0; jl < a.A.maxls; ++jl) {
                Team::sync();
                for (int e = tid; e < N; e += Team::NT) {
                    int jj = e % m;
                    float v = clampf(FMA(-s, g[e], yk[e]), a.C.ulo[jj], a.C.uhi[jj]);
                    xn[e] = v; d1[e] = v - yk[e];
                }
                c_n = team_rollout<Team, F16, PK, MODE>(a, sm, ww, LW, CC, xn, b, tid, false, nullptr);
                float gd = block_dot<Team>(sm, g, d1, N, tid);
                nls = jl + 1;
                if (c_n <= FMA(a.A.coef, gd, c_y)) break;
                if (jl < a.A.maxls - 1) s = s * a.A.dec;
            }
        } else {
            s = a.A.stepsize;
            Team::sync();
            for (int e = tid; e < N; e += Team::NT) { int jj = e % m; xn[e] = clampf(FMA(-s, g[e], yk[e]), a.C.ulo[jj], a.C.uhi[jj]); }
            c_n = team_rollout<Team, F16, PK, MODE>(a, sm, ww, LW, CC, xn, b, tid, false, nullptr);
            nls = 1;
        }
        sum_ls = sum_ls + (float)nls; sum_s = sum_s + s; nit = k + 1; nls_tot += nls;
        int stop = (__builtin_fabsf(c_n - c_x) <= FMA(a.A.rtol, __builtin_fabsf(c_x), a.A.atol));
        Team::sync();
        if (c_n < c_x) {
            for (int e = tid; e < N; e += Team::NT) { d1[e] = yk[e] - xn[e]; d2[e] = xn[e] - xk[e]; }
            float rs = block_dot<Team>(sm, d1, d2, N, tid);
            if (rs > 0.0f) {
                kr = 0; plain = 1;
                for (int e = tid; e < N; e += Team::NT) { yk[e] = xn[e]; xk[e] = xn[e]; }
            } else {
                float bt = a.beta[kr];
                for (int e = tid; e < N; e += Team::NT) { int jj = e % m; yk[e] = clampf(FMA(bt, d2[e], xn[e]), a.C.ulo[jj], a.C.uhi[jj]); xk[e] = xn[e]; }
                kr = kr + 1; plain = 0;
            }
            c_x = c_n; noimp = 0;
        } else {
            if (!plain) stop = 0;
            kr = 0; plain = 1;
            for (int e = tid; e < N; e += Team::NT) yk[e] = xk[e];
            noimp = noimp + 1;
        }
        if (noimp >= a.A.max_noimp) stop = 1;
        if (stop) break;
    }
    Team::sync();
    if (MODE != 2 || CC.wgi == 0)
        for (int e = tid; e < N; e += Team::NT) a.uopt[(size_t)b * N + e] = xk[e];
    team_rollout<Team, F16, PK, MODE>(a, sm, ww, LW, CC, xk, b, tid, false, a.xmean + (size_t)b * (a.H + 1) * NX);
    if (tid == 0 && (MODE != 2 || CC.wgi == 0)) {
        float* inf = a.info + (size_t)b * 8;
        const float fn = (float)nit;
        inf[0] = nit ? sum_ls / fn : 0.0f; inf[1] = s; inf[2] = fn; inf[3] = gsq; inf[4] = nit ? sum_s / fn : 0.0f;
        inf[5] = c_init; inf[6] = c_x; inf[7] = (float)nls_tot;
        if constexpr (MODE == 2) {   // a grid barrier gave up (never seen in testing; bounded so that a fault cannot hang the GPU): poison the telemetry
            if (__hip_atomic_load(CC.bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
                for (int i = 0; i < 8; ++i) inf[i] = __builtin_nanf("");
        }
    }
}

// ================================================================================================
// Speculative cooperative solve (smallest batches): FIVE groups of ceil(P/4) workgroups per instance. While groups 0 and 1
// evaluate the first two line-search trials of iteration k side by side, groups 2, 3 and 4 already evaluate the gradient of
// iteration k+1 at the three points the optimiser can move to: where it goes if it ends on trial 1 resp. trial 2 with an
// improvement, and xk (no improvement). The step sizes of the trials are known before any of them is evaluated (s, s*dec, ...),
// so are the restart tests. One grid barrier per phase. The optimiser itself is unchanged and runs redundantly in every
// workgroup: a gradient is a pure function of its point, so using the pre-computed one (when the speculation hits — the trial
// costs decide that) gives the same bits as computing it afterwards; on a miss the iteration falls back to the sequential order.
// Written as a state machine with ONE call site of the particle work, so that the rollout and the gradient sweep are each
// instantiated once (the first version inlined them at seven sites: 140 KB of code, 2x slower sweeps).
// ================================================================================================
constexpr int SPEC_GROUPS = 5, SPEC_SLOTS = 7;
constexpr int SLOT_SEQ = 6, SLOT_GRAD = 5;        // slots 0..4: the five items of a parallel phase

DI LaneIO lane_io_slot(const KArgs& a, const CoopCtx& C, int b, int p, unsigned par, int slot) {
    const int H = a.H;
    LaneIO io;
    io.x0 = a.x0 + (size_t)b * NX;
    io.nz = a.noise + ((size_t)(b * a.G + (p >> 5)) * H) * NN * 32 + (p & 31);
    io.ck = C.ck + (size_t)p * (H + 1) * COOP_ROW; io.ck_t = COOP_ROW;
    io.xs = io.ck + LANE_ACT_X; io.xs_t = COOP_ROW; io.xs_i = 1;
    io.out = C.pp + (size_t)(par * SPEC_SLOTS + slot) * part_stride(H) * C.Ppad + p; io.os = C.Ppad;
    io.add0 = false;
    return io;
}
// after a phase's barrier, in every workgroup: expected cost of control sequence u whose particle outputs sit in (par, slot)
DI float spec_cost(const KArgs& a, const Smem& sm, const CoopCtx& C, int tid, unsigned par, const float* u, int slot) {
    const int lane = tid & 63, wave = tid >> 6, PS = part_stride(a.H);
    const float cu = block_ucost<TeamBlock>(a, sm, u, tid);
    const float* pbuf = C.pp + (size_t)(par * SPEC_SLOTS + slot) * PS * C.Ppad;
    __syncthreads();
    if (wave == 0) { const float t0 = coop_total(pbuf + (size_t)(PS - 1) * C.Ppad, a.P, a.G, lane); if (lane == 0) sm.red[12] = t0; }
    __syncthreads();
    return FMA(sm.red[12], a.invP, cu);
}
// gradient at y from the adjoint sums in (par, slot)
template <int M>
DI void spec_gradient(const KArgs& a, const Smem& sm, const CoopCtx& C, int tid, unsigned par, const float* y, int slot, float* gout) {
    const int lane = tid & 63, wave = tid >> 6, PS = part_stride(a.H), H = a.H, P = a.P, G = a.G;
    constexpr int nq = M + 4;
    const float* pbuf = C.pp + (size_t)(par * SPEC_SLOTS + slot) * PS * C.Ppad;
    const int hh = lane >> 5, j = lane & 31;
    __syncthreads();
    for (int t = wave; t < H; t += 4) {
        float Sa[nq], Sb[nq];
#pragma unroll
        for (int kq = 0; kq < nq; ++kq) { Sa[kq] = 0.0f; Sb[kq] = 0.0f; }
        for (int g0 = 0; g0 < G; g0 += 4) {
            float v0[nq], v1[nq];
            const int pa = 32 * (g0 + hh) + j, pb = 32 * (g0 + 2 + hh) + j;
            const bool oka = (g0 + hh < G) && pa < P, okb = (g0 + 2 + hh < G) && pb < P;
#pragma unroll
            for (int kq = 0; kq < nq; ++kq) {
                const float* pq = pbuf + (size_t)(t * 12 + kq) * C.Ppad;
                v0[kq] = oka ? coop_load(pq + pa) : 0.0f;
                v1[kq] = okb ? coop_load(pq + pb) : 0.0f;
            }
#pragma unroll
            for (int kq = 0; kq < nq; ++kq) {
                Sa[kq] = Sa[kq] + group_bfly32(v0[kq]);
                if (g0 + 2 < G) Sb[kq] = Sb[kq] + group_bfly32(v1[kq]);
            }
        }
#pragma unroll
        for (int kq = 0; kq < nq; ++kq) {
            const float S0 = readlane_f(Sa[kq], 0), S1 = readlane_f(Sa[kq], 32), S2 = readlane_f(Sb[kq], 0), S3 = readlane_f(Sb[kq], 32);
            if (lane == 0) sm.tot[t * 12 + kq] = ((S0 + S1) + S2) + S3;
        }
    }
    __syncthreads();
    assemble_gradient<TeamBlock, M>(a, sm, y, gout, tid, [&](int q) { return sm.tot[q]; });
    __syncthreads();
}

// one workgroup per CU (512 registers per lane: what does not fit the 256 VGPRs spills to AGPRs, not to scratch memory — with two
// workgroups per CU the adjoint loop carried 43 scratch accesses per step and ran 3x slower)
template <int M>
__global__ void __launch_bounds__(256, 1) sdempc_solve_spec_kernel(KArgs a) {
    using Team = TeamBlock;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // a.coop_ngrp groups (2..5) per instance take the roles T1, T2, S(y2), S(xk), S(y1) in this order of usefulness
    // (measured at C2: the line search ends on trial 2 in 60 %, there is no improvement in 31 %, it ends on trial 1 in 28 % of the iterations)
    const int nwg = a.coop_nwg, ng = a.coop_ngrp, per = ng * nwg, H = a.H, m = a.m, N = H * m, PS = part_stride(H);
    const int b_ = blockIdx.x / per, r_ = blockIdx.x - b_ * per, grp = r_ / nwg;
    const bool have_y2 = ng >= 3, have_xk = ng >= 4, have_y1 = ng >= 5;
    const int grad_grp = ng >= 3 ? 2 : 0;          // who evaluates a gradient outside the parallel phase
    const int b = __builtin_amdgcn_readfirstlane(b_);
    CoopCtx C;
    C.nwg = per; C.wgi = r_ - grp * nwg; C.Ppad = a.G * 32; C.epoch = 0u;
    C.bar = a.coop_bar + 2 * b;
    C.pp = a.coop_pp + (size_t)b * ((size_t)2 * SPEC_SLOTS * PS * C.Ppad + 2 * (size_t)PS);
    C.ck = a.coop_ck + ((size_t)b * 3 + (grp >= 2 ? grp - 2 : 0)) * a.P * (H + 1) * COOP_ROW;    // gradients run on groups 2..4 (or 0 when there are only two)
    Smem sm = carve(smem, H, m, 0, true);
    WaveW ww;
    LaneW LW;
    load_weights(a, sm, ww, tid, Team::BNT);
    __syncthreads();
    if (b >= a.B) return;
    load_lane_weights(a, LW, lane);
    load_common<Team>(a, sm, b, tid);
    const int nv = (N + 3) & ~3;
    float* ex = sm.tot + ((H * 12 + 3) & ~3);
    float *xn1 = ex, *xn2 = ex + nv, *y1 = ex + 2 * nv, *y2 = ex + 3 * nv;
    float *xk = sm.v[0], *yk = sm.v[1], *xn = sm.v[2], *g = sm.v[3], *d1 = sm.v[4], *d2 = sm.v[5];
    for (int e = tid; e < N; e += Team::NT) {
        int jj = e % m;
        float v = clampf(a.u[(size_t)b * N + e], a.C.ulo[jj], a.C.uhi[jj]);
        xk[e] = v; yk[e] = v;
    }
    // PH_RED: the particle sums of a gradient (H*nq totals + the cost total) are reduced ONCE, spread over all workgroups of the
    // instance (one step each), published and read back after one more barrier — every workgroup reducing everything itself took
    // ~45 us per iteration, a third of the time of an iteration
    enum { PH_INIT, PH_GRAD, PH_PAR, PH_SEQ, PH_RED, PH_FINAL, PH_DONE };
    float* gtot_base = C.pp + (size_t)2 * SPEC_SLOTS * PS * C.Ppad;       // [2][PS] published totals, after the per-particle slots
    unsigned red_cnt = 0u, red_par = 0u; int red_slot = 0;
    int phase = PH_INIT;
    float c_init = 0.0f, c_x = 0.0f, s = a.stepsize_in[b], gsq = 0.0f, sum_ls = 0.0f, sum_s = 0.0f, c_y = 0.0f, c_n = 0.0f;
    int k = 0, kr = 0, noimp = 0, nit = 0, nls_tot = 0, plain = 1, nls = 0, jsel = 0, jl = 0;
    unsigned par_cnt = 0u, par_spec = 0u;
    bool spec = false, two = false;
    const bool has_ls = a.A.maxls > 0;
    while (phase != PH_DONE) {
        // ---- this workgroup's work item of the phase ----
        const float* iu = xk; int islot = SLOT_SEQ; bool igrad = false, imean = false, iact = false;
        unsigned par = C.epoch & 1u;
        if (phase == PH_RED) {
            constexpr int nq = M + 4;
            const float* pbuf = C.pp + (size_t)(red_par * SPEC_SLOTS + red_slot) * PS * C.Ppad;
            float* gt = gtot_base + (size_t)(red_cnt & 1u) * PS;
            for (int t = r_; t < H; t += per) {
                for (int kq = wave; kq < nq; kq += 4) {
                    const float sv = coop_total(pbuf + (size_t)(t * 12 + kq) * C.Ppad, a.P, a.G, lane);
                    if (lane == 0) __hip_atomic_store(gt + t * 12 + kq, sv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (r_ == per - 1 && wave == 3) {      // the cost total: last workgroup (idle above unless per <= H)
                const float sv = coop_total(pbuf + (size_t)(PS - 1) * C.Ppad, a.P, a.G, lane);
                if (lane == 0) __hip_atomic_store(gt + PS - 1, sv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        else if (phase == PH_INIT) { iact = grp == 0; }
        else if (phase == PH_GRAD) { iact = grp == grad_grp; iu = yk; islot = SLOT_GRAD; igrad = true; }
        else if (phase == PH_SEQ) { iact = grp == 0; iu = xn; }
        else if (phase == PH_FINAL) { iact = grp == 0; imean = true; }
        else {   // PH_PAR
            par = par_cnt & 1u; par_spec = par; par_cnt += 1u;
            if (grp == 0) { iact = true; iu = xn1; islot = 0; }
            else if (grp == 1) { iact = two; iu = xn2; islot = 1; }
            else if (grp == 2) { iact = spec && two; iu = y2; igrad = true; islot = 3; }     // when there is one trial only, y1 takes this group
            else if (grp == 3) { iact = spec; iu = xk; igrad = true; islot = 4; }
            else { iact = spec; iu = y1; igrad = true; islot = 2; }
            if (grp == 2 && spec && !two) { iact = true; iu = y1; igrad = true; islot = 2; }
        }
        if (iact) {      // the only call site of the particle work
            __syncthreads();
            block_prepass<Team>(a, sm, iu, tid);
            __syncthreads();
            const int p = C.wgi * 4 + wave;
            if (p < a.P) {
                const LaneIO io = lane_io_slot(a, C, b, p, par, islot);
                if (igrad) lane_particle_grad<M>(a, sm, LW, io, lane);
                else lane_particle_rollout(a, sm, LW, io, lane, false, imean);
            }
        }
        coop_barrier(C, tid);
        // ---- the optimiser (SPEC.md §8), advanced as far as the data of this phase allows ----
        bool head = false, tail = false;
        if (phase == PH_INIT) {
            c_init = spec_cost(a, sm, C, tid, par, xk, SLOT_SEQ);
            c_x = c_init;
            phase = a.A.max_iter > 0 ? PH_GRAD : PH_FINAL;
        } else if (phase == PH_GRAD) {
            red_slot = SLOT_GRAD; red_par = par; phase = PH_RED;
        } else if (phase == PH_RED) {
            constexpr int nq = M + 4;
            const float* gt = gtot_base + (size_t)(red_cnt & 1u) * PS;
            red_cnt += 1u;
            for (int q = tid; q < H * 12; q += Team::NT)
                if ((q % 12) < nq) sm.tot[q] = coop_load(gt + q);
            if (tid == 0) sm.red[12] = coop_load(gt + PS - 1);
            const float cu = block_ucost<Team>(a, sm, yk, tid);        // (contains the barriers that publish sm.tot / sm.red)
            __syncthreads();
            c_y = FMA(sm.red[12], a.invP, cu);
            assemble_gradient<Team, M>(a, sm, yk, g, tid, [&](int q) { return sm.tot[q]; });
            __syncthreads();
            head = true;
        } else if (phase == PH_PAR) {
            __syncthreads();
            for (int e = tid; e < N; e += Team::NT) { xn[e] = xn1[e]; d1[e] = xn1[e] - yk[e]; }
            c_n = spec_cost(a, sm, C, tid, par, xn1, 0);
            nls = 1; jsel = 1;
            bool done = !has_ls;
            if (has_ls) {
                const float gd = block_dot<Team>(sm, g, d1, N, tid);
                if (c_n <= FMA(a.A.coef, gd, c_y)) done = true;
                else if (0 < a.A.maxls - 1) s = s * a.A.dec;
            }
            if (!done && two) {
                __syncthreads();
                for (int e = tid; e < N; e += Team::NT) { xn[e] = xn2[e]; d1[e] = xn2[e] - yk[e]; }
                c_n = spec_cost(a, sm, C, tid, par, xn2, 1);
                const float gd = block_dot<Team>(sm, g, d1, N, tid);
                nls = 2; jsel = 2;
                if (c_n <= FMA(a.A.coef, gd, c_y)) done = true;
                else if (1 < a.A.maxls - 1) s = s * a.A.dec;
            }
            if (!done && 2 < a.A.maxls) {        // further trials one at a time
                jl = 2;
                __syncthreads();
                for (int e = tid; e < N; e += Team::NT) {
                    int jj = e % m;
                    float v = clampf(FMA(-s, g[e], yk[e]), a.C.ulo[jj], a.C.uhi[jj]);
                    xn[e] = v; d1[e] = v - yk[e];
                }
                __syncthreads();
                phase = PH_SEQ;
            } else tail = true;
        } else if (phase == PH_SEQ) {
            c_n = spec_cost(a, sm, C, tid, par, xn, SLOT_SEQ);
            const float gd = block_dot<Team>(sm, g, d1, N, tid);
            nls = jl + 1; jsel = 0;
            bool done = c_n <= FMA(a.A.coef, gd, c_y);
            if (!done && jl < a.A.maxls - 1) s = s * a.A.dec;
            if (!done && jl + 1 < a.A.maxls) {
                jl += 1;
                __syncthreads();
                for (int e = tid; e < N; e += Team::NT) {
                    int jj = e % m;
                    float v = clampf(FMA(-s, g[e], yk[e]), a.C.ulo[jj], a.C.uhi[jj]);
                    xn[e] = v; d1[e] = v - yk[e];
                }
                __syncthreads();
            } else tail = true;
        } else {   // PH_FINAL
            if (grp == 0 && C.wgi == 0) {
                const float* pbuf = C.pp + (size_t)(par * SPEC_SLOTS + SLOT_SEQ) * PS * C.Ppad;
                float* xmean_out = a.xmean + (size_t)b * (H + 1) * NX;
                for (int q = wave; q < (H + 1) * NX; q += 4) {
                    const float sv = coop_total(pbuf + (size_t)q * C.Ppad, a.P, a.G, lane);
                    if (lane == 0) xmean_out[q] = sv * a.invP;
                }
                for (int e = tid; e < N; e += Team::NT) a.uopt[(size_t)b * N + e] = xk[e];
                if (tid == 0) {
                    float* inf = a.info + (size_t)b * 8;
                    const float fn = (float)nit;
                    inf[0] = nit ? sum_ls / fn : 0.0f; inf[1] = s; inf[2] = fn; inf[3] = gsq; inf[4] = nit ? sum_s / fn : 0.0f;
                    inf[5] = c_init; inf[6] = c_x; inf[7] = (float)nls_tot;
                    if (__hip_atomic_load(C.bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
                        for (int i = 0; i < 8; ++i) inf[i] = __builtin_nanf("");
                }
            }
            phase = PH_DONE;
        }
        if (tail) {      // end of the line search of iteration k
            sum_ls = sum_ls + (float)nls; sum_s = sum_s + s; nit = k + 1; nls_tot += nls;
            int stop = (__builtin_fabsf(c_n - c_x) <= FMA(a.A.rtol, __builtin_fabsf(c_x), a.A.atol));
            int hit_slot = -1;
            __syncthreads();
            if (c_n < c_x) {
                for (int e = tid; e < N; e += Team::NT) { d1[e] = yk[e] - xn[e]; d2[e] = xn[e] - xk[e]; }
                float rs = block_dot<Team>(sm, d1, d2, N, tid);
                if (rs > 0.0f) {
                    kr = 0; plain = 1;
                    for (int e = tid; e < N; e += Team::NT) { yk[e] = xn[e]; xk[e] = xn[e]; }
                } else {
                    float bt = a.beta[kr];
                    for (int e = tid; e < N; e += Team::NT) { int jj = e % m; yk[e] = clampf(FMA(bt, d2[e], xn[e]), a.C.ulo[jj], a.C.uhi[jj]); xk[e] = xn[e]; }
                    kr = kr + 1; plain = 0;
                }
                c_x = c_n; noimp = 0;
                // the new yk is exactly y_jsel: was its gradient among the speculated ones?
                if (spec && jsel == 2 && have_y2) hit_slot = 3;
                if (spec && jsel == 1 && (have_y1 || (have_y2 && !two))) hit_slot = 2;
            } else {
                if (!plain) stop = 0;
                kr = 0; plain = 1;
                for (int e = tid; e < N; e += Team::NT) yk[e] = xk[e];
                noimp = noimp + 1;
                if (spec && have_xk) hit_slot = 4;               // the new yk is xk, unchanged since the parallel phase
            }
            if (noimp >= a.A.max_noimp) stop = 1;
            k += 1;
            if (stop || k >= a.A.max_iter) phase = PH_FINAL;
            else if (hit_slot >= 0) { red_slot = hit_slot; red_par = par_spec; phase = PH_RED; }
            else phase = PH_GRAD;
        }
        if (head) {      // start of iteration k with (c_y, g) in hand
            gsq = block_dot<Team>(sm, g, g, N, tid);
            if (!(gsq < __builtin_inff())) { phase = PH_FINAL; continue; }
            if (has_ls) {
                if (k > 0 && a.A.reset_inc) s = s * a.A.inc;
                if (s > a.A.smax) s = a.A.smax;
            } else {
                s = a.A.stepsize;
            }
            two = (has_ls ? a.A.maxls : 1) > 1;
            spec = (k + 1 < a.A.max_iter);
            const float s1 = s, s2 = s * a.A.dec;
            __syncthreads();
            for (int e = tid; e < N; e += Team::NT) {
                int jj = e % m;
                xn1[e] = clampf(FMA(-s1, g[e], yk[e]), a.C.ulo[jj], a.C.uhi[jj]);
                xn2[e] = clampf(FMA(-s2, g[e], yk[e]), a.C.ulo[jj], a.C.uhi[jj]);
            }
            __syncthreads();
            // where the optimiser moves if it ends on trial j with an improvement (the expressions of the tail above)
            for (int jtr = 0; jtr < (two ? 2 : 1); ++jtr) {
                const float* xj = jtr ? xn2 : xn1;
                float* yj = jtr ? y2 : y1;
                for (int e = tid; e < N; e += Team::NT) { d1[e] = yk[e] - xj[e]; d2[e] = xj[e] - xk[e]; }
                const float rs = block_dot<Team>(sm, d1, d2, N, tid);
                const float bt = a.beta[kr];
                for (int e = tid; e < N; e += Team::NT) { int jj = e % m; yj[e] = (rs > 0.0f) ? xj[e] : clampf(FMA(bt, d2[e], xj[e]), a.C.ulo[jj], a.C.uhi[jj]); }
                __syncthreads();
            }
            phase = PH_PAR;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static hipError_t set_smem_attr(const void* fn, size_t bytes) {
    if (bytes > 64 * 1024) return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    return hipSuccess;
}
// One wave per instance only when the whole instance is a single particle group AND four instances' LDS fits
// twice per CU (keeps 2 workgroups resident); long horizons fall back to the workgroup-wide team.
static bool use_wave_team(int G, int H, int m) { return G == 1 && smem_bytes(H, m, TeamWave::IPB) <= 80 * 1024; }
int team_ipb(int G, int H, int m) { return use_wave_team(G, H, m) ? TeamWave::IPB : TeamBlock::IPB; }

template <class Kern>
static hipError_t launch_k(Kern k, const KArgs& a, hipStream_t st, int ipb, int bnt = 256) {
    const size_t sb = smem_bytes(a.H, a.m, ipb);
    hipError_t e = set_smem_attr((const void*)k, sb);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3((a.B + ipb - 1) / ipb), dim3(bnt), sb, st, a);
    return hipGetLastError();
}
template <class Team>
static hipError_t launch_rollout_team(const KArgs& a, hipStream_t st) {
    return a.f16 ? launch_k(sdempc_rollout_kernel<Team, true>, a, st, Team::IPB) : launch_k(sdempc_rollout_kernel<Team, false>, a, st, Team::IPB);
}
template <class Team, bool F16>
static hipError_t launch_grad_team(const KArgs& a, hipStream_t st) {
    if (a.m == 4) return launch_k(sdempc_grad_kernel<Team, 4, F16>, a, st, Team::IPB);
    if (a.m == 6) return launch_k(sdempc_grad_kernel<Team, 6, F16>, a, st, Team::IPB);
    return launch_k(sdempc_grad_kernel<Team, 8, F16>, a, st, Team::IPB);
}
// Number of compute units of the current device (cached): a grid of at most that many workgroups leaves one wave per SIMD.
static int device_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
        else return 256;
    }
    return cus;
}
template <class Team, bool F16>
static hipError_t launch_solve_team(const KArgs& a, hipStream_t st) {
    if constexpr (!F16 && !FAST) {
        // small-batch (latency) launches: one workgroup per CU at most -> a lone wave per SIMD is issue-bound -> packed tanh
        const int wgs = (a.B + Team::IPB - 1) / Team::IPB;
        const char* force = getenv("SDEMPC_PK");            // "0" / "1": A/B switch for tools and tests (read per launch); unset: by grid size
        const bool pk = force ? force[0] == '1' : wgs <= device_cus();
        if (pk) {
            if constexpr (Team::IPB == 1) {
                if (a.G > 4) {   // more particle groups than the four waves of a workgroup: eight waves halve the sequential depth
                    if (a.m == 4) return launch_k(sdempc_solve_kernel<TeamBlock8, 4, false, true>, a, st, 1, TeamBlock8::BNT);
                    if (a.m == 6) return launch_k(sdempc_solve_kernel<TeamBlock8, 6, false, true>, a, st, 1, TeamBlock8::BNT);
                    return launch_k(sdempc_solve_kernel<TeamBlock8, 8, false, true>, a, st, 1, TeamBlock8::BNT);
                }
            }
            if (a.m == 4) return launch_k(sdempc_solve_kernel<Team, 4, false, true>, a, st, Team::IPB);
            if (a.m == 6) return launch_k(sdempc_solve_kernel<Team, 6, false, true>, a, st, Team::IPB);
            return launch_k(sdempc_solve_kernel<Team, 8, false, true>, a, st, Team::IPB);
        }
    }
    if (a.m == 4) return launch_k(sdempc_solve_kernel<Team, 4, F16>, a, st, Team::IPB);
    if (a.m == 6) return launch_k(sdempc_solve_kernel<Team, 6, F16>, a, st, Team::IPB);
    return launch_k(sdempc_solve_kernel<Team, 8, F16>, a, st, Team::IPB);
}
// Single-particle lane layout: exact f32 arithmetic only, one wave per instance (SDEMPC_LANE=0 forces the tile layout: A/B, tests)
static bool use_lane(const KArgs& k) {
    if (FAST || k.f16 || k.P != 1 || !use_wave_team(k.G, k.H, k.m)) return false;
    const char* force = getenv("SDEMPC_LANE");
    return !(force && force[0] == '0');
}
#if SDEMPC_FAST
static hipError_t launch_lane_m(int, const KArgs&, hipStream_t) { return hipErrorInvalidValue; }   // never selected (use_lane)
#else
template <int M>
static hipError_t launch_lane(int what, const KArgs& k, hipStream_t st) {
    if (what == 0) return launch_k(sdempc_rollout_kernel<TeamWave, false, 1>, k, st, TeamWave::IPB);
    if (what == 1) return launch_k(sdempc_grad_kernel<TeamWave, M, false, 1>, k, st, TeamWave::IPB);
    return launch_k(sdempc_solve_kernel<TeamWave, M, false, false, 1>, k, st, TeamWave::IPB);
}
static hipError_t launch_lane_m(int what, const KArgs& k, hipStream_t st) {
    if (k.m == 4) return launch_lane<4>(what, k, st);
    if (k.m == 6) return launch_lane<6>(what, k, st);
    return launch_lane<8>(what, k, st);
}
#endif
#if !SDEMPC_FAST
// ---- cooperative latency path (exact arithmetic only) ----
int coop_nwg(int P) { return (P + 3) / 4; }
int coop_max_instances(int P, int H, int m) {
    const char* force = getenv("SDEMPC_COOP");                  // "0" disables the path (A/B, tests; read per launch)
    if ((force && force[0] == '0') || P < 2) return 0;
    if (smem_bytes(H, m, 1, true) > 160 * 1024) return 0;
    // every workgroup of the grid must be resident at once: the kernel is built for two waves per SIMD, i.e. two workgroups per CU
    // (its LDS footprint allows more); a margin of 16 workgroups is left
    return (2 * device_cus() - 16) / coop_nwg(P);
}
// per-instance workspace, sized for the speculative variant (7 output slots, 3 checkpoint regions); the plain cooperative kernel
// uses a prefix of it
size_t coop_pp_floats(int H, int G) { return (size_t)2 * SPEC_SLOTS * part_stride(H) * G * 32 + 2 * (size_t)part_stride(H); }
size_t coop_ck_floats(int H, int P) { return (size_t)3 * P * (H + 1) * COOP_ROW; }
int spec_max_instances(int P, int H, int m) {
    const char* force = getenv("SDEMPC_SPEC");                  // "0" disables the speculative variant (A/B, tests; read per launch)
    const char* fc = getenv("SDEMPC_COOP");
    if ((force && force[0] == '0') || (fc && fc[0] == '0')) return 0;      // P == 1 is welcome here (one wave per workgroup is active)
    const size_t nv = (size_t)((H * m + 3) & ~3);
    if (smem_bytes(H, m, 1, true) + 4 * nv * sizeof(float) > 160 * 1024) return 0;
    return device_cus() / (2 * coop_nwg(P));                    // built for one workgroup per CU; at least two groups per instance
}
template <int M>
static hipError_t launch_spec_m(const KArgs& k, hipStream_t st) {
    auto kern = sdempc_solve_spec_kernel<M>;
    const size_t sb = smem_bytes(k.H, k.m, 1, true) + 4 * (size_t)((k.H * k.m + 3) & ~3) * sizeof(float);
    hipError_t e = set_smem_attr((const void*)kern, sb);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(k.B * k.coop_ngrp * k.coop_nwg), dim3(TeamBlock::BNT), sb, st, k);
    return hipGetLastError();
}
hipError_t launch_solve_spec(const KArgs& a, int B, hipStream_t st) {
    KArgs k = a; k.B = B; k.coop_nwg = coop_nwg(k.P);
    if (B < 1 || B > spec_max_instances(k.P, k.H, k.m) || !k.coop_bar || !k.coop_pp || !k.coop_ck) return hipErrorInvalidValue;
    k.coop_ngrp = device_cus() / (B * k.coop_nwg);
    if (k.coop_ngrp > SPEC_GROUPS) k.coop_ngrp = SPEC_GROUPS;
    if (k.m == 4) return launch_spec_m<4>(k, st);
    if (k.m == 6) return launch_spec_m<6>(k, st);
    return launch_spec_m<8>(k, st);
}
template <int M>
static hipError_t launch_coop_m(const KArgs& k, hipStream_t st) {
    const size_t sb = smem_bytes(k.H, k.m, 1, true);
    if (k.B * k.coop_nwg <= device_cus()) {      // one workgroup per CU: the 512-register build (no scratch spills in the sweeps)
        auto kern = sdempc_solve_kernel<TeamBlock, M, false, true, 2>;
        hipError_t e = set_smem_attr((const void*)kern, sb);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(k.B * k.coop_nwg), dim3(TeamBlock::BNT), sb, st, k);
        return hipGetLastError();
    }
    auto kern = sdempc_solve_kernel<TeamBlock, M, false, false, 2>;
    hipError_t e = set_smem_attr((const void*)kern, sb);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(k.B * k.coop_nwg), dim3(TeamBlock::BNT), sb, st, k);
    return hipGetLastError();
}
hipError_t launch_solve_coop(const KArgs& a, int B, hipStream_t st) {
    KArgs k = a; k.B = B; k.coop_nwg = coop_nwg(k.P);
    if (B < 1 || B > coop_max_instances(k.P, k.H, k.m) || !k.coop_bar || !k.coop_pp || !k.coop_ck) return hipErrorInvalidValue;
    if (k.m == 4) return launch_coop_m<4>(k, st);
    if (k.m == 6) return launch_coop_m<6>(k, st);
    return launch_coop_m<8>(k, st);
}
#endif

hipError_t launch_rollout(const KArgs& a, int B, hipStream_t st) {
    KArgs k = a; k.B = B;
    if (use_lane(k)) return launch_lane_m(0, k, st);
    return use_wave_team(k.G, k.H, k.m) ? launch_rollout_team<TeamWave>(k, st) : launch_rollout_team<TeamBlock>(k, st);
}
hipError_t launch_grad(const KArgs& a, int B, hipStream_t st) {
    KArgs k = a; k.B = B;
    if (use_lane(k)) return launch_lane_m(1, k, st);
    if (use_wave_team(k.G, k.H, k.m)) return k.f16 ? launch_grad_team<TeamWave, true>(k, st) : launch_grad_team<TeamWave, false>(k, st);
    return k.f16 ? launch_grad_team<TeamBlock, true>(k, st) : launch_grad_team<TeamBlock, false>(k, st);
}
hipError_t launch_solve(const KArgs& a, int B, hipStream_t st) {
    KArgs k = a; k.B = B;
    if (use_lane(k)) return launch_lane_m(2, k, st);
    if (use_wave_team(k.G, k.H, k.m)) return k.f16 ? launch_solve_team<TeamWave, true>(k, st) : launch_solve_team<TeamWave, false>(k, st);
    return k.f16 ? launch_solve_team<TeamBlock, true>(k, st) : launch_solve_team<TeamBlock, false>(k, st);
}

// ------------------------------------------------------------------------------------------------
// layout conversion between the canonical tensors of include/sdempc.h ([B][P][C], C = H*6 for the noise,
// (H+1)*13 for the particle x horizon tensor) and the particle-minor device layout [B][G][C][32] the
// kernels stream. HBM-bound transpose through a 32x33 LDS tile: both sides coalesced.
// Padded particles (p >= P) are written as zeros on the way in and skipped on the way out.
// ------------------------------------------------------------------------------------------------
template <bool TO_DEV>
__global__ void __launch_bounds__(256) sdempc_relayout_kernel(const float* __restrict__ in, float* __restrict__ out, int P, int G, int C) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, g = blockIdx.y, b = blockIdx.z;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    if (TO_DEV) {
        for (int r = ty; r < 32; r += 8) {                      // tile[particle][column]
            const int p = g * 32 + r, c = c0 + tx;
            tile[r][tx] = (p < P && c < C) ? in[((size_t)b * P + p) * C + c] : 0.0f;
        }
        __syncthreads();
        for (int r = ty; r < 32; r += 8) {
            const int c = c0 + r;
            if (c < C) out[(((size_t)b * G + g) * C + c) * 32 + tx] = tile[tx][r];
        }
    } else {
        for (int r = ty; r < 32; r += 8) {                      // tile[column][particle]
            const int c = c0 + r;
            tile[r][tx] = c < C ? in[(((size_t)b * G + g) * C + c) * 32 + tx] : 0.0f;
        }
        __syncthreads();
        for (int r = ty; r < 32; r += 8) {
            const int p = g * 32 + r, c = c0 + tx;
            if (p < P && c < C) out[((size_t)b * P + p) * C + c] = tile[tx][r];
        }
    }
}
hipError_t launch_relayout(bool to_dev, const float* in, float* out, int B, int P, int G, int C, hipStream_t st) {
    if (B < 1 || G < 1 || G > 65535 || C < 1 || P < 1 || P > G * 32) return hipErrorInvalidValue;
    const size_t canon = (size_t)P * C, dev = (size_t)G * C * 32;       // floats per instance on either side
    for (int b0 = 0; b0 < B; b0 += 65535) {                             // gridDim.z limit
        const int nb = B - b0 < 65535 ? B - b0 : 65535;
        dim3 grid((C + 31) / 32, G, nb);
        if (to_dev) sdempc_relayout_kernel<true><<<grid, 256, 0, st>>>(in + b0 * canon, out + b0 * dev, P, G, C);
        else sdempc_relayout_kernel<false><<<grid, 256, 0, st>>>(in + b0 * dev, out + b0 * canon, P, G, C);
    }
    return hipGetLastError();
}

#if SDEMPC_FAST
}  // namespace fastm
hipError_t launch_rollout_fast(const KArgs& a, int B, hipStream_t st) { return fastm::launch_rollout(a, B, st); }
hipError_t launch_grad_fast(const KArgs& a, int B, hipStream_t st) { return fastm::launch_grad(a, B, st); }
hipError_t launch_solve_fast(const KArgs& a, int B, hipStream_t st) { return fastm::launch_solve(a, B, st); }
#else
}  // namespace exact
size_t smem_bytes(int H, int m, int ipb) { return exact::smem_bytes(H, m, ipb); }
int team_ipb(int G, int H, int m) { return exact::team_ipb(G, H, m); }
hipError_t launch_rollout(const KArgs& a, int B, hipStream_t st) { return exact::launch_rollout(a, B, st); }
hipError_t launch_grad(const KArgs& a, int B, hipStream_t st) { return exact::launch_grad(a, B, st); }
hipError_t launch_solve(const KArgs& a, int B, hipStream_t st) { return exact::launch_solve(a, B, st); }
int coop_nwg(int P) { return exact::coop_nwg(P); }
int coop_max_instances(int P, int H, int m) { return exact::coop_max_instances(P, H, m); }
size_t coop_pp_floats(int H, int G) { return exact::coop_pp_floats(H, G); }
size_t coop_ck_floats(int H, int P) { return exact::coop_ck_floats(H, P); }
hipError_t launch_solve_coop(const KArgs& a, int B, hipStream_t st) { return exact::launch_solve_coop(a, B, st); }
int spec_max_instances(int P, int H, int m) { return exact::spec_max_instances(P, H, m); }
hipError_t launch_solve_spec(const KArgs& a, int B, hipStream_t st) { return exact::launch_solve_spec(a, B, st); }
hipError_t launch_relayout(bool to_dev, const float* in, float* out, int B, int P, int G, int C, hipStream_t st) {
    return exact::launch_relayout(to_dev, in, out, B, P, G, C, st);
}
#endif

}  // namespace sdempc
