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
//   * template <int F16>: optional fp16-operand contractions on v_mfma_f32_32x32x16_f16 (SPEC.md §9).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "sdempc_kernels.h"

// This file is compiled twice (Makefile): SDEMPC_FAST=0 -> namespace sdempc::exact, the bit-reproducible arithmetic of SPEC.md
// §3 (the default and the path every parity claim is about); SDEMPC_FAST=1 -> namespace sdempc::fastm, the same kernels with the
// hardware transcendentals v_exp_f32 / v_rcp_f32 / v_rsq_f32 in tanh, sigmoid and the quaternion normalisation (SPEC.md §10,
// `math_mode: fast`): the three instructions are described exhaustively by structure + record (SPEC.md §10a, oracle/transc_model.c), so this namespace too
// is compared with the CPU oracle bit for bit (since round 4).
#ifndef SDEMPC_FAST
#define SDEMPC_FAST 0
#endif
// Each arithmetic mode's build is spread over four translation units so that `make -j` compiles them side by side (one hipcc process per
// unit; the solve kernel has ~90 instantiations of ~25k instructions each): SDEMPC_TU = 0 — every kernel except the duo solve
// kernels, and all launchers; 1 — the duo solve kernels of the two-wave teams (TeamPair, TeamBlock2); 2 — those of the four-wave
// team (TeamBlock); 3 — those of the six-team workgroup (TeamHex). Units 1 to 3 hold nothing but explicit instantiations (list macros below), unit 0 declares them `extern template`.
#ifndef SDEMPC_TU
#define SDEMPC_TU 0
#endif
#ifndef SDEMPC_VAR_PHASE_CLK
#define SDEMPC_VAR_PHASE_CLK 0
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
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
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
// Two waves on one instance: the duo tile layout (64 particles per wave, sdempc_duo.inc.h) of instances with up to four particle groups
// (C2: P = 128). 128-thread workgroups, six per CU at three waves per SIMD.
struct TeamBlock2 {
    static constexpr int NT = 128, NWAVES = 2, IPB = 1, BNT = 128;
    DI static int tid() { return threadIdx.x; }
    DI static int team() { return 0; }
    DI static void sync() { __syncthreads(); }
};
// Two instances per four-wave workgroup, two waves each: the duo tile layout of instances with up to four particle groups WITHOUT the
// price of small workgroups (the weights are staged once per workgroup, so the per-step control table stays in LDS at three workgroups =
// twelve waves per CU). The two teams of a workgroup run independent control flow (different instances), so their barrier cannot be
// s_barrier: a counter in LDS per team, two arrivals per episode (the arriving wave waits for the counter's next even value).
__shared__ unsigned sdempc_pair_bar[8];
// Diagnostic builds only (tools/build_variant.sh clk "-DSDEMPC_VAR_PHASE_CLK=1", tools/phase_clock.py): per-wave time by phase
// (s_memrealtime ticks of 10 ns), summed in LDS over a solve and flushed (>> 10: 10.24 us units, two per word) into KArgs::work in place of the work counters.
// slots: 0 solve, 1 cost rollouts, 2 their step loops, 3 gradient evaluations (2, 3: kept in LDS only), 4 forward sweeps, 5 adjoint sweeps, 6 team barriers
#if SDEMPC_VAR_PHASE_CLK
__shared__ unsigned long long sdempc_clk[8][8];
DI unsigned long long clk_now() { return __builtin_amdgcn_s_memrealtime(); }
DI void clk_add(int slot, unsigned long long t0) { if ((threadIdx.x & 63) == 0) sdempc_clk[threadIdx.x >> 6][slot] += clk_now() - t0; }
#define CLK_BEGIN(v) const unsigned long long v = clk_now()
#define CLK_END(slot, v) clk_add(slot, v)
#else
#define CLK_BEGIN(v) ((void)0)
#define CLK_END(slot, v) ((void)0)
#endif
template <int IPB_>
struct TeamPairT {
    static constexpr int NT = 128, NWAVES = 2, IPB = IPB_, BNT = 128 * IPB_;
    DI static int tid() { return threadIdx.x & 127; }
    DI static int team() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 7); }
    DI static void sync() {
        CLK_BEGIN(tb);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");          // this wave's LDS writes are complete before it signals
        unsigned* c = sdempc_pair_bar + team();
        unsigned v = 0;
        if ((threadIdx.x & 63) == 0) v = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        v = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
        const unsigned target = (v | 1u) + 1u;
        while ((int)(__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - target) < 0) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        CLK_END(6, tb);
    }
};
// Two-wave teams: two per workgroup (three workgroups per CU), or six in ONE twelve-wave workgroup per CU (TeamHex: full-GPU launches; the
// weights are staged once per CU instead of three times, which leaves 30 KB of LDS free where three 52 KB workgroups leave none)
using TeamPair = TeamPairT<2>;
using TeamHex = TeamPairT<6>;
struct TeamWave {
    static constexpr int NT = 64, NWAVES = 1, IPB = 4, BNT = 256;
    DI static int tid() { return threadIdx.x & 63; }
    DI static int team() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }   // wave-uniform -> SGPR
    // LDS operations of one wave execute in order; the fence keeps the compiler from moving them and drains lgkmcnt
    DI static void sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier(); }
};

#include "sdempc_math.inc.h"

// ------------------------------------------------------------------------------------------------
// shared-memory carve (floats). One instance per workgroup.
// ------------------------------------------------------------------------------------------------
struct Smem {
    float *W3, *w3n, *b1n, *b2, *b1d, *W1zT, *W1uT;  // weights, row-major in hidden-unit index
    float *A2, *A2T;                                   // MFMA A operands of W2 / W2^T: [q][lane][4]
    float *A2h;                                        // f16 mode: A operands of W2, [half][lane][8 x fp16]
    float *A2x, *A2xT;                                 // f32x3 mode (aliases the three regions above): bf16 limbs of W2 / W2^T, [limb][half][lane][8 x bf16]
    float *ust;                                        // [H][36]: c[32], Tz, tau[3]
    float *xref;                                       // [H+1][13]
    float *dt, *sdt, *disc;                            // [H], [H][6], [H+1]
    float *red;                                        // [16] block-reduction scratch
    float *tot;                                        // cooperative path only: [H*12] particle sums of the adjoint outputs
    float *rec, *nzl, *cend;                           // cooperative path only (sdempc_lane2.inc.h): step records [H][REC], noise rows of the four waves [4][H][NZL], end of the carve
    float *v[6];                                       // N-vectors: 0 xk, 1 yk, 2 xn, 3 g, 4 d1, 5 ucur
    float *nzs;                                        // duo layout only: this WAVE's noise staging rows [6][64] (LDS-DMA target), behind every team's state
};
constexpr int NZ_STAGE = 6 * 64;                       // floats per wave
// SPEC.md §10e (math_mode fast + f32x3): compact A-operand images of the adjoint's two narrow contractions, in the slack of the 3072-float limb region
// (in this mode the forward and the transposed W2 image take two binary16 limbs each: floats 0..1023 and 1536..2559 of the region)
constexpr int ZROWS_D = 15, ZROWS_N = 7;               // drift image: rows 0..5 zbar, 6..13 W1u^T (m <= 8), 14 zero; density image: rows 0..5, 6 zero
constexpr int A3T_ROWS = 33;                           // (-2 W3)^T for the K = 6 contraction with the output adjoints: [limb][row][8 x binary16], k slots 0..5 of the LOWER lane half; row 32 zero (upper half)
// where the four images sit (floats from sm.A2; xt: the transposed (4 W2) image, two limbs = 1024 floats; azd 480, azn 224, a3t 264 floats):
//   f32x3: forward image 0..1023, xt 1536..2559 (= sm.A2xT), azd 1024.., azn 2560.., a3t 2784..   f16: forward image sm.A2h 2048..2559, xt 0..1023, azd 1024.., azn 1536.., a3t 1760..
struct AdjOff { int xt, azd, azn, a3t; };
__host__ __device__ constexpr AdjOff adj_off(int f16) { return f16 == 2 ? AdjOff{1536, 1024, 2560, 2784} : AdjOff{0, 1024, 1536, 1760}; }
static_assert(adj_off(2).azd + 2 * 2 * 2 * ZROWS_D * 4 <= 1536 && adj_off(2).azn + 2 * 2 * 2 * ZROWS_N * 4 <= adj_off(2).a3t && adj_off(2).a3t + 2 * A3T_ROWS * 4 <= 3072, "f32x3: adjoint images must fit the slack of the limb region");
static_assert(adj_off(1).azd + 2 * 2 * 2 * ZROWS_D * 4 <= adj_off(1).azn && adj_off(1).azn + 2 * 2 * 2 * ZROWS_N * 4 <= adj_off(1).a3t && adj_off(1).a3t + 2 * A3T_ROWS * 4 <= 2048, "f16: adjoint images must stay below the fp16 forward image");
constexpr int UST = 36;
constexpr int REC = 64, NZL = 8;                       // cooperative layouts: floats per step record / per noise row in LDS (sdempc_lane2.inc.h)
constexpr int COOP_ROW = 172;                          // floats per (particle, step) checkpoint row of the cooperative layouts (sdempc_lane2.inc.h)
// per-instance output workspace of the cooperative layouts (KArgs::coop_pp): 2 x SPEC_SLOTS slots of [part_stride(H)][Ppad] per-particle outputs — 64-bit
// tagged words {value, tag} in the speculative kernel (zeroed by the host before each of its launches: coop_pp_tagged_bytes), floats in the first two slots'
// worth of the region in the plain cooperative kernel — then [2][part_stride(H)] 64-bit tagged totals
constexpr int SPEC_SLOTS = 9;
__host__ __device__ inline size_t coop_gtot_offset(int H, int Ppad) { return (size_t)2 * 2 * SPEC_SLOTS * part_stride(H) * Ppad; }
__host__ __device__ inline size_t coop_pp_stride(int H, int Ppad) { return coop_gtot_offset(H, Ppad) + 4 * (size_t)part_stride(H); }

DI Smem carve(float* base, int H, int m, int team, bool coop = false, bool ust_lds = true) {
    Smem s;
    float* p = base;
    // ---- shared by every team of the workgroup ----
    s.W3 = p; p += 6 * HID;
    s.w3n = p; p += HID;
    s.b1n = p; p += HID;
    s.b2 = p; p += HID;
    s.b1d = p; p += HID;
    s.W1zT = p; p += NN * 2 * HID;
    s.W1uT = p; p += m * HID;
    s.A2 = p; p += HID * HID;
    s.A2T = p; p += HID * HID;
    s.A2h = p; p += 512 + 512;                 // (f32x3 mode: the same 3072 floats hold 2 x 3 limbs x 2 halves x 64 lanes x 16 bytes)
    s.A2x = s.A2; s.A2xT = s.A2 + 3 * 2 * 64 * 4;
    s.dt = p; p += (H + 3) & ~3;
    s.sdt = p; p += (H * NN + 3) & ~3;
    s.disc = p; p += (H + 1 + 3) & ~3;
    // ---- per team ----
    const int nv = (H * m + 3) & ~3;
    const int per_team = (ust_lds ? H * UST : 0) + (((H + 1) * NX + 3) & ~3) + 16 + 6 * nv;
    p += team * per_team;
    s.ust = p; if (ust_lds) p += H * UST;      // !ust_lds: the kernel points s.ust at its row of KArgs::ustg instead
    s.xref = p; p += ((H + 1) * NX + 3) & ~3;
    s.red = p; p += 16;
    for (int i = 0; i < 6; ++i) { s.v[i] = p; p += nv; }
    s.tot = p;                       // only the cooperative kernel (one team per workgroup) reserves it: see smem_bytes
    s.rec = s.tot + ((H * 12 + 3) & ~3); s.nzl = s.rec + H * REC; s.cend = s.nzl + 4 * H * NZL;      // (coop only)
    s.nzs = nullptr;                 // set by the duo kernel (smem_floats(...) + wave * NZ_STAGE)
    (void)coop;
    return s;
}
__host__ __device__ inline size_t smem_floats(int H, int m, int ipb, bool coop = false, bool ust_lds = true) {
    size_t shared = 6 * HID + 4 * HID + NN * 2 * HID + (size_t)m * HID + 2 * HID * HID + 1024 + ((H + 3) & ~3) + ((H * NN + 3) & ~3) + ((H + 1 + 3) & ~3);
    size_t per_team = (ust_lds ? (size_t)H * UST : 0) + (((H + 1) * NX + 3) & ~3) + 16 + 6 * (size_t)((H * m + 3) & ~3);
    return shared + ipb * per_team + (coop ? (size_t)((H * 12 + 3) & ~3) + (size_t)H * REC + (size_t)4 * H * NZL : 0);
}
// nz_waves: waves per workgroup that get a noise staging area (duo launches), 0 otherwise
inline size_t smem_bytes(int H, int m, int ipb, bool coop = false, bool ust_lds = true, int nz_waves = 0) {
    return (smem_floats(H, m, ipb, coop, ust_lds) + (size_t)nz_waves * NZ_STAGE) * sizeof(float);
}

// blob float payload offsets (SPEC.md §2)
constexpr int OFF_SF = blob::SF;     // sF[3], sT[3]
constexpr int OFF_W1Z = blob::W1Z, OFF_B1 = blob::B1, OFF_W1U = blob::W1U, OFF_W2 = blob::W2, OFF_B2 = blob::B2,
              OFF_W3 = blob::W3, OFF_B3 = blob::B3, OFF_W3N = blob::W3N, OFF_B3N = blob::B3N;

// math_mode fast (SPEC.md §10b): the blob is two blocks of this layout — [0]: what the forward pass uses (pre-scale and affine map of the hardware
// activation folded in), [VJP_BASE]: what the vector-Jacobian products use (W1z, W1u as given; 4 W2, 4 W3, 4 w3n). One block otherwise.
constexpr int BLOB_FLOATS = OFF_B3N + 8;
static_assert(BLOB_FLOATS == 2120, "blob layout (include/sdempc.h: SDEMPC_BLOB_FLOATS)");
constexpr int VJP_BASE = FAST ? BLOB_FLOATS : 0;
// The LDS images W1zT / W1uT and W3 / w3n serve both directions. They hold the VJP block's layer-1 weights (the forward uses take the pre-scale on the
// fly: one rounding, as sdempc_create's) and the forward block's output weights -2 W (the adjoint uses take the exact factor -2 on the fly).
DI float fwd_w1(float w) { return FAST ? TANH_PRESCALE * w : w; }
DI float vjp_w3(float w) { return FAST ? -2.0f * w : w; }

// MFMA A operands kept in registers for the whole kernel
struct WaveW {
    float w1d[3], w1n[3];   // f32 mode: layer-1 A operands (3 k-steps per tile)
    half8 h1d, h1n;         // f16 mode: layer-1 A operands (k slots 0..5 of lanes 0..31, rest zero)
};

DI int opaque_s(int v) { asm volatile("" : "+s"(v)); return v; }
DI int opaque_v(int v) { asm volatile("" : "+v"(v)); return v; }
DI int rowmap(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// SPEC.md §10e (math_mode fast, mlp_dtype f32x3 / f16): A-operand images of the adjoint's contractions, two binary16 limbs each (round to nearest even: the casts; w - limb exact).
// Compact images [limb][K-half][lane half][row][8 x binary16], last row zero: drift tile rows 0..5 = W1z[un][row], rows 6..6+m-1 = W1u[un][row-6]; density tile rows 0..5 =
// W1z[32+un][row] (the VJP block's: as given); (-2 W3)^T [limb][row][8]: k slots 0..5 of the lower lane half. with_xt: also the full transposed (4 W2) image (f16 mode: the
// f32x3 branch of load_weights writes its own beside the forward image).
DI void load_adj_images(const KArgs& a, const Smem& sm, const float* w, int tid, int BNT, const AdjOff O, bool with_xt) {
    if (with_xt) {
        unsigned short* axt = reinterpret_cast<unsigned short*>(sm.A2 + O.xt);
        for (int i = tid; i < 2 * 64 * 8; i += BNT) {
            const int e = i & 7, l = (i >> 3) & 63, hf = i >> 9, jj = l & 31, hh = l >> 5, un = rowmap(8 * hf + e, hh);
            const float wv = w[VJP_BASE + OFF_W2 + un * HID + jj];
            const _Float16 h1 = (_Float16)wv;
            const _Float16 h2 = (_Float16)(wv - (float)h1);
            axt[(0 * 2 + hf) * 512 + l * 8 + e] = __builtin_bit_cast(unsigned short, h1);
            axt[(1 * 2 + hf) * 512 + l * 8 + e] = __builtin_bit_cast(unsigned short, h2);
        }
    }
    unsigned short* zd = reinterpret_cast<unsigned short*>(sm.A2 + O.azd);
    unsigned short* zn = reinterpret_cast<unsigned short*>(sm.A2 + O.azn);
    for (int i = tid; i < 2 * 2 * (ZROWS_D + ZROWS_N) * 8; i += BNT) {
        const int e = i & 7, rw = (i >> 3) % (ZROWS_D + ZROWS_N), hh = ((i >> 3) / (ZROWS_D + ZROWS_N)) & 1, hf = ((i >> 3) / (ZROWS_D + ZROWS_N)) >> 1;
        const int un = rowmap(8 * hf + e, hh);
        const bool dr = rw < ZROWS_D;
        const int row = dr ? rw : rw - ZROWS_D;
        float wz = 0.0f;
        if (dr) wz = row < NN ? w[VJP_BASE + OFF_W1Z + un * NN + row] : (row < NN + a.m ? w[VJP_BASE + OFF_W1U + un * 8 + (row - NN)] : 0.0f);
        else if (row < NN) wz = w[VJP_BASE + OFF_W1Z + (HID + un) * NN + row];
        const _Float16 h1 = (_Float16)wz;
        const _Float16 h2 = (_Float16)(wz - (float)h1);
        unsigned short* dst = dr ? zd : zn;
        const int R = dr ? ZROWS_D : ZROWS_N;
        dst[(((0 * 2 + hf) * 2 + hh) * R + row) * 8 + e] = __builtin_bit_cast(unsigned short, h1);
        dst[(((1 * 2 + hf) * 2 + hh) * R + row) * 8 + e] = __builtin_bit_cast(unsigned short, h2);
    }
    unsigned short* a3 = reinterpret_cast<unsigned short*>(sm.A2 + O.a3t);
    for (int i = tid; i < A3T_ROWS * 8; i += BNT) {
        const int e = i & 7, row = i >> 3;
        const float w3v = (row < HID && e < 6) ? w[OFF_W3 + e * HID + row] : 0.0f;       // the forward block's -2 W3 (SPEC.md §10b)
        const _Float16 h1 = (_Float16)w3v;
        const _Float16 h2 = (_Float16)(w3v - (float)h1);
        a3[(0 * A3T_ROWS + row) * 8 + e] = __builtin_bit_cast(unsigned short, h1);
        a3[(1 * A3T_ROWS + row) * 8 + e] = __builtin_bit_cast(unsigned short, h2);
    }
}

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
    for (int i = tid; i < NN * 2 * HID; i += BNT) { int k = i / (2 * HID), r = i % (2 * HID); sm.W1zT[i] = w[VJP_BASE + OFF_W1Z + r * NN + k]; }
    for (int i = tid; i < a.m * HID; i += BNT) { int j = i / HID, r = i % HID; sm.W1uT[i] = w[VJP_BASE + OFF_W1U + r * 8 + j]; }
    for (int i = tid; i < a.H; i += BNT) sm.dt[i] = a.dt[i];
    for (int i = tid; i < a.H * NN; i += BNT) sm.sdt[i] = a.sdt[i];
    for (int i = tid; i <= a.H; i += BNT) sm.disc[i] = a.disc[i];
#pragma unroll
    for (int s = 0; s < 3; ++s) { ww.w1d[s] = w[OFF_W1Z + j * NN + 2 * s + h]; ww.w1n[s] = w[OFF_W1Z + (HID + j) * NN + 2 * s + h]; }
    if (a.f16 == 2) {
        // SPEC.md §9b: three bf16 limbs of every W2 entry by truncation, w = w1 + w2 + w3 (+ less than 2^-24 |w|): limb l of k slot e
        // of lane half hh in K-half hf is hidden unit rowmap(8 hf + e, hh) — forward A operand row jj, transpose A operand column jj
        unsigned short* ax = reinterpret_cast<unsigned short*>(sm.A2x);
        unsigned short* axt = reinterpret_cast<unsigned short*>(sm.A2xT);
        for (int i = tid; i < 2 * 64 * 8; i += BNT) {
            const int e = i & 7, l = (i >> 3) & 63, hf = i >> 9, jj = l & 31, hh = l >> 5, un = rowmap(8 * hf + e, hh);
            float wv[2] = {w[OFF_W2 + jj * HID + un], w[VJP_BASE + OFF_W2 + un * HID + jj]};
            if constexpr (FAST) {       // SPEC.md §10c / §10e: either operand image as two binary16 limbs, round to nearest even (the casts), w - limb exact
#pragma unroll
                for (int tr = 0; tr < 2; ++tr) {
                    unsigned short* dst = tr ? axt : ax;
                    const _Float16 h1 = (_Float16)wv[tr];
                    const _Float16 h2 = (_Float16)(wv[tr] - (float)h1);
                    dst[(0 * 2 + hf) * 512 + l * 8 + e] = __builtin_bit_cast(unsigned short, h1);
                    dst[(1 * 2 + hf) * 512 + l * 8 + e] = __builtin_bit_cast(unsigned short, h2);
                }
            }
#pragma unroll
            for (int tr = FAST ? 2 : 0; tr < 2; ++tr) {
                unsigned short* dst = tr ? axt : ax;
                float rem = wv[tr];
#pragma unroll
                for (int lb = 0; lb < 3; ++lb) {
                    const unsigned hi = __float_as_uint(rem) & 0xFFFF0000u;
                    dst[(lb * 2 + hf) * 512 + l * 8 + e] = (unsigned short)(hi >> 16);
                    rem = rem - __uint_as_float(hi);
                }
            }
        }
        if constexpr (FAST) load_adj_images(a, sm, w, tid, BNT, adj_off(2), false);
    } else if (FAST && a.f16 == 1) {
        load_adj_images(a, sm, w, tid, BNT, adj_off(1), true);       // (the f32 images below have no reader in this mode: forward from sm.A2h, adjoint from these)
    } else
    // A operand of k-step r for lane l: W2[j][rowmap(r,h)] (forward) / W2[rowmap(r,h)][j] (transpose)
    for (int i = tid; i < HID * HID; i += BNT) {
        int c = i & 3, l = (i >> 2) & 63, q = i >> 8, jj = l & 31, hh = l >> 5, r = 4 * q + c;
        sm.A2[i] = w[OFF_W2 + jj * HID + rowmap(r, hh)];
        sm.A2T[i] = w[VJP_BASE + OFF_W2 + rowmap(r, hh) * HID + jj];
    }
    if (a.f16 == 1) {   // weights are already fp16-representable (quantised on the host): the casts are exact
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
    // A virtual lane reads the elements e = i, i + 256, ...: with 256 threads (or one wave) those are the elements the same thread wrote
    // in the caller's preceding `for (e = tid; e < N; e += NT)` loop; any other team size reads elements other waves wrote: barrier first.
    if constexpr (Team::NT != 256 && Team::NT != 64) Team::sync();
    if constexpr (Team::NT >= 256) {      // the first four waves are the 256 virtual lanes; further waves only take part in the barriers
        float acc = 0.0f;
        if (tid < 256)
            for (int e = tid; e < N; e += 256) acc = elem(e, acc);
        acc = wave_bfly64(acc);
        Team::sync();
        if ((tid & 63) == 0 && tid < 256) sm.red[tid >> 6] = acc;
        Team::sync();
        return ((sm.red[0] + sm.red[1]) + sm.red[2]) + sm.red[3];
    } else {                              // every wave of the team walks all four virtual waves itself: no LDS exchange, no barrier
        const int ln = tid & 63;
        float w[4];
#pragma unroll
        for (int vw = 0; vw < 4; ++vw) {
            float acc = 0.0f;
            for (int e = vw * 64 + ln; e < N; e += 256) acc = elem(e, acc);
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
        for (int j = 0; j < m; ++j) c = FMA(fwd_w1(sm.W1uT[j * HID + r]), u[t * m + j], c);
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

#include "sdempc_step.inc.h"

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

#include "sdempc_lane.inc.h"

#include "sdempc_lane2.inc.h"

#include "sdempc_coop.inc.h"

#include "sdempc_duo.inc.h"

// ------------------------------------------------------------------------------------------------
// block-level rollout: expected cost of control sequence u (LDS). SPEC.md §5.3/§6/§7
//   store_traj: stream x_t to a.traj; want_mean: particle mean trajectory -> xmean_out (global)
// ------------------------------------------------------------------------------------------------
template <class Team, int F16, bool PK = false>
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
template <class Team, int M, int F16, bool PK = false, bool PREF = true>
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
            step_vjp<M, (F16 == 2 || (FAST && F16 == 1)) ? F16 : 0>(a, sm, ww, t, h, lane, xt, xi, A, lam, ebc, lamn, gq);      // (exact f16 mode: the adjoint is the f32 one; fast: SPEC.md §10e in both matrix-pipe modes)
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
// wave, one instance over several workgroups); 3: duo tile layout (64 particles per wave, throughput launches)
template <class Team, int F16, bool PK, int MODE>
DI float team_rollout(const KArgs& a, const Smem& sm, const WaveW& ww, const LaneW& LW, CoopCtx& CC, const float* u, int b, int tid, bool store_traj, float* xmean_out) {
    if constexpr (MODE >= 3) return duo_rollout<Team, F16, MODE == 3>(a, sm, ww, u, b, tid, store_traj, xmean_out);
    else if constexpr (MODE == 2) return coop_rollout<Team>(a, sm, LW, CC, u, b, tid, xmean_out);
    else if constexpr (MODE == 1) return lane_rollout<Team>(a, sm, LW, u, b, tid, store_traj, xmean_out);
    else return block_rollout<Team, F16, PK>(a, sm, ww, u, b, tid, store_traj, xmean_out);
}
template <class Team, int M, int F16, bool PK, bool PREF, int MODE>
DI float team_cost_grad(const KArgs& a, const Smem& sm, const WaveW& ww, const LaneW& LW, CoopCtx& CC, const float* y, float* gout, int b, int tid) {
    if constexpr (MODE >= 3) return duo_cost_grad<Team, M, F16, MODE == 3>(a, sm, ww, y, gout, b, tid);
    else if constexpr (MODE == 2) return coop_cost_grad<Team, M>(a, sm, LW, CC, y, gout, b, tid);
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
#define SDEMPC_KERNEL_PROLOGUE(USTG_)                                                \
    extern __shared__ __attribute__((aligned(16))) float smem[];                     \
    LaneW LW;                                                                        \
    CoopCtx CC;                                                                      \
    const int tid = Team::tid();                                                     \
    int b_;                                                                          \
    if constexpr (MODE == 2) {                                                       \
        b_ = blockIdx.x / a.coop_nwg;                                                \
        CC.nwg = a.coop_nwg; CC.wgi = blockIdx.x - b_ * a.coop_nwg; CC.Ppad = a.G * 32; CC.epoch = 0u; CC.spin_limit = a.coop_spin; CC.fence = a.opt.coop_fence; \
        CC.bar = a.coop_bar + COOP_BAR_WORDS * b_;                                              \
        CC.pp = a.coop_pp + (size_t)b_ * coop_pp_stride(a.H, CC.Ppad);               \
        CC.gtot = reinterpret_cast<unsigned long long*>(CC.pp + coop_gtot_offset(a.H, CC.Ppad)); \
        CC.ck = a.coop_ck + (size_t)b_ * a.P * (a.H + 1) * COOP_ROW;                 \
    } else {                                                                         \
        b_ = blockIdx.x * Team::IPB + Team::team();                                  \
    }                                                                                \
    const int b = __builtin_amdgcn_readfirstlane(b_);                                \
    Smem sm = carve(smem, a.H, a.m, Team::team(), MODE == 2, !(USTG_));              \
    if constexpr (USTG_) sm.ust = a.ustg + (size_t)b * a.H * UST;                    \
    WaveW ww;                                                                        \
    load_weights(a, sm, ww, threadIdx.x, Team::BNT);                                 \
    __syncthreads();                                                                 \
    if (b >= a.B) return; /* no workgroup-wide barrier below this line in TeamWave */ \
    if constexpr (MODE == 2) { if ((int)blockIdx.x == a.opt.absent_wg) return; } /* fault injection: SDEMPC_OPT_TEST_ABSENT_WG */ \
    if constexpr (MODE == 1) load_lane_weights(a, LW, threadIdx.x & 63);              \
    load_common<Team>(a, sm, b, tid);                                                \
    if constexpr (MODE == 2) {                                                       \
        __syncthreads(); lane2_stage<Team>(a, sm, b, CC.wgi, tid);                   \
        if (CC.wgi == 0)   /* tags of an earlier launch must not be taken for this one's (first use: behind a grid barrier) */ \
            for (int i = tid; i < 2 * part_stride(a.H); i += Team::NT) __hip_atomic_store(CC.gtot + i, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
    }

template <class Team, int F16, int MODE = 0>
__global__ void __launch_bounds__(Team::BNT, (MODE ? 2 : 3)) sdempc_rollout_kernel(KArgs a) {
    SDEMPC_KERNEL_PROLOGUE(false);
    const int N = a.H * a.m;
    for (int e = tid; e < N; e += Team::NT) sm.v[5][e] = a.u[(size_t)b * N + e];
    float c = team_rollout<Team, F16, false, MODE>(a, sm, ww, LW, CC, sm.v[5], b, tid, a.store_traj != 0, a.xmean ? a.xmean + (size_t)b * (a.H + 1) * NX : nullptr);
    if (tid == 0) a.cost[b] = c;
}

template <class Team, int M, int F16, int MODE = 0>
__global__ void __launch_bounds__(Team::BNT, (MODE ? 2 : 3)) sdempc_grad_kernel(KArgs a) {
    SDEMPC_KERNEL_PROLOGUE(false);
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
// USTG: the per-step control table [H][36] lives in global memory (KArgs::ustg, L1/L2-resident) instead of LDS: long horizons keep three
// workgroups per CU (C5: 79 KB -> 50 KB per instance)
// One instance's solve (SPEC.md §8) on its team: everything after the LDS staging of the kernel prologue.
template <class Team, int M, int F16, bool PK, int MODE>
DI void solve_instance(const KArgs& a, const Smem& sm, const WaveW& ww, const LaneW& LW, CoopCtx& CC, const int b, const int tid) {
    const int m = a.m, N = a.H * m;
    float *xk = sm.v[0], *yk = sm.v[1], *xn = sm.v[2], *g = sm.v[3], *d1 = sm.v[4], *d2 = sm.v[5];
    for (int e = tid; e < N; e += Team::NT) {
        int jj = e % m;
        float v = clampf(a.u[(size_t)b * N + e], a.C.ulo[jj], a.C.uhi[jj]);
        xk[e] = v; yk[e] = v;
    }
    // wave-uniform optimiser scalars are pinned to SGPRs (uni_f): a uniform value left in a VGPR can be spilled under the partial EXEC
    // mask of a divergent block and restored under the full one (seen in the speculative kernel; see sdempc_spec.inc.h)
#if SDEMPC_VAR_PHASE_CLK
    if ((threadIdx.x & 63) == 0) for (int i = 0; i < 8; ++i) sdempc_clk[threadIdx.x >> 6][i] = 0ull;
#endif
    CLK_BEGIN(t_solve);
    const float c_init = uni_f(team_rollout<Team, F16, PK, MODE>(a, sm, ww, LW, CC, xk, b, tid, false, nullptr));
    float c_x = c_init, s = a.stepsize_in[b], gsq = 0.0f, sum_ls = 0.0f, sum_s = 0.0f;
    int kr = 0, noimp = 0, nit = 0, nls_tot = 0, plain = 1;
    // (c_y, g, |g|^2) are a pure function of yk: when an iteration leaves yk where it was (a plain gradient step from yk == xk that did not
    // lower the cost — every iteration of a cold start until the line search has shrunk the step far enough) the next iteration would
    // recompute the same bits, so they are kept instead. g lives in LDS and is written by the gradient evaluation only.
    int yk_unchanged = 0, ngrad = 0;
    float c_y = 0.0f;
    for (int k = 0; k < a.A.max_iter; ++k) {
        c_x = uni_f(c_x); s = uni_f(s); sum_ls = uni_f(sum_ls); sum_s = uni_f(sum_s); c_y = uni_f(c_y); gsq = uni_f(gsq);
        if (!yk_unchanged) {
            c_y = uni_f(team_cost_grad<Team, M, F16, PK, solve_waves_per_simd<Team, PK>() == 2, MODE>(a, sm, ww, LW, CC, yk, g, b, tid));
            gsq = uni_f(block_dot<Team>(sm, g, g, N, tid));
            ngrad += 1;
        }
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
                c_n = uni_f(team_rollout<Team, F16, PK, MODE>(a, sm, ww, LW, CC, xn, b, tid, false, nullptr));
                float gd = uni_f(block_dot<Team>(sm, g, d1, N, tid));
                nls = jl + 1;
                if (c_n <= FMA(a.A.coef, gd, c_y)) break;
                if (jl < a.A.maxls - 1) s = s * a.A.dec;
            }
        } else {
            s = a.A.stepsize;
            Team::sync();
            for (int e = tid; e < N; e += Team::NT) { int jj = e % m; xn[e] = clampf(FMA(-s, g[e], yk[e]), a.C.ulo[jj], a.C.uhi[jj]); }
            c_n = uni_f(team_rollout<Team, F16, PK, MODE>(a, sm, ww, LW, CC, xn, b, tid, false, nullptr));
            nls = 1;
        }
        sum_ls = sum_ls + (float)nls; sum_s = sum_s + s; nit = k + 1; nls_tot += nls;
        int stop = (__builtin_fabsf(c_n - c_x) <= FMA(a.A.rtol, __builtin_fabsf(c_x), a.A.atol));
        Team::sync();
        yk_unchanged = plain && !(c_n < c_x);       // yk == xk already and it stays there
        if (c_n < c_x) {
            for (int e = tid; e < N; e += Team::NT) { d1[e] = yk[e] - xn[e]; d2[e] = xn[e] - xk[e]; }
            float rs = uni_f(block_dot<Team>(sm, d1, d2, N, tid));
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
#if SDEMPC_VAR_PHASE_CLK
    CLK_END(0, t_solve);
    if (a.work && (threadIdx.x & 63) == 0) {
        const unsigned long long* c = sdempc_clk[threadIdx.x >> 6];
        __hip_atomic_fetch_add(a.work + 0, ((c[0] >> 10) << 32) | (c[1] >> 10), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) __hip_atomic_fetch_add(a.work + 1, (unsigned long long)ngrad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (as in the product build)
        __hip_atomic_fetch_add(a.work + 2, ((c[4] >> 10) << 32) | (c[5] >> 10), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(a.work + 3, ((c[6] >> 10) << 32) | 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#endif
    if (tid == 0 && (MODE != 2 || CC.wgi == 0)) {
        float* inf = a.info + (size_t)b * 8;
        const float fn = (float)nit;
        inf[0] = nit ? sum_ls / fn : 0.0f; inf[1] = s; inf[2] = fn; inf[3] = gsq; inf[4] = nit ? sum_s / fn : 0.0f;
        inf[5] = c_init; inf[6] = c_x; inf[7] = (float)nls_tot;
        if (a.work && !SDEMPC_VAR_PHASE_CLK) {   // work actually done (roofline accounting, sdempc_work_counters): solves, gradient evaluations, forward-only rollouts
            __hip_atomic_fetch_add(a.work + 0, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(a.work + 1, (unsigned long long)ngrad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(a.work + 2, (unsigned long long)(nls_tot + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#if SDEMPC_VAR_PHASE_CLK    // diagnostic build: where and when this instance was solved, in place of three telemetry words
        inf[0] = __uint_as_float(__builtin_amdgcn_s_getreg(63492));                       // HW_REG_HW_ID
        inf[1] = __uint_as_float(__builtin_amdgcn_s_getreg(63508));                       // HW_REG_XCC_ID
        inf[3] = (float)(unsigned)(t_solve & 0x3FFFFFFFull) * 1e-5f;                      // start, ms (wraps at 10.7 s)
        inf[4] = (float)(clk_now() - t_solve) * 1e-5f;                                    // duration, ms
#endif
        if constexpr (MODE == 2) {   // a grid barrier gave up (never seen in testing; bounded so that a fault cannot hang the GPU): poison the telemetry
            if (__hip_atomic_load(CC.bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
                for (int i = 0; i < 8; ++i) inf[i] = __builtin_nanf("");
        }
    }
}

template <class Team, int M, int F16, bool PK = false, int MODE = 0, bool USTG = false>
__global__ void __launch_bounds__(Team::BNT, (MODE == 2 && PK ? 1 : (MODE == 1 || MODE == 2) ? 2 : solve_waves_per_simd<Team, PK>())) sdempc_solve_kernel(KArgs a) {
    if constexpr (MODE >= 3) {
        // Duo throughput launches (MODE 3: noise through LDS staging rows; 4: through registers) are PERSISTENT: the grid holds as many workgroups as are resident at once (launch_duo_m) and every
        // workgroup walks the instances b = blockIdx.x, + gridDim.x, ... With only two rounds of six small workgroups per CU the
        // hardware dispatcher's placement of a plain one-workgroup-per-instance grid leaves CUs idle for a sixth of the launch
        // (tools/occ_probe.hip; measured with SQ_BUSY_CYCLES / SQ_CYCLES); no workgroup depends on another, so nothing needs co-residency.
        extern __shared__ __attribute__((aligned(16))) float smem[];
        LaneW LW;
        CoopCtx CC;
        const int tid = Team::tid();
        Smem sm = carve(smem, a.H, a.m, Team::team(), false, !USTG);
        if constexpr (MODE == 3) sm.nzs = smem + smem_floats(a.H, a.m, Team::IPB, false, !USTG) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * NZ_STAGE;
        WaveW ww;
        if constexpr (Team::IPB > 1) { if (threadIdx.x < Team::IPB) sdempc_pair_bar[threadIdx.x] = 0u; }
        load_weights(a, sm, ww, threadIdx.x, Team::BNT);
        __syncthreads();                           // weights staged (whole workgroup); from here on every team runs on its own
        // From three instances per team on, instances are handed out in order of completion (a ticket word behind KArgs::work, set by
        // the launcher to the first instance beyond the grid's initial ones) instead of striped over the teams. The three waves that
        // share a SIMD do not advance at the same rate: the issue arbiter favours the older wave slot, and a diagnostic build
        // (tools/phase_clock.py: start, duration and hardware wave slot of every instance) shows solves of identical work taking 382 /
        // 438 / 765 ms in wave slots 0 / 1 / 2. A striped launch is over when the teams in the slowest slots have finished their
        // share; with tickets the fast slots solve twice as many instances as the slow ones: C2 2,925 -> 3,120-3,200 solves/s from
        // three rounds on (C3 1,486 -> 1,576). With two instances per team the coarse granularity at the end of the launch costs more
        // than the balance gains (2,743), so short launches stay striped. Which team solves an instance does not change its bits.
        // The ticket word is never reset (a reset by hipMemsetD32Async ahead of the launch was seen to take effect late on some boxes:
        // the first 1,536 instances were then solved twice — same bits, wasted work — and a stale HIGH value would have left instances
        // unsolved): a draw is relative to the word's value at launch, which the launcher knows because every launch advances it by B.
        unsigned* ticket = a.tickets ? reinterpret_cast<unsigned*>(a.work + 4) : nullptr;
        const unsigned nteams = gridDim.x * Team::IPB, ndraw = (unsigned)a.B - nteams;     // successful draws of this launch
        int bb = blockIdx.x * Team::IPB + Team::team();
        while (bb < a.B) {
            const int b = __builtin_amdgcn_readfirstlane(bb);
            if constexpr (USTG) sm.ust = a.ustg + (size_t)duo_workspace_slot<Team>() * a.H * UST;      // per slot, like the other workspaces
            Team::sync();                          // the team's previous instance has finished reading its LDS state
            load_common<Team>(a, sm, b, tid);
            solve_instance<Team, M, F16, PK, MODE>(a, sm, ww, LW, CC, b, tid);
            if (ticket) {
                Team::sync();                      // every wave of the team is done with the reduction scratch
                if (tid == 0) reinterpret_cast<unsigned*>(sm.red)[0] = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.ticket_base;
                Team::sync();
                const unsigned t = (unsigned)__builtin_amdgcn_readfirstlane((int)reinterpret_cast<const unsigned*>(sm.red)[0]);
                bb = t < ndraw ? (int)(nteams + t) : a.B;      // (anything else — this team's one failing draw — ends the team)
            } else {
                bb += gridDim.x * Team::IPB;
            }
        }
        return;
    }
    SDEMPC_KERNEL_PROLOGUE(USTG);
    solve_instance<Team, M, F16, PK, MODE>(a, sm, ww, LW, CC, b, tid);
}

// duo solve kernels (MODE 3: noise staging rows, control table in LDS or global memory; MODE 4: neither in LDS), by team shape
#define SDEMPC_DUO_M(X, TEAM, M)                                                                                   \
    X(TEAM, M, 0, 3, false) X(TEAM, M, 1, 3, false) X(TEAM, M, 2, 3, false) X(TEAM, M, 0, 3, true) X(TEAM, M, 1, 3, true) X(TEAM, M, 2, 3, true)  \
    X(TEAM, M, 0, 4, true) X(TEAM, M, 1, 4, true) X(TEAM, M, 2, 4, true)
#if SDEMPC_ALL_VARIANTS
#define SDEMPC_DUO_M8(X, TEAM) SDEMPC_DUO_M(X, TEAM, 8)
#define SDEMPC_DUO_T8(X, TEAM) X(TEAM, 8, 0, 3, false) X(TEAM, 8, 1, 3, false) X(TEAM, 8, 2, 3, false)
#else
#define SDEMPC_DUO_M8(X, TEAM)
#define SDEMPC_DUO_T8(X, TEAM)
#endif
#define SDEMPC_DUO_TEAM(X, TEAM) SDEMPC_DUO_M(X, TEAM, 4) SDEMPC_DUO_M(X, TEAM, 6) SDEMPC_DUO_M8(X, TEAM)
#define SDEMPC_DUO_PAIR(X)                                                                                          \
    X(TeamPair, 4, 0, 3, false) X(TeamPair, 4, 1, 3, false) X(TeamPair, 4, 2, 3, false) X(TeamPair, 6, 0, 3, false) X(TeamPair, 6, 1, 3, false) X(TeamPair, 6, 2, 3, false) \
    SDEMPC_DUO_T8(X, TeamPair)
#define SDEMPC_DUO_HEX(X)                                                                                           \
    X(TeamHex, 4, 0, 3, false) X(TeamHex, 4, 1, 3, false) X(TeamHex, 4, 2, 3, false) X(TeamHex, 6, 0, 3, false) X(TeamHex, 6, 1, 3, false) X(TeamHex, 6, 2, 3, false) \
    SDEMPC_DUO_T8(X, TeamHex)
#define SDEMPC_DUO_DECL(TEAM, M, F16, MODE, USTG) extern template __global__ void sdempc_solve_kernel<TEAM, M, F16, false, MODE, USTG>(KArgs);
#define SDEMPC_DUO_DEF(TEAM, M, F16, MODE, USTG) template __global__ void sdempc_solve_kernel<TEAM, M, F16, false, MODE, USTG>(KArgs);

#if SDEMPC_TU == 1
SDEMPC_DUO_PAIR(SDEMPC_DUO_DEF)
SDEMPC_DUO_TEAM(SDEMPC_DUO_DEF, TeamBlock2)
}  // namespace exact / fastm
#elif SDEMPC_TU == 2
SDEMPC_DUO_TEAM(SDEMPC_DUO_DEF, TeamBlock)
}  // namespace exact / fastm
#elif SDEMPC_TU == 3
SDEMPC_DUO_HEX(SDEMPC_DUO_DEF)
}  // namespace exact / fastm
#else
SDEMPC_DUO_PAIR(SDEMPC_DUO_DECL)
SDEMPC_DUO_HEX(SDEMPC_DUO_DECL)
SDEMPC_DUO_TEAM(SDEMPC_DUO_DECL, TeamBlock2)
SDEMPC_DUO_TEAM(SDEMPC_DUO_DECL, TeamBlock)

#include "sdempc_spec.inc.h"

#if SDEMPC_DEV_KERNEL
// Development aid (tools/dev_isa.sh): only ONE latency kernel is instantiated, device side only, so that a change to the lane-layout step can
// be compiled to ISA and counted in seconds instead of the four minutes of the whole translation unit. Never part of a build.
#if SDEMPC_DEV_KERNEL == 1
template __global__ void sdempc_solve_spec_kernel<4, false>(KArgs);
#elif SDEMPC_DEV_KERNEL == 2
template __global__ void sdempc_solve_spec_kernel<4, true>(KArgs);
#elif SDEMPC_DEV_KERNEL == 3
template __global__ void sdempc_solve_kernel<TeamBlock, 4, false, true, 2>(KArgs);
#elif SDEMPC_DEV_KERNEL == 4
template __global__ void sdempc_solve_kernel<TeamWave, 4, false, false, 1>(KArgs);
#endif
}  // namespace exact
#else
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
static hipError_t launch_k(Kern k, const KArgs& a, hipStream_t st, int ipb, int bnt = 256, bool ust_lds = true) {
    const size_t sb = smem_bytes(a.H, a.m, ipb, false, ust_lds);
    hipError_t e = set_smem_attr((const void*)k, sb);
    if (e != hipSuccess) return e;
    note_kernel((const void*)k);
    hipLaunchKernelGGL(k, dim3((a.B + ipb - 1) / ipb), dim3(bnt), sb, st, a);
    return hipGetLastError();
}
template <class Team>
static hipError_t launch_rollout_team(const KArgs& a, hipStream_t st) {
    if (a.f16 == 2) return launch_k(sdempc_rollout_kernel<Team, 2>, a, st, Team::IPB);
    return a.f16 ? launch_k(sdempc_rollout_kernel<Team, 1>, a, st, Team::IPB) : launch_k(sdempc_rollout_kernel<Team, 0>, a, st, Team::IPB);
}
template <class Team, int F16>
static hipError_t launch_grad_team(const KArgs& a, hipStream_t st) {
    if (a.m == 4) return launch_k(sdempc_grad_kernel<Team, 4, F16>, a, st, Team::IPB);
    if (a.m == 6) return launch_k(sdempc_grad_kernel<Team, 6, F16>, a, st, Team::IPB);
    return launch_k(sdempc_grad_kernel<Team, 8, F16>, a, st, Team::IPB);
}
// LaunchOpts::cus = compute units of the handle's device (set by the C ABI when the device is bound): a grid of at most that many
// workgroups leaves one wave per SIMD. Every dispatch decision below comes from the handle's LaunchOpts (sdempc_set_option); no
// environment variable is read on the launch path.
// Throughput launches of the workgroup-wide team: is the per-step control table better kept in global memory? (SDEMPC_OPT_USTG forces)
bool use_global_ust(int H, int m, const LaunchOpts& o, int nwaves = 4) {
    if (o.ustg >= 0) return o.ustg == 1;
    // workgroups per CU: by registers (launch bounds: three waves per SIMD = twelve per CU) at most 12 / nwaves, otherwise what the
    // 160 KB of LDS hold
    const size_t cap = 156 * 1024, by_regs = 12 / (size_t)nwaves;      // usable LDS per CU: tools/occ_probe.hip
    auto per_cu = [&](size_t bytes) { const size_t n = bytes ? cap / bytes : by_regs; return n > by_regs ? by_regs : n; };
    return per_cu(smem_bytes(H, m, 1, false, false)) > per_cu(smem_bytes(H, m, 1));
}
// Duo tile layout (sdempc_duo.inc.h) for throughput launches of multi-group instances: two waves per instance up to four groups, four
// waves beyond (SDEMPC_OPT_DUO = 0 keeps the one-group-per-wave layout: A/B, tests)
// persistent grid: as many workgroups as the device holds at once (registers: twelve waves per CU; LDS: 156 KB usable per CU, measured
// with tools/occ_probe.hip — three 52 KB workgroups fit, three 53 KB ones do not), each walking its share of the instances
template <class Kern>
static hipError_t launch_persistent(Kern k, const KArgs& a, hipStream_t st, int wg_waves, int bnt, bool ust_lds, int ipb = 1, bool stage = true) {
    const size_t sb = smem_bytes(a.H, a.m, ipb, false, ust_lds, stage ? wg_waves : 0);      // + one noise staging area per wave
    hipError_t e = set_smem_attr((const void*)k, sb);
    if (e != hipSuccess) return e;
    size_t per_cu = 12 / (size_t)wg_waves;
    const size_t by_lds = (156 * 1024) / (sb ? sb : 1);
    if (by_lds < per_cu) per_cu = by_lds;
    if (per_cu < 1) per_cu = 1;
    size_t grid = per_cu * (size_t)(a.opt.cus > 0 ? a.opt.cus : 256);
    const size_t need = ((size_t)a.B + ipb - 1) / ipb;
    if (grid > need) grid = need;
    if (a.ws_rows > 0 && grid * (size_t)ipb > (size_t)a.ws_rows) return hipErrorInvalidValue;      // team slots beyond the workspace rows: never launch (sdempc_api.cpp sizes them from solve_workspace_rows)
    KArgs ka = a;
    ka.tickets = a.work != nullptr && a.ticket_host != nullptr && (size_t)a.B >= 3 * grid * (size_t)ipb;
#if SDEMPC_VAR_STATIC       // diagnostic builds: striped assignment at every batch size (tools/phase_clock.py)
    ka.tickets = 0;
#endif
    if (ka.tickets) ka.ticket_base = *a.ticket_host;       // the teams' initial instances are 0 .. grid * ipb - 1; the draws hand out the rest
    note_kernel((const void*)k);
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(bnt), sb, st, ka);
    e = hipGetLastError();
    if (e == hipSuccess && ka.tickets) *a.ticket_host = ka.ticket_base + (unsigned)a.B;
    return e;
}
// Which of the three builds of a duo team shape: noise staging rows + control table in LDS, staging rows + table in global memory, or
// neither in LDS (long horizons) — the first in that order of preference that keeps the most workgroups per CU.
template <class TeamD, int M, int F16>
static hipError_t launch_duo_m(const KArgs& a, hipStream_t st) {
    constexpr int W = TeamD::NWAVES;
    const size_t cap = 156 * 1024, by_regs = 12 / (size_t)W;
    auto per_cu = [&](bool ust_lds, bool stage) { const size_t n = cap / smem_bytes(a.H, a.m, 1, false, ust_lds, stage ? W : 0); return n > by_regs ? by_regs : n; };
    const bool can_g = a.ustg != nullptr && a.opt.ustg != 0, must_g = can_g && a.opt.ustg == 1;
    int pick = 0;                                            // 0: stage + LDS table, 1: stage + global table, 2: global table only
    size_t best = must_g ? 0 : per_cu(true, true);
    if (must_g) pick = 1, best = per_cu(false, true);
    else if (can_g && per_cu(false, true) > best) pick = 1, best = per_cu(false, true);
    if (can_g && per_cu(false, false) > best) pick = 2;
    if (pick == 2) return launch_persistent(sdempc_solve_kernel<TeamD, M, F16, false, 4, true>, a, st, W, TeamD::BNT, false, 1, false);
    if (pick == 1) return launch_persistent(sdempc_solve_kernel<TeamD, M, F16, false, 3, true>, a, st, W, TeamD::BNT, false);
    return launch_persistent(sdempc_solve_kernel<TeamD, M, F16, false, 3, false>, a, st, W, TeamD::BNT, true);
}
// up to four groups: two waves per instance, two instances per four-wave workgroup (TeamPair) when both fit with the control table in
// LDS at three workgroups per CU; otherwise 128-thread workgroups (TeamBlock2)
template <int M, int F16>
static hipError_t launch_duo_small(const KArgs& a, hipStream_t st) {
    // a launch that fills every team slot of the device anyway: one twelve-wave workgroup of six teams per CU
    const int cus = a.opt.cus > 0 ? a.opt.cus : 256;
    if (a.opt.ustg != 1 && a.opt.hex != 0 && a.B >= 6 * cus && smem_bytes(a.H, a.m, TeamHex::IPB, false, true, 2 * TeamHex::IPB) <= 156 * 1024)
        return launch_persistent(sdempc_solve_kernel<TeamHex, M, F16, false, 3, false>, a, st, 2 * TeamHex::IPB, TeamHex::BNT, true, TeamHex::IPB);
    if (a.opt.ustg != 1 && smem_bytes(a.H, a.m, 2, false, true, 4) * 3 <= 156 * 1024)
        return launch_persistent(sdempc_solve_kernel<TeamPair, M, F16, false, 3, false>, a, st, 4, TeamPair::BNT, true, 2);
    return launch_duo_m<TeamBlock2, M, F16>(a, st);
}
// motor counts whose instantiations exist in every layout: the reference's two vehicles; the generic 8-slot instantiation of the duo / six-team /
// cooperative / speculative layouts only in an all-variants build (other motor counts otherwise run one group per wave: same bits)
static bool every_layout_built(int m) { return m == 4 || m == 6 || SDEMPC_ALL_VARIANTS != 0; }
template <int F16>
static hipError_t launch_duo(const KArgs& a, hipStream_t st) {
    if (a.G <= 4) {
        if (a.m == 4) return launch_duo_small<4, F16>(a, st);
        if (a.m == 6) return launch_duo_small<6, F16>(a, st);
#if SDEMPC_ALL_VARIANTS
        return launch_duo_small<8, F16>(a, st);
#else
        return hipErrorInvalidValue;      // (not reached: launch_solve_team asks every_layout_built)
#endif
    }
    if (a.m == 4) return launch_duo_m<TeamBlock, 4, F16>(a, st);
    if (a.m == 6) return launch_duo_m<TeamBlock, 6, F16>(a, st);
#if SDEMPC_ALL_VARIANTS
    return launch_duo_m<TeamBlock, 8, F16>(a, st);
#else
    return hipErrorInvalidValue;
#endif
}
// duo = auto, up to four groups: does a batch of B fit resident with one group per wave (launch_solve_team, solve_workspace_rows)?
static bool one_group_per_wave_batch(const KArgs& a, int B) {
    if (a.opt.duo >= 0 || a.G > 4) return false;
    if (!every_layout_built(a.m)) return false;
    const bool gtab = use_global_ust(a.H, a.m, a.opt) && a.ustg;
    size_t per_cu = (156 * 1024) / smem_bytes(a.H, a.m, 1, false, !gtab);
    if (per_cu > 3) per_cu = 3;
    return (size_t)B <= per_cu * (size_t)(a.opt.cus > 0 ? a.opt.cus : 256);
}
template <class Team, int F16>
static hipError_t launch_solve_team(const KArgs& a, hipStream_t st) {
#if SDEMPC_ALL_VARIANTS
    if constexpr (F16 == 0 && !FAST) {
        // small-batch (latency) launches: one workgroup per CU at most -> a lone wave per SIMD is issue-bound -> packed tanh
        const int wgs = (a.B + Team::IPB - 1) / Team::IPB;
        const bool pk = a.opt.pk >= 0 ? a.opt.pk == 1 : wgs <= a.opt.cus;     // SDEMPC_OPT_PK forces either instantiation (A/B, tests)
        if (pk) {
            if constexpr (Team::IPB == 1) {
                if (a.G > 4) {   // more particle groups than the four waves of a workgroup: eight waves halve the sequential depth
                    if (a.m == 4) return launch_k(sdempc_solve_kernel<TeamBlock8, 4, 0, true>, a, st, 1, TeamBlock8::BNT);
                    if (a.m == 6) return launch_k(sdempc_solve_kernel<TeamBlock8, 6, 0, true>, a, st, 1, TeamBlock8::BNT);
                    return launch_k(sdempc_solve_kernel<TeamBlock8, 8, 0, true>, a, st, 1, TeamBlock8::BNT);
                }
            }
            if (a.m == 4) return launch_k(sdempc_solve_kernel<Team, 4, 0, true>, a, st, Team::IPB);
            if (a.m == 6) return launch_k(sdempc_solve_kernel<Team, 6, 0, true>, a, st, Team::IPB);
            return launch_k(sdempc_solve_kernel<Team, 8, 0, true>, a, st, Team::IPB);
        }
    }
#endif
    if constexpr (Team::IPB == 1) {
        // auto: every multi-group instance (measured, same box: C2 +1.3 %, C3 +4.4 %, C5 +7.5 % over one group per wave)
        // Batches that one 32-particle group per wave holds resident at once (three four-wave workgroups per CU: B <= 3 x CUs at C2) run
        // that way: four waves per instance instead of the duo layout's two, i.e. the shorter chain per instance and twice the waves per
        // CU (same box, C2 f32x3: B = 1..256 149 against 245 ms per launch, 512 189 / 260, 768 273 / 306; from 1,024 on the duo layout's
        // 1,536 resident instances win: 383 / 343)
        if (a.G >= 2 && a.opt.duo != 0 && every_layout_built(a.m) && !one_group_per_wave_batch(a, a.B)) return launch_duo<F16>(a, st);
        // long horizons: with the control table in LDS only two workgroups fit a CU; without it three do (the kernel is built for three)
        if (use_global_ust(a.H, a.m, a.opt) && a.ustg) {
            if (a.m == 4) return launch_k(sdempc_solve_kernel<Team, 4, F16, false, 0, true>, a, st, 1, Team::BNT, false);
            if (a.m == 6) return launch_k(sdempc_solve_kernel<Team, 6, F16, false, 0, true>, a, st, 1, Team::BNT, false);
#if SDEMPC_ALL_VARIANTS
            return launch_k(sdempc_solve_kernel<Team, 8, F16, false, 0, true>, a, st, 1, Team::BNT, false);
#endif
        }
    }
    if (a.m == 4) return launch_k(sdempc_solve_kernel<Team, 4, F16>, a, st, Team::IPB);
    if (a.m == 6) return launch_k(sdempc_solve_kernel<Team, 6, F16>, a, st, Team::IPB);
    return launch_k(sdempc_solve_kernel<Team, 8, F16>, a, st, Team::IPB);
}
// Single-particle lane layout: f32 contractions only, one wave per instance (SDEMPC_OPT_LANE = 0 forces the tile layout: A/B, tests)
static bool use_lane(const KArgs& k) {
    if (k.f16 || k.P != 1 || k.C.sc_n != 0 || !use_wave_team(k.G, k.H, k.m)) return false;   // (state bounds: tile layouts only)
    return k.opt.lane != 0;
}
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
// ---- cooperative latency path (f32 contractions; both math modes) ----
int coop_nwg(int P) { return (P + 3) / 4; }
int coop_max_instances(int P, int H, int m, const LaunchOpts& o) {
    if (!o.coop || P < 2 || o.cus < 16 || !every_layout_built(m)) return 0;               // SDEMPC_OPT_COOP = 0 disables the path (A/B, tests)
    if (smem_bytes(H, m, 1, true) > 160 * 1024) return 0;
    // every workgroup of the grid must be resident at once: the kernel is built for two waves per SIMD, i.e. two workgroups per CU
    // (its LDS footprint allows more); a margin of 16 workgroups is left
    return (2 * o.cus - 16) / coop_nwg(P);
}
// per-instance workspace, sized for the speculative variant (7 output slots, 3 checkpoint regions); the plain cooperative kernel
// uses a prefix of it
size_t coop_pp_floats(int H, int G) { return coop_pp_stride(H, G * 32); }
size_t coop_ck_floats(int H, int P) { return (size_t)SPEC_CKS * P * (H + 1) * COOP_ROW; }
int spec_max_instances(int P, int H, int m, const LaunchOpts& o) {
    if (!o.spec || !o.coop || !every_layout_built(m)) return 0;      // SDEMPC_OPT_SPEC / SDEMPC_OPT_COOP = 0; P == 1 is welcome here (one wave per workgroup is active)
    const size_t nv = (size_t)((H * m + 3) & ~3);
    if (smem_bytes(H, m, 1, true) + (SPEC_XV * nv + SPEC_MRED) * sizeof(float) > 160 * 1024) return 0;
    return o.cus / (2 * coop_nwg(P));                    // built for one workgroup per CU; at least two groups per instance
}
// KArgs::coop_spin (how long one grid barrier may wait before it gives up and raises the instance's error flag) is set by the C ABI
// from the handle's spin budget: sdempc_api.cpp, SDEMPC_OPT_COOP_SPIN_US.
// All workgroups of a cooperative-layout grid must be resident at once. Default: a plain launch of a grid sized to fit
// (coop_max_instances / spec_max_instances) with every barrier bounded (sdempc_coop.inc.h). SDEMPC_OPT_COOP_LAUNCH = 1 goes through
// hipLaunchCooperativeKernel instead: the runtime validates the grid against the kernel's occupancy and schedules it as a unit —
// same latency and same results on MI355X (all parity tests), but a process that used it crashes at exit under rocprofv3
// (ROCm 7.2: SIGSEGV in the exit handlers after the profile is written), so it is opt-in.
template <class K>
static hipError_t launch_resident(K kern, dim3 grid, dim3 block, size_t sb, hipStream_t st, const KArgs& k) {
    note_kernel((const void*)kern);
    if (k.opt.coop_launch == 1) {
        KArgs kk = k;
        void* args[] = {(void*)&kk};
        const hipError_t rc = hipLaunchCooperativeKernel((const void*)kern, grid, block, args, (unsigned)sb, st);
        if (rc == hipSuccess) return rc;
        (void)hipGetLastError();
        if (rc != hipErrorCooperativeLaunchTooLarge && rc != hipErrorNotSupported && rc != hipErrorInvalidConfiguration) return rc;
    }
    hipLaunchKernelGGL(kern, grid, block, sb, st, k);
    return hipGetLastError();
}
template <int M>
static hipError_t launch_spec_m(const KArgs& k, hipStream_t st) {
    auto kern = k.P == 1 ? sdempc_solve_spec_kernel<M, true> : sdempc_solve_spec_kernel<M, false>;
    const size_t sb = smem_bytes(k.H, k.m, 1, true) + (SPEC_XV * (size_t)((k.H * k.m + 3) & ~3) + SPEC_MRED) * sizeof(float);
    hipError_t e = set_smem_attr((const void*)kern, sb);
    if (e != hipSuccess) return e;
    return launch_resident(kern, dim3(k.B * k.coop_ngrp * k.coop_nwg), dim3(TeamBlock::BNT), sb, st, k);
}
hipError_t launch_solve_spec(const KArgs& a, int B, hipStream_t st) {
    KArgs k = a; k.B = B; k.coop_nwg = coop_nwg(k.P);
    if (B < 1 || B > (spec_max_instances)(k.P, k.H, k.m, k.opt) || !k.coop_bar || !k.coop_pp || !k.coop_ck) return hipErrorInvalidValue;
    k.coop_ngrp = k.opt.cus / (B * k.coop_nwg);
    if (k.coop_ngrp > SPEC_GROUPS) k.coop_ngrp = SPEC_GROUPS;
    if (k.m == 4) return launch_spec_m<4>(k, st);
    if (k.m == 6) return launch_spec_m<6>(k, st);
#if SDEMPC_ALL_VARIANTS
    return launch_spec_m<8>(k, st);
#else
    return hipErrorInvalidValue;
#endif
}
template <int M>
static hipError_t launch_coop_m(const KArgs& k, hipStream_t st) {
    const size_t sb = smem_bytes(k.H, k.m, 1, true);
    if (k.B * k.coop_nwg <= k.opt.cus) {      // one workgroup per CU: the 512-register build (no scratch spills in the sweeps)
        auto kern = sdempc_solve_kernel<TeamBlock, M, false, true, 2>;
        hipError_t e = set_smem_attr((const void*)kern, sb);
        if (e != hipSuccess) return e;
        return launch_resident(kern, dim3(k.B * k.coop_nwg), dim3(TeamBlock::BNT), sb, st, k);
    }
    auto kern = sdempc_solve_kernel<TeamBlock, M, false, false, 2>;
    hipError_t e = set_smem_attr((const void*)kern, sb);
    if (e != hipSuccess) return e;
    return launch_resident(kern, dim3(k.B * k.coop_nwg), dim3(TeamBlock::BNT), sb, st, k);
}
hipError_t launch_solve_coop(const KArgs& a, int B, hipStream_t st) {
    KArgs k = a; k.B = B; k.coop_nwg = coop_nwg(k.P);
    if (B < 1 || B > (coop_max_instances)(k.P, k.H, k.m, k.opt) || !k.coop_bar || !k.coop_pp || !k.coop_ck) return hipErrorInvalidValue;
    if (k.m == 4) return launch_coop_m<4>(k, st);
    if (k.m == 6) return launch_coop_m<6>(k, st);
#if SDEMPC_ALL_VARIANTS
    return launch_coop_m<8>(k, st);
#else
    return hipErrorInvalidValue;
#endif
}

// Rows of the per-instance / per-slot workspaces (KArgs::traj, act, part, ustg) a solve launch of B instances touches: B for the
// layouts that run one workgroup (or wave) per instance, the number of team slots for the persistent duo launches (at most six teams
// per CU: 128-thread workgroups six per CU, or three four-wave workgroups of two teams). Mirrors the dispatch of launch_solve_team.
int solve_workspace_rows(const KArgs& k, int B) {
    if (use_lane(k) || use_wave_team(k.G, k.H, k.m)) return B;
#if SDEMPC_ALL_VARIANTS
    if (k.f16 == 0 && !FAST) {
        const bool pk = k.opt.pk >= 0 ? k.opt.pk == 1 : B <= k.opt.cus;
        if (pk) return B;
    }
#endif
    if (one_group_per_wave_batch(k, B)) return B;      // (launch_solve_team: batches that fit resident with one group per wave)
    if (k.G >= 2 && k.opt.duo != 0 && every_layout_built(k.m)) {       // (team slots come in workgroups of up to two teams: an odd batch leaves the last slot idle but counted; six-team workgroups only run full)
        const int slots = 6 * (k.opt.cus > 0 ? k.opt.cus : 256), even = (B + 1) & ~1;
        return even < slots ? even : slots;
    }
    return B;
}
hipError_t launch_rollout(const KArgs& a, int B, hipStream_t st) {
    KArgs k = a; k.B = B;
    if (use_lane(k)) return launch_lane_m(0, k, st);
    return use_wave_team(k.G, k.H, k.m) ? launch_rollout_team<TeamWave>(k, st) : launch_rollout_team<TeamBlock>(k, st);
}
hipError_t launch_grad(const KArgs& a, int B, hipStream_t st) {
    KArgs k = a; k.B = B;
    if (use_lane(k)) return launch_lane_m(1, k, st);
    if (use_wave_team(k.G, k.H, k.m)) return k.f16 == 2 ? launch_grad_team<TeamWave, 2>(k, st) : k.f16 ? launch_grad_team<TeamWave, 1>(k, st) : launch_grad_team<TeamWave, 0>(k, st);
    return k.f16 == 2 ? launch_grad_team<TeamBlock, 2>(k, st) : k.f16 ? launch_grad_team<TeamBlock, 1>(k, st) : launch_grad_team<TeamBlock, 0>(k, st);
}
hipError_t launch_solve(const KArgs& a, int B, hipStream_t st) {
    KArgs k = a; k.B = B;
    if (use_lane(k)) return launch_lane_m(2, k, st);
    if (use_wave_team(k.G, k.H, k.m)) return k.f16 == 2 ? launch_solve_team<TeamWave, 2>(k, st) : k.f16 ? launch_solve_team<TeamWave, 1>(k, st) : launch_solve_team<TeamWave, 0>(k, st);
    return k.f16 == 2 ? launch_solve_team<TeamBlock, 2>(k, st) : k.f16 ? launch_solve_team<TeamBlock, 1>(k, st) : launch_solve_team<TeamBlock, 0>(k, st);
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
int solve_workspace_rows_fast(const KArgs& a, int B) { return fastm::solve_workspace_rows(a, B); }
hipError_t launch_solve_coop_fast(const KArgs& a, int B, hipStream_t st) { return fastm::launch_solve_coop(a, B, st); }
hipError_t launch_solve_spec_fast(const KArgs& a, int B, hipStream_t st) { return fastm::launch_solve_spec(a, B, st); }
#else
}  // namespace exact
static thread_local const void* g_last_kernel_fn = nullptr;
void note_kernel(const void* host_fn) { g_last_kernel_fn = host_fn; }
const void* last_launched_kernel() { return g_last_kernel_fn; }
size_t smem_bytes(int H, int m, int ipb) { return exact::smem_bytes(H, m, ipb); }
int team_ipb(int G, int H, int m) { return exact::team_ipb(G, H, m); }
hipError_t launch_rollout(const KArgs& a, int B, hipStream_t st) { return exact::launch_rollout(a, B, st); }
hipError_t launch_grad(const KArgs& a, int B, hipStream_t st) { return exact::launch_grad(a, B, st); }
hipError_t launch_solve(const KArgs& a, int B, hipStream_t st) { return exact::launch_solve(a, B, st); }
int solve_workspace_rows(const KArgs& a, int B) { return exact::solve_workspace_rows(a, B); }
int coop_nwg(int P) { return exact::coop_nwg(P); }
int coop_max_instances(int P, int H, int m, const LaunchOpts& o) { return exact::coop_max_instances(P, H, m, o); }
size_t coop_pp_floats(int H, int G) { return exact::coop_pp_floats(H, G); }
size_t coop_ck_floats(int H, int P) { return exact::coop_ck_floats(H, P); }
hipError_t launch_solve_coop(const KArgs& a, int B, hipStream_t st) { return exact::launch_solve_coop(a, B, st); }
int spec_max_instances(int P, int H, int m, const LaunchOpts& o) { return exact::spec_max_instances(P, H, m, o); }
hipError_t launch_solve_spec(const KArgs& a, int B, hipStream_t st) { return exact::launch_solve_spec(a, B, st); }
hipError_t launch_relayout(bool to_dev, const float* in, float* out, int B, int P, int G, int C, hipStream_t st) {
    return exact::launch_relayout(to_dev, in, out, B, P, G, C, st);
}
#endif
#endif  // SDEMPC_DEV_KERNEL
#endif  // SDEMPC_TU == 0

}  // namespace sdempc
