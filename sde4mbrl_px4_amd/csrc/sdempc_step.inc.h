// sdempc_step.inc.h — one Euler-Maruyama step and its vector-Jacobian product: uniform head / tail, MLPs in the MFMA tile layout, stage cost
// Fragment of sdempc_kernels.hip: included inside namespace sdempc::{exact|fastm} (it is compiled twice, see there); not a
// stand-alone header.
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

// ---- SPEC.md §9b: three-limb bf16 split contraction on the matrix pipe (`mlp_dtype: f32x3`) ----
// A 32-wide contraction out[i] = c[i] + sum_k W[i][k] v[k] with both operands split by truncation into three bf16 limbs
// (x = x1 + x2 + x3 up to 2^-24 |x|) and the six leading limb products accumulated in f32 by v_mfma_f32_32x32x16_bf16, in the fixed order
//     (W3,v1) (W2,v2) (W2,v1) (W1,v3) (W1,v2) (W1,v1),   each as two K = 16 instructions (K-half hf = 0, 1):
// k slot 8 hh + e of K-half hf is hidden unit rowmap(8 hf + e, hh), i.e. accumulator register 8 hf + e of lane half hh. The dropped
// products (W2,v3) (W3,v2) (W3,v3) are below 2^-23 of the leading one: f32-level accuracy, with all twelve instructions on the matrix
// pipe. The instruction's own accumulation (two groups of eight products, fixed-point alignment, SPEC.md §9a) is reproduced by the oracle.
struct Limbs3 { u32x4 l[3][2]; };      // [limb][K-half]: 8 bf16 per lane = this lane half's k slots
DI void split3_tile(const f32x16& v, Limbs3& L) {
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            const float x = v[8 * hf + 2 * pr], y = v[8 * hf + 2 * pr + 1];
            const unsigned x1 = __float_as_uint(x) & 0xFFFF0000u, y1 = __float_as_uint(y) & 0xFFFF0000u;
            const float xr = x - __uint_as_float(x1), yr = y - __uint_as_float(y1);
            const unsigned x2 = __float_as_uint(xr) & 0xFFFF0000u, y2 = __float_as_uint(yr) & 0xFFFF0000u;
            const float xs = xr - __uint_as_float(x2), ys = yr - __uint_as_float(y2);
            // two bf16 per dword: the high halves of (x, y) -> {x.hi16, y.hi16}
            L.l[0][hf][pr] = __builtin_amdgcn_perm(y1, x1, 0x07060302u);
            L.l[1][hf][pr] = __builtin_amdgcn_perm(y2, x2, 0x07060302u);
            L.l[2][hf][pr] = __builtin_amdgcn_perm(__float_as_uint(ys), __float_as_uint(xs), 0x07060302u);
        }
    }
}
// acc += sum of the six limb products; Aimg: this layer's A-operand image in LDS, [limb][K-half][lane][8 x bf16] (load_weights)
DI void mfma_x3(const float* Aimg, int lane, const Limbs3& L, f32x16& acc) {
    constexpr int WA[6] = {2, 1, 1, 0, 0, 0}, VB[6] = {0, 1, 0, 2, 1, 0};     // limb indices (0-based) of the six products, in order
    u32x4 aw[2];
#pragma unroll
    for (int s6 = 0; s6 < 6; ++s6) {
        if (s6 == 0 || WA[s6] != WA[s6 - 1]) {     // the weight limb changes three times: six 16-byte LDS reads per contraction
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) aw[hf] = *reinterpret_cast<const u32x4*>(Aimg + ((WA[s6] * 2 + hf) * 64 + lane) * 4);
        }
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aw[hf]), __builtin_bit_cast(bf16x8, L.l[VB[s6]][hf]), acc, 0, 0, 0);
    }
}

// ---- SPEC.md §10c / §10e: contractions from TWO binary16 limbs of either operand (math_mode fast + mlp_dtype f32x3) ----
// A tile's sixteen values per lane split by round to nearest (v_cvt_pk_f16_f32 packs two values per instruction, v_fma_mix_f32 forms the exact residual
// x - limb straight from the packed half): 2 instructions per value, against 5.5 for the three-limb bf16 split.
typedef _Float16 h8x __attribute__((ext_vector_type(8)));
struct Limbs2 { u32x4 r1[2], r2[2]; };      // [K-half]: 8 binary16 per lane = this lane half's k slots; r1 the leading limb
DI void split2h_tile(const f32x16& v, Limbs2& L) {
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            const float x = v[8 * hf + 2 * pr], y = v[8 * hf + 2 * pr + 1];
            unsigned p1, p2; float xr, yr;
            asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p1) : "v"(x), "v"(y));
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(xr) : "v"(p1), "v"(x));                      // x - (float)p1.lo, exact
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(yr) : "v"(p1), "v"(y));     // y - (float)p1.hi
            asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p2) : "v"(xr), "v"(yr));
            L.r1[hf][pr] = p1; L.r2[hf][pr] = p2;
        }
}
// acc += the four limb products (w2,v2) (w2,v1) (w1,v2) (w1,v1), each over the two K halves: eight v_mfma_f32_32x32x16_f16. aw[limb][K-half]: the weight fragments
DI void mfma_2h(const u32x4 (&aw)[2][2], const Limbs2& L, f32x16& acc) {
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8x, aw[s4 < 2 ? 1 : 0][hf]), __builtin_bit_cast(h8x, (s4 & 1) ? L.r1[hf] : L.r2[hf]), acc, 0, 0, 0);
}
// weight fragments of a full 32-row image [limb][K-half][lane][8 x binary16] (load_weights: sm.A2x forward, sm.A2xT transposed)
DI void load_aw_full(const float* img, int lane, u32x4 (&aw)[2][2]) {
#pragma unroll
    for (int lb = 0; lb < 2; ++lb)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) aw[lb][hf] = *reinterpret_cast<const u32x4*>(img + ((lb * 2 + hf) * 64 + lane) * 4);
}
// SPEC.md §10e: the adjoint's two narrow contractions (6 rows: W1z^T abar1n; 6 + m rows: [W1z; W1u]^T abar1d) take their A operand from COMPACT images
// [limb][K-half][lane half][ROWS][8 x binary16] whose last row is zero: lanes whose output row does not exist read that row
// (ZROWS_D / ZROWS_N rows, at sm.A2 + adj_off(F16).azd / .azn: sdempc_kernels.hip, beside Smem)
template <int ROWS>
DI void load_aw_rows(const float* img, int lane, u32x4 (&aw)[2][2]) {
    const int row = lane & 31, hh = lane >> 5, rr = row < ROWS - 1 ? row : ROWS - 1;
#pragma unroll
    for (int lb = 0; lb < 2; ++lb)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) aw[lb][hf] = *reinterpret_cast<const u32x4*>(img + ((((lb * 2 + hf) * 2 + hh) * ROWS) + rr) * 4);
}
// One power of two per particle brings the seven output adjoints of the MLPs to 2^eoff (the largest of them into [2^eoff, 2^(eoff+1)) once the -2 of the
// forward output weights is in): everything between here and the unscaling is linear in them, every scaling is exact, and the bounded quantities can
// take binary16 limbs. s2 = -2 * 2^(eoff - e) multiplies the adjoints, inv = 2^(e - eoff) the results; KArgs::M carries -2 * 2^eoff and 2^-eoff (sdempc_create).
struct AdjScale { float s2, inv; };
DI AdjScale adj_scale(const KArgs& a, float ebraw, const float* ob) {
    float mx = fmaxf(fabsf(ebraw), fabsf(ob[0]));
#pragma unroll
    for (int i = 1; i < 6; ++i) mx = fmaxf(mx, fabsf(ob[i]));
    int e = __builtin_amdgcn_frexp_expf(mx);          // 0 for zero, infinity and NaN
    e = e < -100 ? -100 : e;
    AdjScale S;
    S.s2 = __builtin_amdgcn_ldexpf(a.M.adj_s0, -e);
    S.inv = __builtin_amdgcn_ldexpf(a.M.adj_i0, e);
    return S;
}
// The six scaled output adjoints of a particle as two binary16 limbs each, packed in pairs: the B operand (k slots 0..5) of the K = 6 contraction with (-2 W3)^T
struct ObLimbs { unsigned r1[3], r2[3]; };
DI ObLimbs ob_limbs(const float* ob) {
    ObLimbs O;
#pragma unroll
    for (int pr = 0; pr < 3; ++pr) {
        const float x = ob[2 * pr], y = ob[2 * pr + 1];
        unsigned p1, p2; float xr, yr;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p1) : "v"(x), "v"(y));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(xr) : "v"(p1), "v"(x));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(yr) : "v"(p1), "v"(y));
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p2) : "v"(xr), "v"(yr));
        O.r1[pr] = p1; O.r2[pr] = p2;
    }
    return O;
}
// Results of the narrow contractions in the accumulator layout: register r of lane half hh holds output row rowmap(r, hh) — rows 0..3 in registers 0..3 of the
// lower half, 4..7 in registers 0..3 of the upper half, 8..11 in registers 4..7 of the lower half. Row k < 6 is zbar_k, row 6 + j is (W1u^T abar1d)_j.
// One group per wave (both lane halves hold the same particle): a swap of a register with its copy leaves {lower, lower} and {upper, upper}.
// (row -> register / lane half; ADJ_REGS<M>: accumulator registers that hold an existing row)
DI constexpr int adj_row_reg(int row) { return (row >> 3) * 4 + (row & 3); }
DI constexpr bool adj_row_upper(int row) { return ((row >> 2) & 1) != 0; }
template <int M> struct AdjRegs { static constexpr int N = M <= 2 ? 4 : (M <= 6 ? M + 2 : 8); };
template <int M>
DI void adj_rows_tile(const float* accZ, float inv, float* zb, float* gq) {
    float lo[8], hi[8];
#pragma unroll
    for (int r = 0; r < AdjRegs<M>::N; ++r) {
        unsigned a0 = __builtin_bit_cast(unsigned, accZ[r]), b0;
        asm("v_mov_b32 %0, %1" : "=v"(b0) : "v"(a0));
        auto sw = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
        const unsigned s0 = sw[0], s1 = sw[1];
        lo[r] = __builtin_bit_cast(float, s0); hi[r] = __builtin_bit_cast(float, s1);
    }
#pragma unroll
    for (int k = 0; k < NN; ++k) zb[k] = (adj_row_upper(k) ? hi[adj_row_reg(k)] : lo[adj_row_reg(k)]) * inv;
#pragma unroll
    for (int jj = 0; jj < M; ++jj) gq[jj] = (adj_row_upper(NN + jj) ? hi[adj_row_reg(NN + jj)] : lo[adj_row_reg(NN + jj)]) * inv;
}
// the three contractions of one 32-particle pass (SPEC.md §10e); ebraw / ob already carry the scale; hn / hd / h2: the r-tiles of the density net's and the drift net's
// first layer and of the drift net's second layer (from registers, a recompute or the checkpoint: callers differ). The density tile is consumed first.
// The density tile's contraction has six rows: registers 0..3 of its result — rows 0..7 — are added to the drift tile's rows (SPEC.md §10e states the sum that
// way; one float32 addition each). It runs LAST in a pass: by then the pass's second-layer checkpoint and the drift tiles are dead, and pass B's checkpoint, which
// is in flight since before pass A, keeps its registers (requested a phase earlier this loop spilled it: three synchronous HBM round trips per step).
template <int F16>
DI void adj_mp_density(const Smem& sm, int h, int lane, f32x16& hn, float ebraw, float* rows) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float4 wn4 = *reinterpret_cast<const float4*>(sm.w3n + 8 * q + 4 * h);
        hn[4 * q] = (wn4.x * ebraw) * dact(hn[4 * q]); hn[4 * q + 1] = (wn4.y * ebraw) * dact(hn[4 * q + 1]);
        hn[4 * q + 2] = (wn4.z * ebraw) * dact(hn[4 * q + 2]); hn[4 * q + 3] = (wn4.w * ebraw) * dact(hn[4 * q + 3]);
    }
    SCHED_PHASE();
    Limbs2 L;
    split2h_tile(hn, L);
    u32x4 aw[2][2];
    load_aw_rows<ZROWS_N>(sm.A2 + adj_off(F16).azn, lane, aw);
    f32x16 accZ;
#pragma unroll
    for (int r = 0; r < 16; ++r) accZ[r] = 0.0f;
    mfma_2h(aw, L, accZ);
#pragma unroll
    for (int r = 0; r < 4; ++r) rows[r] = rows[r] + accZ[r];
}
template <int F16, class H2>
DI void adj_mp_layer2(const Smem& sm, int lane, const H2& h2_of, const ObLimbs& O, f32x16& accB) {
    // abar2 = ((-2 W3)^T obar') (r2 - r2^2): the K = 6 contraction as four v_mfma_f32_32x32x16_f16 — limb products (w2,o2) (w2,o1) (w1,o2) (w1,o1), k slots 0..5 of the
    // lower lane half; the upper half's A operand is the zero row (its k slots 8..15 do not exist) — straight into the accumulator layout the rest of the pass works in
    const int rr = (lane >> 5) ? A3T_ROWS - 1 : (lane & 31);
    const u32x4 a1 = *reinterpret_cast<const u32x4*>(sm.A2 + adj_off(F16).a3t + (0 * A3T_ROWS + rr) * 4);
    const u32x4 a2 = *reinterpret_cast<const u32x4*>(sm.A2 + adj_off(F16).a3t + (1 * A3T_ROWS + rr) * 4);
    u32x4 b1, b2;
    b1[0] = O.r1[0]; b1[1] = O.r1[1]; b1[2] = O.r1[2]; b1[3] = 0u;
    b2[0] = O.r2[0]; b2[1] = O.r2[1]; b2[2] = O.r2[2]; b2[3] = 0u;
    f32x16 a2b;
#pragma unroll
    for (int r = 0; r < 16; ++r) a2b[r] = 0.0f;
    a2b = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8x, a2), __builtin_bit_cast(h8x, b2), a2b, 0, 0, 0);
    a2b = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8x, a2), __builtin_bit_cast(h8x, b1), a2b, 0, 0, 0);
    a2b = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8x, a1), __builtin_bit_cast(h8x, b2), a2b, 0, 0, 0);
    a2b = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8x, a1), __builtin_bit_cast(h8x, b1), a2b, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 hq = h2_of(q);
        a2b[4 * q] = a2b[4 * q] * dact(hq.x); a2b[4 * q + 1] = a2b[4 * q + 1] * dact(hq.y); a2b[4 * q + 2] = a2b[4 * q + 2] * dact(hq.z); a2b[4 * q + 3] = a2b[4 * q + 3] * dact(hq.w);
    }
    SCHED_PHASE();
    Limbs2 L;
    split2h_tile(a2b, L);
    u32x4 aw[2][2];
    load_aw_full(sm.A2 + adj_off(F16).xt, lane, aw);
#pragma unroll
    for (int r = 0; r < 16; ++r) accB[r] = 0.0f;
    mfma_2h(aw, L, accB);
}
template <int NR, int F16>
DI void adj_mp_drift(const Smem& sm, int lane, const f32x16& hd, const f32x16& accB, float* rows) {
    u32x4 aw[2][2];
    f32x16 ad;
#pragma unroll
    for (int r = 0; r < 16; ++r) ad[r] = accB[r] * dact(hd[r]);
    load_aw_rows<ZROWS_D>(sm.A2 + adj_off(F16).azd, lane, aw);
    Limbs2 L;
    split2h_tile(ad, L);
    f32x16 accZ;
#pragma unroll
    for (int r = 0; r < 16; ++r) accZ[r] = 0.0f;
    mfma_2h(aw, L, accZ);
#pragma unroll
    for (int r = 0; r < NR; ++r) rows[r] = accZ[r];
}

// ---- MLPs of a step in the MFMA tile layout (32 particles per wave) ----
// fwd_mlp_partials: the hidden tiles (A.h1d, A.h1n, A.h2) and the per-half partial chains of the seven output-layer dot products
// (Po[0..5]: residual force / torque, Po[6]: density pre-activation); fwd_mlp_tiles adds the halves: (P0 + P1) + bias (SPEC.md §5.2).
// OB: how many of the seven output-layer weight quads of a quarter are requested from LDS together (7: all; the gradient's forward sweep,
// which also carries the noise prefetch and the checkpoint stream, takes 4 + 3: with all seven in flight its noise prefetch spilled)
// layer-1 C operands: per-step offsets (drift tile), bias (density tile)
DI void load_l1_c(const Smem& sm, const float* ust, int h, f32x16& accD, f32x16& accN) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float4 c4 = *reinterpret_cast<const float4*>(ust + 8 * q + 4 * h);
        float4 n4 = *reinterpret_cast<const float4*>(sm.b1n + 8 * q + 4 * h);
        accD[4 * q] = c4.x; accD[4 * q + 1] = c4.y; accD[4 * q + 2] = c4.z; accD[4 * q + 3] = c4.w;
        accN[4 * q] = n4.x; accN[4 * q + 1] = n4.y; accN[4 * q + 2] = n4.z; accN[4 * q + 3] = n4.w;
    }
}
template <int F16, bool PK, int OB = 7>
DI void fwd_mlp_partials(const KArgs& a, const Smem& sm, const WaveW& ww, const float* ust, int h, int lane, const float* z, StepAux& A, float* Po) {
    // layer 1: C operand = per-step offsets (drift) / bias (density); K = 6 -> 3 MFMAs per tile
    f32x16 accD, accN;
    load_l1_c(sm, ust, h, accD, accN);
    if constexpr (F16 == 1) {
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
    if constexpr (F16 == 1) {
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
    } else if constexpr (F16 == 2) {
        if constexpr (FAST) {
            // SPEC.md §10c: math_mode fast keeps activations in [0, 1], so the forward contraction splits them into TWO binary16 limbs by round to
            // nearest (v_cvt_pk_f16_f32 packs two values per instruction, v_fma_mix_f32 forms the exact residual x - limb straight from the packed
            // half): 2 instructions per value instead of 5.5, and eight v_mfma_f32_32x32x16_f16 on the limb products (w2,r2) (w2,r1) (w1,r2) (w1,r1)
            // instead of twelve bf16 ones. The four weight fragments (load_weights: two binary16 limbs of the forward weights) are requested first
            // and fly under the split.
            u32x4 aw[2][2];
            load_aw_full(sm.A2x, lane, aw);
            SCHED_PHASE();
            Limbs2 L;
            split2h_tile(accD, L);
            SCHED_PHASE();
            mfma_2h(aw, L, acc2);
        } else {
            Limbs3 L;
            split3_tile(accD, L);
            mfma_x3(sm.A2x, lane, L, acc2);
        }
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 w4 = *reinterpret_cast<const float4*>(sm.A2 + (q * 64 + lane) * 4);      // (requesting quad q+1 before these MFMAs: -4 % while the loop was short of registers, neutral since)
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.x, accD[4 * q], acc2, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.y, accD[4 * q + 1], acc2, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.z, accD[4 * q + 2], acc2, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.w, accD[4 * q + 3], acc2, 0, 0, 0);
        }
    }
    SCHED_PHASE();

    // math_mode fast: the seven weight quads of quarter q + 1 are requested before the fmas of quarter q (quarter 0: before the layer-2 activation) —
    // a second buffer of 28 registers the exact mode's loops do not have; + 0.3 % (tools/ab_power.sh), same operations in the same order
    if constexpr (FAST) {
        float4 wb[2][7];
        auto req = [&](int q, float4* w) {
#pragma unroll
            for (int i = 0; i < 6; ++i) w[i] = *reinterpret_cast<const float4*>(sm.W3 + i * HID + 8 * q + 4 * h);
            w[6] = *reinterpret_cast<const float4*>(sm.w3n + 8 * q + 4 * h);
        };
        req(0, wb[0]);
        SCHED_PHASE();
        tanh_tile<PK>(acc2);
        A.h2 = acc2;
        SCHED_PHASE();
#pragma unroll
        for (int i = 0; i < 7; ++i) Po[i] = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q < 3) req(q + 1, wb[(q + 1) & 1]);
            SCHED_PHASE();
            const float4* w = wb[q & 1];
#pragma unroll
            for (int io = 0; io < 6; ++io) {
                Po[io] = FMA(w[io].x, acc2[4 * q], Po[io]); Po[io] = FMA(w[io].y, acc2[4 * q + 1], Po[io]); Po[io] = FMA(w[io].z, acc2[4 * q + 2], Po[io]); Po[io] = FMA(w[io].w, acc2[4 * q + 3], Po[io]);
            }
            Po[6] = FMA(w[6].x, accN[4 * q], Po[6]); Po[6] = FMA(w[6].y, accN[4 * q + 1], Po[6]); Po[6] = FMA(w[6].z, accN[4 * q + 2], Po[6]); Po[6] = FMA(w[6].w, accN[4 * q + 3], Po[6]);
            SCHED_PHASE();
        }
        return;
    }
    tanh_tile<PK>(acc2);
    A.h2 = acc2;
    SCHED_PHASE();

    // output layers on the VALU: per-half partial chains (six residual outputs over h2, the density output over h1n). The seven weight
    // quads of a quarter are requested from LDS together and only then consumed: one exposed LDS round trip per quarter instead of one per
    // quad (the compiler, left alone, re-used four registers and waited after every read: 28 serialised round trips per pass)
    {
#pragma unroll
        for (int i = 0; i < 7; ++i) Po[i] = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int i0 = 0; i0 < 7; i0 += OB) {
                float4 w[OB];
#pragma unroll
                for (int i = 0; i < OB; ++i) {
                    if (i0 + i < 6) w[i] = *reinterpret_cast<const float4*>(sm.W3 + (i0 + i) * HID + 8 * q + 4 * h);
                    else if (i0 + i == 6) w[i] = *reinterpret_cast<const float4*>(sm.w3n + 8 * q + 4 * h);
                }
                SCHED_PHASE();
#pragma unroll
                for (int i = 0; i < OB; ++i) {
                    const int io = i0 + i;
                    if (io < 6) {
                        Po[io] = FMA(w[i].x, acc2[4 * q], Po[io]); Po[io] = FMA(w[i].y, acc2[4 * q + 1], Po[io]); Po[io] = FMA(w[i].z, acc2[4 * q + 2], Po[io]); Po[io] = FMA(w[i].w, acc2[4 * q + 3], Po[io]);
                    } else if (io == 6) {
                        Po[6] = FMA(w[i].x, accN[4 * q], Po[6]); Po[6] = FMA(w[i].y, accN[4 * q + 1], Po[6]); Po[6] = FMA(w[i].z, accN[4 * q + 2], Po[6]); Po[6] = FMA(w[i].w, accN[4 * q + 3], Po[6]);
                    }
                }
                SCHED_PHASE();
            }
        }
    }
}
template <int F16, bool PK>
DI void fwd_mlp_tiles(const KArgs& a, const Smem& sm, const WaveW& ww, const float* ust, int h, int lane, const float* z, StepAux& A, float* o, float& eta_out) {
    float Po[7];
    fwd_mlp_partials<F16, PK>(a, sm, ww, ust, h, lane, z, A, Po);
#pragma unroll
    for (int i = 0; i < 6; ++i) o[i] = xor32_sum(Po[i]) + a.M.b3[i];
    eta_out = sigmoid_spec(xor32_sum(Po[6]) + a.M.b3n);
    SCHED_PHASE();
}

// ---- uniform tail of a step: rigid body, Euler-Maruyama update, quaternion renormalisation ----
// (the _v forms take the step's constants as values / pointers of the caller's choice: tz = {Tz, tau0, tau1, tau2}, sdt = sigma_i sqrt(dt_t))
// (_ft: body force A.Fb and body torque taub already formed by the caller — the lane layouts scale the MLP outputs in the lanes that hold them)
DI void fwd_tail_ft(const KArgs& a, const float dt, const float* taub, const float* sdt, const float* x, const float* xi, const float* Rm, float eta, float* xn, StepAux& A) {
    const float qw = x[6], qx = x[7], qy = x[8], qz = x[9];
    A.eta = eta;
    float acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float Fw = FMA(Rm[3 * i + 2], A.Fb[2], FMA(Rm[3 * i + 1], A.Fb[1], Rm[3 * i] * A.Fb[0]));
        acc[i] = Fw * a.M.inv_mass;
    }
    acc[2] = acc[2] - a.M.grav;
#pragma unroll
    for (int i = 0; i < 3; ++i) A.Jom[i] = a.M.J[i] * x[10 + i];
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
DI void fwd_tail_v(const KArgs& a, const float dt, const float* tz, const float* sdt, const float* x, const float* xi, const float* Rm, const float* o, float eta, float* xn, StepAux& A) {
    A.Fb[0] = a.M.sF[0] * o[0]; A.Fb[1] = a.M.sF[1] * o[1]; A.Fb[2] = FMA(a.M.sF[2], o[2], tz[0]);
    float taub[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) taub[i] = FMA(a.M.sT[i], o[3 + i], tz[1 + i]);
    fwd_tail_ft(a, dt, taub, sdt, x, xi, Rm, eta, xn, A);
}
DI void fwd_tail(const KArgs& a, const Smem& sm, const float* ust, int t, const float* x, const float* xi, const float* Rm, const float* o, float eta, float* xn, StepAux& A) {
    fwd_tail_v(a, sm.dt[t], ust + 32, sm.sdt + t * NN, x, xi, Rm, o, eta, xn, A);
}

template <int F16, bool PK = false>
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
// SC: with the state-bound terms (tile layouts). The lane layouts (single particle, cooperative, speculative) are issue-bound — every
// instruction of a step is on the critical path, and even the never-taken branch measured +10 % on a single solve — so they are built
// without them and the C ABI keeps instances with state bounds on the tile layouts.
template <bool GX, bool SC = true>
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
    // state_constr, penalty form (SPEC.md §5.3). Cold path: ONE wave-uniform branch, a rolled loop over the bounded states (ascending index)
    // whose table is read from memory there, and the state picked by uniform selects — the common case (no bounds) must not pay registers or
    // basic blocks for it (an unrolled per-state version cost the throughput kernel a third of its speed through register pressure alone).
    if (SC && __builtin_expect(a.C.sc_n != 0, 0)) {
#pragma nounroll
        for (int k = 0; k < a.C.sc_n; ++k) {
            const int i = a.C.sc_tab[k].id;
            const float w = a.C.sc_tab[k].w;
            float xi = x[0];
#pragma unroll
            for (int jx = 1; jx < NX; ++jx) xi = (i == jx) ? x[jx] : xi;
            float hi = xi - a.C.sc_tab[k].hi; hi = hi < 0.0f ? 0.0f : hi;
            float lo = a.C.sc_tab[k].lo - xi; lo = lo < 0.0f ? 0.0f : lo;
            l = FMA(w * hi, hi, l);
            l = FMA(w * lo, lo, l);
            if (GX) {
                const float gnew = 2.0f * w;
#pragma unroll
                for (int jx = 0; jx < NX; ++jx) gx[jx] = (i == jx) ? FMA(gnew, hi - lo, gx[jx]) : gx[jx];
            }
        }
    }
    return l;
}

// ------------------------------------------------------------------------------------------------
// vector-Jacobian product of one step (SPEC.md §5.4). gq[0..m-1] = W1u^T abar1, gq[m] = Tz adjoint,
// gq[m+1..m+3] = rotor-torque adjoint
// ------------------------------------------------------------------------------------------------
// Uniform (per particle) quantities that the three parts of the step's vector-Jacobian product share
struct VjpTmp {
    float ebraw, qtb[4], omb[3], Fwb[3], ob[6];      // (dqb = qtb * dt is formed in the tail: four registers less across the MLP part)
};

// ---- head: everything upstream of the MLPs (per particle); gq[M..M+3] = thrust / rotor-torque adjoints ----
template <int M>
DI void vjp_head_v(const KArgs& a, const float dt, const float* sdt, const float* x, const float* xi, const StepAux& A, const float* L, float etabar_cost, VjpTmp& T, float* gq) {
    const float* Rm = A.Rm;
    const float* om = x + 10;
    float eb = etabar_cost;
#pragma unroll
    for (int i = 0; i < 3; ++i) eb = FMA(L[3 + i] * sdt[i], xi[i], eb);
#pragma unroll
    for (int i = 0; i < 3; ++i) eb = FMA(L[10 + i] * sdt[3 + i], xi[3 + i], eb);
    T.ebraw = eb * (A.eta * (1.0f - A.eta));
    float dotq = FMA(A.qn[3], L[9], FMA(A.qn[2], L[8], FMA(A.qn[1], L[7], A.qn[0] * L[6])));
    float* qtb = T.qtb;
#pragma unroll
    for (int i = 0; i < 4; ++i) qtb[i] = A.rn * FMA(-A.qn[i], dotq, L[6 + i]);
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
template <int M>
DI void vjp_head(const KArgs& a, const Smem& sm, int t, const float* x, const float* xi, const StepAux& A, const float* L, float etabar_cost, VjpTmp& T, float* gq) {
    vjp_head_v<M>(a, sm.dt[t], sm.sdt + t * NN, x, xi, A, L, etabar_cost, T, gq);
}

// ---- MLP part in the MFMA tile layout: zb[6] = adjoint of z, gq[0..M-1] = W1u^T abar1 (per particle) ----
// vjp_mlp_partials leaves the per-half partial chains Pz[6], Pu[M]; vjp_mlp_tiles adds the halves.
template <int M, int F16 = 0>
DI void vjp_mlp_partials(const Smem& sm, int h, int lane, const StepAux& A, float ebraw_in, const float* ob_in, float* Pz, float* Pu) {
    // (math_mode fast: sm.W3 / sm.w3n hold the forward pass's -2 W3 / -2 w3n, the adjoint wants 4 W3: the exact factor -2 goes onto the seven output adjoints)
    float ob[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) ob[i] = FAST ? -2.0f * ob_in[i] : ob_in[i];
    const float ebraw = FAST ? -2.0f * ebraw_in : ebraw_in;
    // MLP VJP. Order chosen to keep few tiles live: density tile first (frees h1n), then the drift
    // tile: abar2 on the VALU, W2^T abar2 by MFMA in the accumulator layout.
    {
#pragma unroll
        for (int k = 0; k < NN; ++k) Pz[k] = 0.0f;
#pragma unroll
        for (int jj = 0; jj < M; ++jj) Pu[jj] = 0.0f;
        // density net: abar1n = (w3n * etaraw_bar) * (1 - h1n^2); zbar += W1z[32:64]^T abar1n
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 wn4 = *reinterpret_cast<const float4*>(sm.w3n + 8 * q + 4 * h);
            float an0 = (wn4.x * ebraw) * dact(A.h1n[4 * q]);
            float an1 = (wn4.y * ebraw) * dact(A.h1n[4 * q + 1]);
            float an2 = (wn4.z * ebraw) * dact(A.h1n[4 * q + 2]);
            float an3 = (wn4.w * ebraw) * dact(A.h1n[4 * q + 3]);
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
            a2b[4 * q] = hb0 * dact(A.h2[4 * q]);
            a2b[4 * q + 1] = hb1 * dact(A.h2[4 * q + 1]);
            a2b[4 * q + 2] = hb2 * dact(A.h2[4 * q + 2]);
            a2b[4 * q + 3] = hb3 * dact(A.h2[4 * q + 3]);
            SCHED_PHASE();
        }
        f32x16 accB;
#pragma unroll
        for (int r = 0; r < 16; ++r) accB[r] = 0.0f;
        if constexpr (F16 == 2) {       // SPEC.md §9b: W2^T abar2 as the three-limb bf16 split on the matrix pipe
            Limbs3 L;
            split3_tile(a2b, L);
            mfma_x3(sm.A2xT, lane, L, accB);
        } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 w4 = *reinterpret_cast<const float4*>(sm.A2T + (q * 64 + lane) * 4);
            accB = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.x, a2b[4 * q], accB, 0, 0, 0);
            accB = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.y, a2b[4 * q + 1], accB, 0, 0, 0);
            accB = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.z, a2b[4 * q + 2], accB, 0, 0, 0);
            accB = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.w, a2b[4 * q + 3], accB, 0, 0, 0);
        }
        }
        SCHED_PHASE();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float ad0 = accB[4 * q] * dact(A.h1d[4 * q]);
            float ad1 = accB[4 * q + 1] * dact(A.h1d[4 * q + 1]);
            float ad2 = accB[4 * q + 2] * dact(A.h1d[4 * q + 2]);
            float ad3 = accB[4 * q + 3] * dact(A.h1d[4 * q + 3]);
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
    }
}
// One layer-1 tile (3 f32 MFMAs, or one fp16 MFMA) on top of its C operand, then tanh: the drift tile (DRIFT: C = per-step control
// terms c_t, A = W1z rows 0..31) or the density tile (C = b1 rows 32..63, A = W1z rows 32..63)
template <int F16, bool DRIFT>
DI void layer1_tile(const Smem& sm, const WaveW& ww, const float* ust, int h, const float* z, f32x16& acc) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float4 c4 = *reinterpret_cast<const float4*>((DRIFT ? ust : sm.b1n) + 8 * q + 4 * h);
        acc[4 * q] = c4.x; acc[4 * q + 1] = c4.y; acc[4 * q + 2] = c4.z; acc[4 * q + 3] = c4.w;
    }
    if constexpr (F16 == 1) {
        half8 bv;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            auto pk = __builtin_amdgcn_cvt_pkrtz(h ? 0.0f : z[2 * e], h ? 0.0f : z[2 * e + 1]);
            bv[2 * e] = (_Float16)pk[0]; bv[2 * e + 1] = (_Float16)pk[1];
        }
        bv[6] = (_Float16)0.0f; bv[7] = (_Float16)0.0f;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(DRIFT ? ww.h1d : ww.h1n, bv, acc, 0, 0, 0);
    } else {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            float bb = h ? z[2 * s + 1] : z[2 * s];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(DRIFT ? ww.w1d[s] : ww.w1n[s], bb, acc, 0, 0, 0);
        }
    }
    SCHED_PHASE();
    tanh_tile<false>(acc);
}

// SPEC.md §10e (math_mode fast + f32x3): the same pass with the scaled adjoints and its three contractions on the matrix pipe; leaves the result tile (rows 0..5 zbar,
// 6.. W1u^T abar1d, still scaled) as the NR accumulator registers that hold existing rows. One hidden tile at a time, like adj_mlp_pass.
// DENS_FIRST: the density tile before the drift net (pass B of the duo layout); otherwise the drift net first (pass A: its checkpoint is already there). The two tiles'
// rows are added (commutative: the same bits either way). load_ckpt: called right before the second-layer phase — pass B loads its checkpoint THERE, into registers
// that live for one phase only: its lines were touched (one dword each) at the top of the step, a pass and a half earlier, so the load is an L2 hit. Holding the tile in
// registers from before pass A, as the f32 / three-limb forms do, leaves the allocator the choice of what to spill, and it chooses the checkpoint itself: a synchronous
// HBM round trip per spilled quad and step — 0 to 2 of them, changing with every source line (profiles/r5_ab.txt §2 – §3).
template <int NR, bool DENS_FIRST, int F16, class Hook>
DI void adj_mlp_pass_mp(const Smem& sm, const WaveW& ww, const float* ust, int h, int lane, const float* z, const float4* h2c, float ebraw, const ObLimbs& O, float* rows, const Hook& load_ckpt) {
    float rn[4];
    if constexpr (DENS_FIRST) {
#pragma unroll
        for (int r = 0; r < 4; ++r) rn[r] = 0.0f;
        f32x16 hn;
        layer1_tile<F16, false>(sm, ww, ust, h, z, hn);
        adj_mp_density<F16>(sm, h, lane, hn, ebraw, rn);
        SCHED_PHASE();
    }
    load_ckpt();
    SCHED_PHASE();
    f32x16 accB;
    adj_mp_layer2<F16>(sm, lane, [&](int q) { return h2c[q]; }, O, accB);
    SCHED_PHASE();
    {
        f32x16 hd;
        layer1_tile<F16, true>(sm, ww, ust, h, z, hd);
        adj_mp_drift<NR, F16>(sm, lane, hd, accB, rows);
    }
    if constexpr (DENS_FIRST) {
#pragma unroll
        for (int r = 0; r < 4; ++r) rows[r] = rows[r] + rn[r];
    } else {
        SCHED_PHASE();
        f32x16 hn;
        layer1_tile<F16, false>(sm, ww, ust, h, z, hn);
        adj_mp_density<F16>(sm, h, lane, hn, ebraw, rows);
    }
}

// The adjoint's MLP work of one 32-particle pass with short live ranges (one hidden tile at a time): density tile recomputed and
// consumed, then abar2 from the checkpointed second layer (h2c: this lane's four float4 of the tile), W2^T abar2 by MFMA, and only
// then the drift layer-1 tile recomputed and consumed. Same operations and the same chain order per value as step_fwd's layer 1 +
// vjp_mlp_partials (density tile first, then the drift tile: SPEC.md §5.4), hence the same bits; peak 32 tile registers instead of 64.
template <int M, int F16>
DI void adj_mlp_pass(const Smem& sm, const WaveW& ww, const float* ust, int h, int lane, const float* z, const float4* h2c,
                     float ebraw_in, const float* ob_in, float* Pz, float* Pu) {
    float ob[6];       // (math_mode fast: -2 onto the output adjoints, see vjp_mlp_partials)
#pragma unroll
    for (int i = 0; i < 6; ++i) ob[i] = FAST ? -2.0f * ob_in[i] : ob_in[i];
    const float ebraw = FAST ? -2.0f * ebraw_in : ebraw_in;
#pragma unroll
    for (int k = 0; k < NN; ++k) Pz[k] = 0.0f;
#pragma unroll
    for (int jj = 0; jj < M; ++jj) Pu[jj] = 0.0f;
    {   // density net: abar1n = (w3n * etaraw_bar) * (1 - h1n^2); zbar += W1z[32:64]^T abar1n
        f32x16 hn;
        layer1_tile<F16, false>(sm, ww, ust, h, z, hn);
        // (weight quads of a quarter are requested from LDS together, then consumed: one exposed LDS round trip per quarter, see fwd_mlp_partials)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 wn4 = *reinterpret_cast<const float4*>(sm.w3n + 8 * q + 4 * h);
            float an0, an1, an2, an3;
#pragma unroll
            for (int k0 = 0; k0 < NN; k0 += 3) {       // three weight quads in flight at a time (the adjoint has no registers for six)
                float4 wz[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) wz[k] = *reinterpret_cast<const float4*>(sm.W1zT + (k0 + k) * 2 * HID + HID + 8 * q + 4 * h);
                SCHED_PHASE();
                if (k0 == 0) {
                    an0 = (wn4.x * ebraw) * dact(hn[4 * q]);
                    an1 = (wn4.y * ebraw) * dact(hn[4 * q + 1]);
                    an2 = (wn4.z * ebraw) * dact(hn[4 * q + 2]);
                    an3 = (wn4.w * ebraw) * dact(hn[4 * q + 3]);
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    Pz[k0 + k] = FMA(wz[k].x, an0, Pz[k0 + k]); Pz[k0 + k] = FMA(wz[k].y, an1, Pz[k0 + k]); Pz[k0 + k] = FMA(wz[k].z, an2, Pz[k0 + k]); Pz[k0 + k] = FMA(wz[k].w, an3, Pz[k0 + k]);
                }
                SCHED_PHASE();
            }
        }
    }
    f32x16 accB;
    {   // drift net, second layer: abar2 = (W3^T obar) * (1 - h2^2) on the VALU, W2^T abar2 by MFMA in the accumulator layout
        f32x16 a2b;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float hb0 = 0.0f, hb1 = 0.0f, hb2 = 0.0f, hb3 = 0.0f;
#pragma unroll
            for (int i0 = 0; i0 < 6; i0 += 3) {
                float4 w3[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) w3[i] = *reinterpret_cast<const float4*>(sm.W3 + (i0 + i) * HID + 8 * q + 4 * h);
                SCHED_PHASE();
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    hb0 = FMA(w3[i].x, ob[i0 + i], hb0); hb1 = FMA(w3[i].y, ob[i0 + i], hb1); hb2 = FMA(w3[i].z, ob[i0 + i], hb2); hb3 = FMA(w3[i].w, ob[i0 + i], hb3);
                }
            }
            const float4 hq = h2c[q];
            a2b[4 * q] = hb0 * dact(hq.x);
            a2b[4 * q + 1] = hb1 * dact(hq.y);
            a2b[4 * q + 2] = hb2 * dact(hq.z);
            a2b[4 * q + 3] = hb3 * dact(hq.w);
            SCHED_PHASE();
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) accB[r] = 0.0f;
        if constexpr (F16 == 2) {       // SPEC.md §9b: W2^T abar2 as the three-limb bf16 split on the matrix pipe
            Limbs3 L;
            split3_tile(a2b, L);
            mfma_x3(sm.A2xT, lane, L, accB);
        } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 w4 = *reinterpret_cast<const float4*>(sm.A2T + (q * 64 + lane) * 4);
            accB = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.x, a2b[4 * q], accB, 0, 0, 0);
            accB = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.y, a2b[4 * q + 1], accB, 0, 0, 0);
            accB = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.z, a2b[4 * q + 2], accB, 0, 0, 0);
            accB = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.w, a2b[4 * q + 3], accB, 0, 0, 0);
        }
        }
        SCHED_PHASE();
    }
    {   // drift net, first layer: recomputed only now
        f32x16 hd;
        layer1_tile<F16, true>(sm, ww, ust, h, z, hd);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float ad0 = accB[4 * q] * dact(hd[4 * q]);
            float ad1 = accB[4 * q + 1] * dact(hd[4 * q + 1]);
            float ad2 = accB[4 * q + 2] * dact(hd[4 * q + 2]);
            float ad3 = accB[4 * q + 3] * dact(hd[4 * q + 3]);
#pragma unroll
            for (int k0 = 0; k0 < NN; k0 += 3) {
                float4 wz[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) wz[k] = *reinterpret_cast<const float4*>(sm.W1zT + (k0 + k) * 2 * HID + 8 * q + 4 * h);
                SCHED_PHASE();
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    Pz[k0 + k] = FMA(wz[k].x, ad0, Pz[k0 + k]); Pz[k0 + k] = FMA(wz[k].y, ad1, Pz[k0 + k]); Pz[k0 + k] = FMA(wz[k].z, ad2, Pz[k0 + k]); Pz[k0 + k] = FMA(wz[k].w, ad3, Pz[k0 + k]);
                }
            }
#pragma unroll
            for (int j0 = 0; j0 < M; j0 += 2) {
                float4 wu[2];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) wu[jj] = *reinterpret_cast<const float4*>(sm.W1uT + (j0 + jj) * HID + 8 * q + 4 * h);
                SCHED_PHASE();
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    Pu[j0 + jj] = FMA(wu[jj].x, ad0, Pu[j0 + jj]); Pu[j0 + jj] = FMA(wu[jj].y, ad1, Pu[j0 + jj]); Pu[j0 + jj] = FMA(wu[jj].z, ad2, Pu[j0 + jj]); Pu[j0 + jj] = FMA(wu[jj].w, ad3, Pu[j0 + jj]);
                }
            }
            SCHED_PHASE();
        }
    }
}

template <int M, int F16 = 0>
DI void vjp_mlp_tiles(const KArgs& a, const Smem& sm, int h, int lane, const StepAux& A, const VjpTmp& T, float* zb, float* gq) {
    if constexpr (FAST && F16 != 0) {       // SPEC.md §10e: scaled adjoints, every contraction of the sweep on the matrix pipe from binary16 limbs
        const AdjScale S = adj_scale(a, T.ebraw, T.ob);
        float ob[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) ob[i] = T.ob[i] * S.s2;
        const ObLimbs O = ob_limbs(ob);
        f32x16 accB, hn = A.h1n;
        float rows[AdjRegs<M>::N];
        adj_mp_layer2<F16>(sm, lane, [&](int q) { return make_float4(A.h2[4 * q], A.h2[4 * q + 1], A.h2[4 * q + 2], A.h2[4 * q + 3]); }, O, accB);
        adj_mp_drift<AdjRegs<M>::N, F16>(sm, lane, A.h1d, accB, rows);
        adj_mp_density<F16>(sm, h, lane, hn, T.ebraw * S.s2, rows);
        adj_rows_tile<M>(rows, S.inv, zb, gq);
        SCHED_PHASE();
        return;
    }
    float Pz[NN], Pu[M];
    vjp_mlp_partials<M, F16>(sm, h, lane, A, T.ebraw, T.ob, Pz, Pu);
#pragma unroll
    for (int k = 0; k < NN; ++k) zb[k] = xor32_sum(Pz[k]);
#pragma unroll
    for (int jj = 0; jj < M; ++jj) gq[jj] = xor32_sum(Pu[jj]);
    SCHED_PHASE();
}

// ---- tail: adjoint of the state (per particle) ----
DI void vjp_tail_v(const float dt, const float* x, const StepAux& A, const float* L, const VjpTmp& T, const float* zb, float* lam) {
    const float* Rm = A.Rm;
    const float qw = x[6], qx = x[7], qy = x[8], qz = x[9];
    const float* v = x + 3;
    const float* om = x + 10;
    const float* qtb = T.qtb; const float* Fwb = T.Fwb;
    float dqb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) dqb[i] = qtb[i] * dt;
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
DI void vjp_tail(const Smem& sm, int t, const float* x, const StepAux& A, const float* L, const VjpTmp& T, const float* zb, float* lam) {
    vjp_tail_v(sm.dt[t], x, A, L, T, zb, lam);
}

template <int M, int F16 = 0>
DI void step_vjp(const KArgs& a, const Smem& sm, const WaveW& ww, int t, int h, int lane, const float* x, const float* xi, const StepAux& A,
                 const float* L, float etabar_cost, float* lam, float* gq) {
    VjpTmp T;
    vjp_head<M>(a, sm, t, x, xi, A, L, etabar_cost, T, gq);
    SCHED_PHASE();
    float zb[NN];
    vjp_mlp_tiles<M, F16>(a, sm, h, lane, A, T, zb, gq);
    vjp_tail(sm, t, x, A, L, T, zb, lam);
}

