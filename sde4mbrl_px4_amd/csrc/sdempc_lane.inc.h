// sdempc_lane.inc.h — single-particle layout (one hidden unit per lane) and the P == 1 team
// Fragment of sdempc_kernels.hip: included inside namespace sdempc::{exact|fastm} (it is compiled twice, see there); not a
// stand-alone header.
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
DI float uni_f(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }   // wave-uniform value -> SGPR
DI float readlane_f(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
// base_bytes: byte address of the chain's first source lane, kept below 256 so that the compiler can fold the constant part into the
// instruction's offset field (one VALU address add per gather otherwise)
DI float bperm_f(int base_bytes, int lane_off, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((base_bytes & 252) + 4 * lane_off, __builtin_bit_cast(int, v)));
}

// 16 gathers of `src` from lanes (base_bytes/4 + koff(r) + EXTRA) mod 64, r = 0..15, in ONE asm block: the constant part of every address rides
// in the instruction's offset field (the builtin costs a VALU address op per gather) and a single wait follows (for a lone wave a
// s_waitcnt costs as much issue time as a vector instruction, tools/salu_probe.hip). koff(r)*4 = 0,4,8,12,32,36,40,44,64,...,108.
template <int EXTRA_BYTES>
DI void gather16(int base_bytes, float src, float* g) {
    static_assert(EXTRA_BYTES == 0 || EXTRA_BYTES == 128, "lane offset 0 or 32");
    if constexpr (EXTRA_BYTES == 0) {
        asm volatile(
            "ds_bpermute_b32 %0, %16, %17\n ds_bpermute_b32 %1, %16, %17 offset:4\n ds_bpermute_b32 %2, %16, %17 offset:8\n ds_bpermute_b32 %3, %16, %17 offset:12\n"
            "ds_bpermute_b32 %4, %16, %17 offset:32\n ds_bpermute_b32 %5, %16, %17 offset:36\n ds_bpermute_b32 %6, %16, %17 offset:40\n ds_bpermute_b32 %7, %16, %17 offset:44\n"
            "ds_bpermute_b32 %8, %16, %17 offset:64\n ds_bpermute_b32 %9, %16, %17 offset:68\n ds_bpermute_b32 %10, %16, %17 offset:72\n ds_bpermute_b32 %11, %16, %17 offset:76\n"
            "ds_bpermute_b32 %12, %16, %17 offset:96\n ds_bpermute_b32 %13, %16, %17 offset:100\n ds_bpermute_b32 %14, %16, %17 offset:104\n ds_bpermute_b32 %15, %16, %17 offset:108\n"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3]), "=&v"(g[4]), "=&v"(g[5]), "=&v"(g[6]), "=&v"(g[7]),
              "=&v"(g[8]), "=&v"(g[9]), "=&v"(g[10]), "=&v"(g[11]), "=&v"(g[12]), "=&v"(g[13]), "=&v"(g[14]), "=&v"(g[15])
            : "v"(base_bytes), "v"(src));
    } else {
        asm volatile(
            "ds_bpermute_b32 %0, %16, %17 offset:128\n ds_bpermute_b32 %1, %16, %17 offset:132\n ds_bpermute_b32 %2, %16, %17 offset:136\n ds_bpermute_b32 %3, %16, %17 offset:140\n"
            "ds_bpermute_b32 %4, %16, %17 offset:160\n ds_bpermute_b32 %5, %16, %17 offset:164\n ds_bpermute_b32 %6, %16, %17 offset:168\n ds_bpermute_b32 %7, %16, %17 offset:172\n"
            "ds_bpermute_b32 %8, %16, %17 offset:192\n ds_bpermute_b32 %9, %16, %17 offset:196\n ds_bpermute_b32 %10, %16, %17 offset:200\n ds_bpermute_b32 %11, %16, %17 offset:204\n"
            "ds_bpermute_b32 %12, %16, %17 offset:224\n ds_bpermute_b32 %13, %16, %17 offset:228\n ds_bpermute_b32 %14, %16, %17 offset:232\n ds_bpermute_b32 %15, %16, %17 offset:236\n"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3]), "=&v"(g[4]), "=&v"(g[5]), "=&v"(g[6]), "=&v"(g[7]),
              "=&v"(g[8]), "=&v"(g[9]), "=&v"(g[10]), "=&v"(g[11]), "=&v"(g[12]), "=&v"(g[13]), "=&v"(g[14]), "=&v"(g[15])
            : "v"(base_bytes), "v"(src));
    }
}

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
    float bo;            // bias of this lane's output chain: b3[c] (lanes c < 6), b3n (lane 6), else 0
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
    for (int i = 0; i < HID; ++i) { W.w2row[i] = w[OFF_W2 + k * HID + i]; W.w2col[i] = w[VJP_BASE + OFF_W2 + i * HID + k]; }      // (VJP_BASE: math_mode fast keeps the weights of the vector-Jacobian products in a second block, SPEC.md §10b)
#pragma unroll
    for (int i = 0; i < 6; ++i) W.w3col[i] = w[VJP_BASE + OFF_W3 + i * HID + k];
    W.w3nk = w[VJP_BASE + OFF_W3N + k];
    const int c = lane & 7, hs = (lane >> 3) & 1;
    const bool row0 = lane < 16, row1 = lane >= 16 && lane < 32;
    W.bo = lane < 6 ? a.M.b3[lane < 6 ? lane : 0] : lane == 6 ? a.M.b3n : 0.0f;
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
        if (row0 && c < 6) { zd = w[VJP_BASE + OFF_W1Z + (HID + unit) * NN + c]; zf = w[VJP_BASE + OFF_W1Z + unit * NN + c]; }
        if (row1 && c < a.m) zf = w[VJP_BASE + OFF_W1U + unit * 8 + c];
        W.wz[r] = zd; W.wz[16 + r] = zf;
    }
}

// SPEC.md §3.4 tanh4 with the four values of a group in the four lanes of a DPP quad
// (quad broadcasts without an initialised destination: every lane of a quad_perm is valid; the per-lane pick of r_q is two bit-field
// selects on lane masks — left to `?:` the compiler built EXEC-mask regions around the multiplications: ten scalar instructions per call,
// and for a lone wave every instruction, scalar ones included, costs five cycles of issue time)
template <int CTRL>
DI float dppq_f(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true)); }
DI float sel_f(float a, float b, unsigned m) {      // m all ones: a, zero: b   (v_bfi_b32)
    return __uint_as_float((__float_as_uint(a) & m) | (__float_as_uint(b) & ~m));
}
struct LaneSel { unsigned q1, q2; };                // all ones where lane & 1 / lane & 2
DI LaneSel lane_sel(int lane) { LaneSel s; s.q1 = (lane & 1) ? ~0u : 0u; s.q2 = (lane & 2) ? ~0u : 0u; return s; }
DI float lane_tanh(float av, const LaneSel& ls) {
    if constexpr (FAST) return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(av));      // SPEC.md §10b, as tanh16_hw
    const float d = 1.0f + exp2_spec(clampf(av, -9.0f, 9.0f), 2.885390043258667f);
    const float d0 = dppq_f<0x00>(d), d1 = dppq_f<0x55>(d), d2 = dppq_f<0xAA>(d), d3 = dppq_f<0xFF>(d);
    const float p2 = d0 * d1, p3 = p2 * d2, p4 = p3 * d3;
    float r = rcp_spec(p4);
    const float r3 = r * p3; r = r * d3;
    const float r2 = r * p2; r = r * d2;
    const float r1 = r * d0;
    const float r0 = r * d1;
    const float rq = sel_f(sel_f(r3, r2, ls.q1), sel_f(r1, r0, ls.q1), ls.q2);
    return FMA(-2.0f, rq, 1.0f);
}

// forward MLPs of one step; h1: drift (lanes 0..31) / density (32..63) hidden unit, h2: layer-2 unit (both halves)
DI void lane_fwd_mlp(const KArgs& a, const LaneW& W, const float* ust, int lane, const float* z, float& h1, float& h2, float* o, float& eta) {
    const int k = lane & 31, hh = lane >> 5;
    const LaneSel ls = lane_sel(lane);
    float a1 = hh ? W.c1n : ust[k];
#pragma unroll
    for (int j = 0; j < NN; ++j) a1 = FMA(W.w1[j], z[j], a1);
    h1 = lane_tanh(a1, ls);
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
    h2 = lane_tanh(a2, ls);
    const float Mreg = hh ? h1 : h2;     // lanes 0..31: layer-2 activations, lanes 32..63: density hidden units
    // all 16 gathers first, ONE wait, then the fma chain (left to itself the compiler waits before every fma: for a lone wave a
    // s_waitcnt costs as much issue time as a vector instruction, tools/salu_probe.hip)
    float gsrc[16];
    gather16<0>(W.obase << 2, Mreg, gsrc);
    float P = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) P = FMA(W.wo[r], gsrc[r], P);
    // lane c < 7: (P_0 + P_1) + bias of its chain, added in the lane (two scalar operands in one add would cost a move each)
    const float Pc = (P + dpp_f<0x128>(P)) + W.bo;   // row_ror:8 -> lane c: P_0 + P_1
#pragma unroll
    for (int i = 0; i < 6; ++i) o[i] = readlane_f(Pc, i);
    eta = sigmoid_spec(readlane_f(Pc, 6));
}

// adjoint of the MLPs: zb[6], gq[0..M-1]
template <int M>
DI void lane_vjp_mlp(const LaneW& W, int lane, float h1, float h2, const VjpTmp& T, float* zb, float* gq) {
    const int hh = lane >> 5;
    float hb = 0.0f;
#pragma unroll
    for (int i = 0; i < 6; ++i) hb = FMA(W.w3col[i], T.ob[i], hb);
    const float a2b = hb * dact(h2);
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
    const float g1 = dact(h1);
    const float ad = accB * g1;
    const float an = (W.w3nk * T.ebraw) * g1;
    const float Abar = hh ? an : ad;
    float gsrc[16];                     // gather (one block, one wait: see lane_fwd_mlp), then the chain; density units first
    float Pz = 0.0f;
    gather16<128>(W.zbase << 2, Abar, gsrc);
#pragma unroll
    for (int p = 0; p < 16; ++p) Pz = FMA(W.wz[p], gsrc[p], Pz);     // density units
    Pz = W.is_u ? 0.0f : Pz;             // the gu-bar chains start at the drift units (whatever the density part produced in those lanes is dropped)
    gather16<0>(W.zbase << 2, Abar, gsrc);
#pragma unroll
    for (int p = 0; p < 16; ++p) Pz = FMA(W.wz[16 + p], gsrc[p], Pz);
    const float Pc = Pz + dpp_f<0x128>(Pz);
#pragma unroll
    for (int kk = 0; kk < NN; ++kk) zb[kk] = readlane_f(Pc, kk);
#pragma unroll
    for (int jj = 0; jj < M; ++jj) gq[jj] = readlane_f(Pc, 16 + jj);
}

constexpr int LANE_ACT_H1 = 0, LANE_ACT_H2 = 64, LANE_ACT_SC = 128, LANE_ACT_X = 136;   // offsets inside one checkpoint row

// Where one particle's streams live (the same device functions serve the P == 1 team and the cooperative multi-workgroup path)
struct LaneIO {
    const float* x0;          // [13]
    const float* nz;          // noise: element (t, i) at nz[(t*6 + i) * 32]
    float* xs; int xs_t, xs_i;   // x_t kept for the adjoint / traj output: element (t, i) at xs[t*xs_t + i*xs_i]
    float* ck; int ck_t;      // checkpoint rows: row t at ck + t*ck_t
    float* out; int os;       // per-particle outputs: quantity q at out[q*os]  (q: t*12+k adjoint sums, t*13+i states, PS-1 cost)
    bool add0;                // P == 1: store v + 0.0f (what the SPEC.md §6.1 butterfly over 31 zero lanes leaves)
};
// Cooperative path: the handed-off values are always written and read with agent-scope (sc1) accesses — write-through stores that
// do not stay dirty in the XCD's L2, loads that bypass the CU's L1 — so the grid barrier needs no L2 write-back / invalidate (the
// per-XCD L2s are not coherent with each other; a release fence would flush every dirty line of the checkpoint stream as well).
// See coop_barrier for the ordering argument and the optional fences.
DI void out_store(const LaneIO& io, size_t q, float v) {
    if (io.add0) io.out[q] = v + 0.0f;                 // P == 1 team (os == 1)
    else __hip_atomic_store(io.out + q * io.os, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
DI float coop_load(const float* p) {
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
        float l = stage_cost<false, false>(a, xn, sm.xref + (t + 1) * NX, nullptr);
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
        float l = stage_cost<false, false>(a, xn, sm.xref + (t + 1) * NX, nullptr);
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
            stage_cost<true, false>(a, x, sm.xref + (t + 1) * NX, gx);
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

