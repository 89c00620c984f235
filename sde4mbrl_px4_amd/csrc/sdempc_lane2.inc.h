// sdempc_lane2.inc.h — particle sweeps of the cooperative latency layouts (one particle per wave), written for instruction count
// Fragment of sdempc_kernels.hip: included inside namespace sdempc::{exact|fastm} (it is compiled twice, see there); not a
// stand-alone header.
// ================================================================================================
// A lone wave per SIMD pays about five cycles of issue time for EVERY instruction it executes — vector, scalar, wait, move or no-op
// alike (tools/valu_probe.hip, tools/salu_probe.hip; a C2 solve in the speculative layout is 1,136 instructions per forward + adjoint
// step at 5.36 cycles each). The arithmetic of a step is SPEC.md's, operation by operation, so what can shrink is everything around it.
// This file is the single-particle step of sdempc_lane.inc.h (same device functions for the MLP chains, the rigid body, the cost and
// the vector-Jacobian product, hence the same bits) with the bookkeeping rebuilt:
//   * one LDS record per step (`rec`, 64 floats): the control-dependent layer-1 terms c_t[32] and rotor thrust / torques of the control
//     sequence being evaluated (written by the prepass), and behind them what never changes during a solve — dt, stage weight, noise
//     amplitudes, the reference state of step t + 1 — so that a step reads everything through ONE address register and immediate offsets;
//   * the particle's noise rows live in LDS for the whole solve (they were six strided global loads per step, double-buffered through
//     twelve register moves);
//   * the forward sweep of a gradient evaluation leaves in its checkpoint row what the adjoint would otherwise recompute from x_{t+1}
//     and x_t: the stage-cost gradient, the next attitude, the rotation matrix — the adjoint step then depends on nothing but its own row,
//     which lets the loop be unrolled by two over two register sets (rows are requested one step ahead; no copies);
//   * the weights of a sweep are read from the LDS images at its start and die at its end (the forward and the adjoint chains use
//     disjoint halves of what LaneW keeps alive for the whole kernel: 125 registers, i.e. AGPR round trips inside the loops);
//   * the adjoint's per-step outputs leave from the lanes that hold them (one store instead of eight, no read-lane round trip).
// ================================================================================================
// record: c_t[32] | -0 -0 Tz tau[3] | dt w_t | sdt[6] 2 pad | xref_{t+1}[13] 3 pad.   Elements 32 .. 37 are what lane c < 6 adds to its scaled MLP
// output: F_b = (sF0 o0, sF1 o1, fma(sF2, o2, Tz)), tau_b = fma(sT_i, o_{3+i}, tau_i) — fma(s, o, -0) is s * o bit for bit
constexpr int RC_Z = 32, RC_T = 34, RC_DT = 38, RC_SDT = 40, RC_XR = 48;
// checkpoint row of the cooperative layouts, floats per (particle, step): h1[64] h2[64] | x_t[13] eta F_b[3] 1/|q~| q_{t+1}[4] R(q_t)[9] gx[13]
// (COOP_ROW = 128 + 44, sdempc_kernels.hip)
constexpr int CK_H1 = 0, CK_H2 = 64, CK_X = 128, CK_NU = 11;      // CK_NU float4 of uniform values behind the two activation rows
static_assert(COOP_ROW == CK_X + 4 * CK_NU, "checkpoint row");


// ---- stores by a subset of lanes WITHOUT a divergent region in the compiler's view ----
// `if (lane == 0) { stores }` makes the compiler build an EXEC-masked region; under the register pressure of these sweeps it then splits live
// ranges inside it, and a vector register saved to an AGPR under the partial mask and restored under the full one comes back with garbage in
// the lanes that were off (seen: the per-lane output pointers of the adjoint sweep, odd horizons only — a memory fault). Here EXEC is narrowed and
// restored inside ONE asm block, so the compiler sees straight-line code under the full mask (and two scalar instructions instead of three).
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define SDEMPC_STR2(x) #x
#define SDEMPC_STR(x) SDEMPC_STR2(x)
// (the lane mask is a compile-time 32-bit literal of the instruction, not a pair of scalar registers held across the sweep)
template <unsigned MASK>
DI void store_sc1_masked(float* p, float v) {      // agent-scope (write-through) store, as __hip_atomic_store(relaxed, agent) emits it
    unsigned long long sv;
    asm volatile("s_and_saveexec_b64 %0, %1\n global_store_dword %2, %3, off sc1\n s_mov_b64 exec, %0"
                 : "=&s"(sv) : "i"(MASK), "v"(p), "v"(v) : "memory", "scc");      // (s_and_saveexec writes SCC)
}
// the same store of a TAGGED output word {value, tag} (64 bits, one aligned store: the datum is its own hand-off flag — sdempc_spec.inc.h, "streamed hand-off")
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <unsigned MASK>
DI void store_sc1_masked_tagged(float* p, float v, unsigned tag) {
    unsigned long long sv;
    const u32x2 w = {__float_as_uint(v), tag};
    asm volatile("s_and_saveexec_b64 %0, %1\n global_store_dwordx2 %2, %3, off sc1\n s_mov_b64 exec, %0"
                 : "=&s"(sv) : "i"(MASK), "v"(p), "v"(w) : "memory", "scc");
}
// the CK_NU float4 of uniform values of a checkpoint row, by lane 0
DI void store_row_uniform(float* row, const f32x4 (&u)[11]) {
    unsigned long long sv;
    const unsigned long long mask = 1ull;
    asm volatile("s_and_saveexec_b64 %0, %1\n"
                 "global_store_dwordx4 %2, %3, off offset:512\n global_store_dwordx4 %2, %4, off offset:528\n global_store_dwordx4 %2, %5, off offset:544\n"
                 "global_store_dwordx4 %2, %6, off offset:560\n global_store_dwordx4 %2, %7, off offset:576\n global_store_dwordx4 %2, %8, off offset:592\n"
                 "global_store_dwordx4 %2, %9, off offset:608\n global_store_dwordx4 %2, %10, off offset:624\n global_store_dwordx4 %2, %11, off offset:640\n"
                 "global_store_dwordx4 %2, %12, off offset:656\n global_store_dwordx4 %2, %13, off offset:672\n"
                 "s_mov_b64 exec, %0"
                 : "=&s"(sv) : "s"(mask), "v"(row), "v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(u[3]), "v"(u[4]), "v"(u[5]), "v"(u[6]), "v"(u[7]), "v"(u[8]), "v"(u[9]), "v"(u[10])
                 : "memory", "scc");
}
static_assert(CK_X * 4 == 512 && CK_NU == 11, "store_row_uniform's offsets");

struct Lane2Lds {
    const float* rec;  // [H][REC]
    const float* nz;   // this wave's noise rows [H][NZL]
};
DI Lane2Lds lane2_lds(const KArgs& a, const Smem& sm, int wave) { Lane2Lds L; L.rec = sm.rec; L.nz = sm.nzl + wave * a.H * NZL; return L; }

// once per kernel, after load_common: the static tail of every step record and the noise rows of this workgroup's four particles
template <class Team>
DI void lane2_stage(const KArgs& a, const Smem& sm, int b, int wgi, int tid) {
    const int H = a.H;
    for (int i = tid; i < H * 32; i += Team::NT) {
        const int t = i >> 5, j = (i & 31) + 32;          // record element 32 .. 63 (34 .. 37 belong to the prepass)
        float v = 0.0f;
        if (j < RC_T) v = -0.0f;
        else if (j == RC_DT) v = sm.dt[t];
        else if (j == RC_DT + 1) v = sm.disc[t];
        else if (j >= RC_SDT && j < RC_SDT + NN) v = sm.sdt[t * NN + (j - RC_SDT)];
        else if (j >= RC_XR && j < RC_XR + NX) v = sm.xref[(t + 1) * NX + (j - RC_XR)];
        if (j < RC_T || j >= RC_DT) sm.rec[t * REC + j] = v;
    }
    for (int i = tid; i < 4 * H * NZL; i += Team::NT) {
        const int w = i / (H * NZL), r = i - w * (H * NZL), t = r >> 3, c = r & 7, p = wgi * 4 + w;
        float v = 0.0f;
        if (c < NN && p < a.P) v = a.noise[(((size_t)(b * a.G + (p >> 5)) * H + t) * NN + c) * 32 + (p & 31)];
        sm.nzl[i] = v;
    }
}

// SPEC.md §5.1 into the step records (block_prepass writes the same values into the [H][36] table of the other layouts)
template <class Team>
DI void lane2_prepass(const KArgs& a, const Smem& sm, const float* u, int tid) {
    const int H = a.H, m = a.m;
    for (int e = tid; e < H * HID; e += Team::NT) {
        int t = e >> 5, r = e & 31;
        float c = sm.b1d[r];
        for (int j = 0; j < m; ++j) c = FMA(fwd_w1(sm.W1uT[j * HID + r]), u[t * m + j], c);
        sm.rec[t * REC + r] = c;
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
        float* r = sm.rec + t * REC + RC_T;
        r[0] = Tz; r[1] = t0; r[2] = t1; r[3] = t2;
    }
}

// ---- weights of one sweep, from the LDS images load_weights staged (f32 mode: A2 / A2T hold W2 / W2^T in MFMA lane order) ----
struct L2FwdW {
    float w1[NN], c1n, b2k, w2row[HID], wo[16], bo, so;      // bo / so: bias and residual scale (sF, sT) of this lane's output chain
    int obase;
};
struct L2AdjW {
    float w2col[HID], w3col[6], w3nk, wz[32];
    int zbase;
    bool is_u;
};
DI float4 lds4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// row k of the matrix an MFMA A-operand image holds: image[(q*64 + l)*4 + c] = Mat[l & 31][rowmap(4q + c, l >> 5)]
DI void image_row(const float* img, int k, float* row) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int hs = 0; hs < 2; ++hs) {
            const float4 v = lds4(img + (q * 64 + k + 32 * hs) * 4);
            row[8 * q + 4 * hs] = v.x; row[8 * q + 4 * hs + 1] = v.y; row[8 * q + 4 * hs + 2] = v.z; row[8 * q + 4 * hs + 3] = v.w;
        }
    }
}
DI void lane2_load_fwd(const KArgs& a, const Smem& sm, L2FwdW& W, int lane) {
    const int k = lane & 31, hh = lane >> 5, row = 32 * hh + k;
#pragma unroll
    for (int j = 0; j < NN; ++j) W.w1[j] = fwd_w1(sm.W1zT[j * 2 * HID + row]);
    W.c1n = sm.b1n[k];
    W.b2k = sm.b2[k];
    image_row(sm.A2, k, W.w2row);
    const int c = lane & 7, hs = (lane >> 3) & 1;
    const bool row0 = lane < 16;
    W.obase = (row0 && c == 6) ? 32 + 4 * hs : 4 * hs;
    const float* src = c < 6 ? sm.W3 + c * HID : sm.w3n;              // (lanes that hold no chain read somewhere harmless and are zeroed below)
    const bool live = row0 && c < 7;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 v = lds4(src + 8 * q + 4 * hs);
        W.wo[4 * q] = live ? v.x : 0.0f; W.wo[4 * q + 1] = live ? v.y : 0.0f; W.wo[4 * q + 2] = live ? v.z : 0.0f; W.wo[4 * q + 3] = live ? v.w : 0.0f;
    }
    const float bv = a.wts[lane < 6 ? OFF_B3 + lane : OFF_B3N];       // (every lane loads: no divergent region)
    W.bo = lane <= 6 ? bv : 0.0f;
    const float sv = a.wts[OFF_SF + (lane < 6 ? lane : 0)];           // sF[3], sT[3] (SPEC.md §2)
    W.so = lane < 6 ? sv : 0.0f;
}
DI void lane2_load_adj(const KArgs& a, const Smem& sm, L2AdjW& W, int lane) {
    const int k = lane & 31;
    image_row(sm.A2T, k, W.w2col);
#pragma unroll
    for (int i = 0; i < 6; ++i) W.w3col[i] = vjp_w3(sm.W3[i * HID + k]);
    W.w3nk = vjp_w3(sm.w3n[k]);
    const int c = lane & 7, hs = (lane >> 3) & 1;
    const bool row0 = lane < 16, row1 = lane >= 16 && lane < 32;
    W.zbase = 4 * hs;
    W.is_u = row1;
    const bool zlive = row0 && c < 6, ulive = row1 && c < a.m;
    const int cz = c < 6 ? c : 0, cu = c < a.m ? c : 0;
    const float* srcd = sm.W1zT + cz * 2 * HID + HID;                  // density rows of W1z, column c
    const float* srcf = row1 ? sm.W1uT + cu * HID : sm.W1zT + cz * 2 * HID;
    const bool flive = zlive || ulive;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 vd = lds4(srcd + 8 * q + 4 * hs), vf = lds4(srcf + 8 * q + 4 * hs);
        W.wz[4 * q] = zlive ? vd.x : 0.0f; W.wz[4 * q + 1] = zlive ? vd.y : 0.0f; W.wz[4 * q + 2] = zlive ? vd.z : 0.0f; W.wz[4 * q + 3] = zlive ? vd.w : 0.0f;
        W.wz[16 + 4 * q] = flive ? vf.x : 0.0f; W.wz[16 + 4 * q + 1] = flive ? vf.y : 0.0f; W.wz[16 + 4 * q + 2] = flive ? vf.z : 0.0f; W.wz[16 + 4 * q + 3] = flive ? vf.w : 0.0f;
    }
}

// forward MLPs of one step (lane_fwd_mlp with the sweep's own weights; ck: c_t[k] of this lane's drift unit; zt: what lane c < 6 adds to its
// scaled output, RC_Z + c of the record). ft[0..2] = F_b, ft[3..5] = tau_b
DI void lane2_fwd_mlp(const L2FwdW& W, const LaneSel& ls, int hh, float ck, float zt, const float* z, float& h1, float& h2, float* ft, float& eta) {
    float a1 = hh ? W.c1n : ck;
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
    float gsrc[16];
    gather16<0>(W.obase << 2, Mreg, gsrc);
    float P = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) P = FMA(W.wo[r], gsrc[r], P);
    const float Pc = (P + dpp_f<0x128>(P)) + W.bo;   // row_ror:8 -> lane c: (P_0 + P_1) + bias of its chain
    const float S = FMA(W.so, Pc, zt);               // lanes 0 .. 2: F_b, lanes 3 .. 5: tau_b (SPEC.md §5.2)
#pragma unroll
    for (int i = 0; i < 6; ++i) ft[i] = readlane_f(S, i);
    eta = sigmoid_spec(readlane_f(Pc, 6));
}

// adjoint of the MLPs: zb[6] to every lane; Pc carries the per-particle outputs gq[jj] in lanes 16 + jj (lane_vjp_mlp read them back to every lane)
DI float lane2_vjp_mlp(const L2AdjW& W, int hh, float h1, float h2, const VjpTmp& T, float* zb) {
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
    float gsrc[16];
    float Pz = 0.0f;
    gather16<128>(W.zbase << 2, Abar, gsrc);
#pragma unroll
    for (int p = 0; p < 16; ++p) Pz = FMA(W.wz[p], gsrc[p], Pz);     // density units
    Pz = W.is_u ? 0.0f : Pz;             // the gu-bar chains start at the drift units
    gather16<0>(W.zbase << 2, Abar, gsrc);
#pragma unroll
    for (int p = 0; p < 16; ++p) Pz = FMA(W.wz[16 + p], gsrc[p], Pz);
    const float Pc = Pz + dpp_f<0x128>(Pz);
#pragma unroll
    for (int kk = 0; kk < NN; ++kk) zb[kk] = readlane_f(Pc, kk);
    return Pc;
}

// the constants of step t, from its record
struct StepK { float dt, wt, sdt[NN], xr[NX], xi[NN]; };
DI float2 lds2(const float* p) { return *reinterpret_cast<const float2*>(p); }
DI void lane2_step_consts(const float* rp, const float* np, StepK& K) {
    const float2 d4 = lds2(rp + RC_DT);
    const float4 s4 = lds4(rp + RC_SDT), s5 = lds4(rp + RC_SDT + 4);
    const float4 x0 = lds4(rp + RC_XR), x1 = lds4(rp + RC_XR + 4), x2 = lds4(rp + RC_XR + 8), x3 = lds4(rp + RC_XR + 12);
    const float4 n0 = lds4(np), n1 = lds4(np + 4);
    K.dt = d4.x; K.wt = d4.y;
    K.sdt[0] = s4.x; K.sdt[1] = s4.y; K.sdt[2] = s4.z; K.sdt[3] = s4.w; K.sdt[4] = s5.x; K.sdt[5] = s5.y;
    K.xr[0] = x0.x; K.xr[1] = x0.y; K.xr[2] = x0.z; K.xr[3] = x0.w; K.xr[4] = x1.x; K.xr[5] = x1.y; K.xr[6] = x1.z; K.xr[7] = x1.w;
    K.xr[8] = x2.x; K.xr[9] = x2.y; K.xr[10] = x2.z; K.xr[11] = x2.w; K.xr[12] = x3.x;
    K.xi[0] = n0.x; K.xi[1] = n0.y; K.xi[2] = n0.z; K.xi[3] = n0.w; K.xi[4] = n1.x; K.xi[5] = n1.y;
}

// Where one particle's streams live in the cooperative layouts
struct Lane2IO {
    const float* x0;   // [13]
    float* ck;         // checkpoint rows of this particle: row t at ck + t * COOP_ROW
    float* out;        // per-particle outputs: quantity q at out[q * os] (q: t*12 + k adjoint sums, t*13 + i states, PS - 1 cost)
    int os;
    unsigned tag;      // TAGGED sweeps (speculative kernel): every output is a 64-bit word {value, tag}, os counts floats (two per word)
};
template <bool TAGGED, unsigned MASK>
DI void lane2_out(const Lane2IO& io, float* p, float v) {
    if constexpr (TAGGED) store_sc1_masked_tagged<MASK>(p, v, io.tag);
    else store_sc1_masked<MASK>(p, v);
}

// one particle: rollout and cost; MEAN: the states go to io.out (the final rollout of a solve)
template <bool MEAN, bool TAGGED = false>
DI void lane2_rollout(const KArgs& a, const Smem& sm, const Lane2Lds& L, const Lane2IO& io, int lane) {
    const int H = a.H, PS = part_stride(H), hh = lane >> 5;
    const LaneSel ls = lane_sel(lane);
    L2FwdW W;
    lane2_load_fwd(a, sm, W, lane);
    float x[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = io.x0[i];
    if (MEAN) {
#pragma unroll
        for (int i = 0; i < NX; ++i) lane2_out<TAGGED, 1u>(io, io.out + (size_t)i * io.os, x[i]);
    }
    float J = 0.0f;
    StepAux A;
    const float* rp = L.rec;
    const float* rk = L.rec + (lane & 31);
    const float* rz = L.rec + RC_Z + (lane < 6 ? lane : 0);
    const float* np = L.nz;
    for (int t = 0; t < H; ++t) {
        const float ck = *rk, zt = *rz;
        StepK K;
        lane2_step_consts(rp, np, K);
        float z[NN], h1, h2, ft[6], eta, xn[NX];
        fwd_head(x, A.Rm, z);
        lane2_fwd_mlp(W, ls, hh, ck, zt, z, h1, h2, ft, eta);
        A.Fb[0] = ft[0]; A.Fb[1] = ft[1]; A.Fb[2] = ft[2];
        fwd_tail_ft(a, K.dt, ft + 3, K.sdt, x, K.xi, A.Rm, eta, xn, A);
        float l = stage_cost<false, false>(a, xn, K.xr, nullptr);
        l = FMA(a.C.res_mult * A.eta, A.eta, l);
        J = FMA(K.wt, l, J);
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = xn[i];
        if (MEAN) {
#pragma unroll
            for (int i = 0; i < NX; ++i) lane2_out<TAGGED, 1u>(io, io.out + (size_t)((t + 1) * NX + i) * io.os, x[i]);
        }
        rp += REC; rk += REC; rz += REC; np += NZL;
    }
    lane2_out<TAGGED, 1u>(io, io.out + (size_t)(PS - 1) * io.os, J);
}

// what the adjoint step t reads: its checkpoint row
struct AdjRow { float h1, h2; float4 u[CK_NU]; };
DI void lane2_request_row(const float* row, int lane, AdjRow& R) {
    R.h1 = row[CK_H1 + lane];
    R.h2 = row[CK_H2 + lane];
#pragma unroll
    for (int i = 0; i < CK_NU; ++i) R.u[i] = *reinterpret_cast<const float4*>(row + CK_X + 4 * i);
}

// one particle: cost, forward sweep with checkpoint, adjoint sweep; per-step adjoint outputs gq[0..M+3] -> io.out
template <int M, bool TAGGED = false>
DI void lane2_grad(const KArgs& a, const Smem& sm, const Lane2Lds& L, const Lane2IO& io, int lane) {
    const int H = a.H, PS = part_stride(H), hh = lane >> 5;
    constexpr int nq = M + 4;
    {   // ---- forward sweep: row t <- h1, h2 | x_t, eta, F_b | 1/|q~|, q_{t+1}, R(q_t) | cost gradient at x_{t+1} ----
        // (two steps per trip over two state registers sets: x_t's registers are still being stored when x_{t+1} is formed)
        const LaneSel ls = lane_sel(lane);
        L2FwdW W;
        lane2_load_fwd(a, sm, W, lane);
        float xa[NX], xb[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) xa[i] = io.x0[i];
        float J = 0.0f;
        const float* rp = L.rec;
        const float* rk = L.rec + (lane & 31);
        const float* rz = L.rec + RC_Z + (lane < 6 ? lane : 0);
        const float* np = L.nz;
        float* row = io.ck;
        auto fstep = [&](const float (&x)[NX], float (&xn)[NX]) {
            const float ck = *rk, zt = *rz;
            StepK K;
            lane2_step_consts(rp, np, K);
            StepAux A;
            float z[NN], h1, h2, ft[6], eta, gx[NX];
            fwd_head(x, A.Rm, z);
            lane2_fwd_mlp(W, ls, hh, ck, zt, z, h1, h2, ft, eta);
            A.Fb[0] = ft[0]; A.Fb[1] = ft[1]; A.Fb[2] = ft[2];
            fwd_tail_ft(a, K.dt, ft + 3, K.sdt, x, K.xi, A.Rm, eta, xn, A);
            float l = stage_cost<true, false>(a, xn, K.xr, gx);
            l = FMA(a.C.res_mult * A.eta, A.eta, l);
            J = FMA(K.wt, l, J);
            row[CK_H1 + lane] = h1;
            row[CK_H2 + lane] = h2;
            const f32x4 u[11] = {{x[0], x[1], x[2], x[3]}, {x[4], x[5], x[6], x[7]}, {x[8], x[9], x[10], x[11]}, {x[12], A.eta, A.Fb[0], A.Fb[1]},
                                 {A.Fb[2], A.rn, A.qn[0], A.qn[1]}, {A.qn[2], A.qn[3], A.Rm[0], A.Rm[1]}, {A.Rm[2], A.Rm[3], A.Rm[4], A.Rm[5]},
                                 {A.Rm[6], A.Rm[7], A.Rm[8], gx[0]}, {gx[1], gx[2], gx[3], gx[4]}, {gx[5], gx[6], gx[7], gx[8]}, {gx[9], gx[10], gx[11], gx[12]}};
            store_row_uniform(row, u);
            rp += REC; rk += REC; rz += REC; np += NZL; row += COOP_ROW;
        };
        int t = 0;
        for (; t + 1 < H; t += 2) { fstep(xa, xb); fstep(xb, xa); }
        if (t < H) fstep(xa, xb);
        lane2_out<TAGGED, 1u>(io, io.out + (size_t)(PS - 1) * io.os, J);
    }
    // ---- adjoint sweep: two steps per trip over two register sets; the row of step t - 1 is requested while step t is processed ----
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    L2AdjW W;
    lane2_load_adj(a, sm, W, lane);
    float lam[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) lam[i] = 0.0f;
    // lanes 16 .. 16 + nq - 1 store the step's outputs: gq[jj] sits in lane 16 + jj of the chain register, the four uniform ones are picked in
    const int oq = lane - 16;
    const bool olane = oq >= 0 && oq < nq;
    float* op = io.out + (size_t)(olane ? oq : 0) * io.os;      // + t * 12 * os per step
    const size_t ostep = (size_t)12 * io.os;
    constexpr unsigned omask = ((1u << nq) - 1u) << 16;
    auto step = [&](int t, const AdjRow& R) -> float {
        const float* rp = L.rec + t * REC;
        const float* np = L.nz + t * NZL;
        const float2 d4 = lds2(rp + RC_DT);
        const float4 s4 = lds4(rp + RC_SDT), s5 = lds4(rp + RC_SDT + 4), n0 = lds4(np), n1 = lds4(np + 4);
        const float dt = d4.x, dsc = d4.y;
        const float sdt[NN] = {s4.x, s4.y, s4.z, s4.w, s5.x, s5.y}, xi[NN] = {n0.x, n0.y, n0.z, n0.w, n1.x, n1.y};
        const float4* u = R.u;
        const float xt[NX] = {u[0].x, u[0].y, u[0].z, u[0].w, u[1].x, u[1].y, u[1].z, u[1].w, u[2].x, u[2].y, u[2].z, u[2].w, u[3].x};
        StepAux A;
        A.eta = u[3].y; A.Fb[0] = u[3].z; A.Fb[1] = u[3].w; A.Fb[2] = u[4].x; A.rn = u[4].y;
        A.qn[0] = u[4].z; A.qn[1] = u[4].w; A.qn[2] = u[5].x; A.qn[3] = u[5].y;
        A.Rm[0] = u[5].z; A.Rm[1] = u[5].w; A.Rm[2] = u[6].x; A.Rm[3] = u[6].y; A.Rm[4] = u[6].z; A.Rm[5] = u[6].w; A.Rm[6] = u[7].x; A.Rm[7] = u[7].y; A.Rm[8] = u[7].z;
        const float gx[NX] = {u[7].w, u[8].x, u[8].y, u[8].z, u[8].w, u[9].x, u[9].y, u[9].z, u[9].w, u[10].x, u[10].y, u[10].z, u[10].w};
#pragma unroll
        for (int i = 0; i < NX; ++i) lam[i] = FMA(dsc, gx[i], lam[i]);
#pragma unroll
        for (int i = 0; i < 3; ++i) A.Jom[i] = a.M.J[i] * xt[10 + i];
        const float ebc = dsc * ((2.0f * a.C.res_mult) * A.eta);
        float lamn[NX], gq[12], zb[NN];
        VjpTmp T;
        vjp_head_v<M>(a, dt, sdt, xt, xi, A, lam, ebc, T, gq);
        const float Pc = lane2_vjp_mlp(W, hh, R.h1, R.h2, T, zb);
        vjp_tail_v(dt, xt, A, lam, T, zb, lamn);
#pragma unroll
        for (int i = 0; i < NX; ++i) lam[i] = lamn[i];
        // outputs: lane 16 + jj holds gq[jj] (jj < M); lanes 16 + M .. 16 + M + 3 take the thrust / rotor-torque adjoints
        float ov = Pc;
        ov = oq == M ? gq[M] : ov; ov = oq == M + 1 ? gq[M + 1] : ov; ov = oq == M + 2 ? gq[M + 2] : ov; ov = oq == M + 3 ? gq[M + 3] : ov;
        return ov;
    };
    // A step's outputs are stored BEHIND the row request that follows the step. gfx9 has ONE counter for loads and stores, in issue order: the wait for a
    // checkpoint row also waits for every store issued before that row's request, and the acknowledgement of an agent-scope (write-through, cross-XCD) store
    // takes longer than a step (~ 1 us). With the store in front of the request every step waited for the store of the step before: 8 us of a 50-step sweep
    // (C2 parallel phase 101.3 -> 93.4 us, measured with tools/spec_clock.py; single solve 22.6 -> 20.3 ms). Now a store has two steps to complete
    // (one step more — behind the second request after its step — evens out the last 0.8 us between the gradient groups but costs 0.6 % in moves: not taken).
    float* ow = op + (size_t)(H - 1) * ostep;      // the stores leave in step order H - 1, H - 2, ...: one running pointer
    auto out = [&](float ov) { lane2_out<TAGGED, omask>(io, ow, ov); ow -= ostep; };
    AdjRow RA, RB;
    const float* rowH = io.ck + (size_t)(H - 1) * COOP_ROW;
    lane2_request_row(rowH, lane, RA);
    int t = H - 1;
    float ob = 0.0f;
    for (; t >= 1; t -= 2) {
        lane2_request_row(io.ck + (size_t)(t - 1) * COOP_ROW, lane, RB);
        if (t != H - 1) out(ob);
        const float oa = step(t, RA);
        lane2_request_row(io.ck + (size_t)(t >= 2 ? t - 2 : 0) * COOP_ROW, lane, RA);      // (t == 1: row 0 once more, never used)
        out(oa);
        ob = step(t - 1, RB);
    }
    if (H >= 2) out(ob);
    if (t == 0) out(step(0, RA));
}
