// sdempc_duo.inc.h — tile layout with 64 particles per wave ("duo": two 32-particle groups per wave)
// Fragment of sdempc_kernels.hip: included inside namespace sdempc::{exact|fastm}; not a stand-alone header.
// ================================================================================================
// In the plain tile layout (block_rollout / block_cost_grad) a wave owns ONE 32-particle group: lane l holds particle l & 31 and half
// h = l >> 5 of the hidden units, so everything that is per particle rather than per hidden unit — rotation matrix, rigid body, Euler-
// Maruyama update, quaternion normalisation, stage cost and the head / tail of the vector-Jacobian product, ≈15 % of the forward and
// ≈22 % of the adjoint step's vector instructions — is computed twice, identically, by both lane halves.
// Here a wave owns a PAIR of groups: lanes 0..31 carry the particles of group gA = 2 gp, lanes 32..63 those of gB = 2 gp + 1. The per-
// particle work runs once on 64 distinct particles; the MLPs run as two passes in the unchanged MFMA accumulator layout (pass A on group
// A's inputs broadcast to both halves, then pass B), and v_permlane32_swap both broadcasts the six inputs of a pass and, in one swap +
// one add per value, folds the per-half partial sums of both passes so that every lane ends up with ITS particle's total:
//     swap(PA, PB) = {PA.lo, PB.lo}, {PA.hi, PB.hi};  sum: lanes < 32 -> PA.lo + PA.hi, lanes >= 32 -> PB.lo + PB.hi   ( = (P0 + P1), SPEC.md §4 )
// Every value is produced by the same operations in the same order as in the plain layout: results are bit-identical (same tests).
// Teams: two waves own an instance of up to four groups (C2: P = 128 -> 128-thread workgroups, six per CU), four waves otherwise.
// ================================================================================================

// {v.lo, v.lo} and {v.hi, v.hi}: a group's per-particle value made visible to both lane halves
DI void half_split(float v, float& a_both, float& b_both) {
    unsigned a = __builtin_bit_cast(unsigned, v), b;
    asm("v_mov_b32 %0, %1" : "=v"(b) : "v"(a));      // the swap needs two registers (given one register twice it aliases, see xor32_sum)
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    const unsigned r0 = r[0], r1 = r[1];
    a_both = __builtin_bit_cast(float, r0); b_both = __builtin_bit_cast(float, r1);
}
// per-half partial sums of pass A (pa) and pass B (pb) -> this lane's particle total: lanes < 32: pa.lo + pa.hi, lanes >= 32: pb.lo + pb.hi
DI float half_join_sum(float pa, float pb) {
    auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, pa), __builtin_bit_cast(unsigned, pb), false, false);
    const unsigned r0 = r[0], r1 = r[1];
    return __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
}

// The duo layout only runs in persistent launches (launch_persistent): a team walks many instances, one at a time, so its workspaces — the
// particle x horizon tensor, the activation checkpoint, the per-group partial sums, the control table when that lives in global memory —
// are indexed by the team's SLOT (workgroup x teams per workgroup), not by the instance: 1,536 rows on an MI355X whatever the batch size
// (C2 at 12,288 instances: 2.3 GB instead of 18.8 GB; a 98,304-instance launch fits one GPU).
template <class Team>
DI int duo_workspace_slot() { return __builtin_amdgcn_readfirstlane((int)(blockIdx.x * Team::IPB) + Team::team()); }

// Which groups a wave's halves own in pair gp of an instance with G groups. Odd G: the last pair has no group B; its upper half then
// shadows group A (same inputs, so it stays finite) and neither stores nor contributes to a sum, and pass B is skipped.
struct DuoPair {
    int g;          // this lane's group
    bool hasB;      // wave-uniform: the pair has a real group B
    bool own;       // this lane half owns a real group
};
DI DuoPair duo_pair(int gp, int G, int h) {
    DuoPair p;
    p.hasB = (2 * gp + 1) < G;
    p.own = (h == 0) || p.hasB;
    p.g = 2 * gp + ((h && p.hasB) ? 1 : 0);
    return p;
}

// The rotation matrix formed a second time (same expressions, same bits) where it is needed again, instead of being kept alive across both MLP
// passes; the empty asm hides the quaternion's provenance so that the compiler does not simply keep the first copy
DI void duo_reform_rotation(const float* x, float* Rm) {
    float xq[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) xq[i] = x[i];
#pragma unroll
    for (int i = 6; i < 10; ++i) asm volatile("" : "+v"(xq[i]));
    rot_from_q(xq, Rm);
}

// TIMING-ONLY diagnostic (tools/build_variant.sh ckl2 "-DSDEMPC_VAR_CKPT_CACHED=1"; results are WRONG): every step's layer-2 activation rows alias the rows of
// step 0, so the checkpoint stream of the gradient evaluations stays in the caches instead of crossing HBM twice — the upper bound of what removing the
// stream (recomputing layer 2 in the adjoint) could buy at the power cap, before the recompute is paid for (profiles/r4_ab.txt).
#ifndef SDEMPC_VAR_CKPT_CACHED
#define SDEMPC_VAR_CKPT_CACHED 0
#endif
#if SDEMPC_VAR_CKPT_CACHED
#define CKPT_T(t) 0
#else
#define CKPT_T(t) (t)
#endif
// One Euler-Maruyama step for the wave's 64 particles. CKPT: stream the second hidden layer of both passes to the groups' checkpoint
// rows (acA / acB: this step's rows of group A / group B).
// Issue priority, rotated among the waves of a SIMD. The arbiter serves the higher s_setprio first and, among equals, the OLDER wave slot:
// left alone, identical work takes 382 / 438 / 765 ms in wave slots 0 / 1 / 2 (tools/phase_clock.py), and a launch that gives every team the
// same number of instances waits for the teams in slot 2. Every step each wave sets its priority to (time slice + its wave slot) mod 3: the
// three waves of a SIMD read the same clock, so at any moment they hold three different priorities and each is the favoured one for a third of
// the time (slices of 0.66 ms; 0.08 ... 42 ms measured alike, 10 us slices recover only half). Striped launches: C2 at 3,072 instances
// 2,905 -> 3,150 solves/s, C3 at 1,536 1,486 -> 1,583, C5 at 768 (one round) 884 -> 959; ticketed launches unchanged (3,150). Only the
// persistent duo launches rotate: a plain grid of one-wave teams (C1) is rebalanced by the dispatcher as workgroups retire, and there the
// arbiter's own oldest-first order is the better one (rotation: 55,300 -> 52,750 solves/s). Priorities by PHASE instead of by time slice
// (the stall-prone adjoint sweep first, or last) were measured too: -0.4 % / -2.2 %.
constexpr int PRIO_SLICE_LOG2 = 16;               // 2^16 ticks of the 100 MHz s_memrealtime clock
DI void duo_rotate_priority() {
#if SDEMPC_VAR_NO_ROTATE      // diagnostic builds: the arbiter's own order (profiles/r2_phase_clock.txt)
    return;
#endif
    const unsigned slot = __builtin_amdgcn_s_getreg(63492) & 15u;                              // HW_REG_HW_ID[3:0]: wave slot on its SIMD
    const unsigned r = ((unsigned)(__builtin_amdgcn_s_memrealtime() >> PRIO_SLICE_LOG2) + slot) % 3u;
    if (r == 0) __builtin_amdgcn_s_setprio(0);
    else if (r == 1) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(2);
}

// ------------------------------------------------------------------------------------------------
// Noise staging of the forward sweeps. The six noise rows of step t are used once, by the Euler-Maruyama update at the END of step t.
// Held in registers from a request early enough to cover HBM latency they are live across both MLP passes, and the compiler spilled
// them: load -> s_waitcnt -> scratch store, six synchronous HBM round trips per step. They travel by LDS-DMA instead
// (global_load_lds_dword: no destination registers): requested right after step t - 1 has consumed its own rows, taken from LDS just
// before the update of step t. vmcnt retires in order, so the wait leaves the PENDING youngest operations (this step's checkpoint
// stores, all issued after the request) in flight.
// ------------------------------------------------------------------------------------------------
DI void duo_noise_request(const float* nzb, unsigned nzo, int t, const float* stage) {
    // Written as one asm block: the compiler knows nothing of these six DMAs (declared through the builtin it made every LDS read that
    // follows — the weights at the top of the next step — wait for them: vmcnt(0) right behind the request). Row i lands at
    // stage + 256 i bytes: LDS address = M0 + instruction offset + 4 lane, memory address = base + lane offset + instruction offset,
    // so M0 advances by 128 per row beside the instruction offset's 128. (SALU write of M0 -> LDS-DMA needs a wait state: s_nop.)
    const float* rows = nzb + (size_t)t * (NN * 32);                                  // uniform: SGPR pair
    const unsigned m0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(const __attribute__((address_space(3))) float*)stage);
    unsigned m0_keep;        // M0 is saved and restored inside the block: whatever the compiler keeps there (movrel, readlane-by-M0) survives it
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\t"
        "s_add_u32 m0, m0, 0x80\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2 offset:128\n\t"
        "s_add_u32 m0, m0, 0x80\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2 offset:256\n\t"
        "s_add_u32 m0, m0, 0x80\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2 offset:384\n\t"
        "s_add_u32 m0, m0, 0x80\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2 offset:512\n\t"
        "s_add_u32 m0, m0, 0x80\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2 offset:640\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(m0_keep) : "v"(nzo * 4u), "s"(rows), "s"(m0) : "memory", "scc");
}
template <int PENDING>
DI void duo_noise_take(const float* stage, int lane, float* xi) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PENDING) : "memory");
#pragma unroll
    for (int i = 0; i < NN; ++i) xi[i] = stage[i * 64 + lane];
}

// NZS: noise from the staging rows (above); !NZS: from the caller's registers xi (instances whose LDS has no room for staging rows
// without losing a workgroup per CU: long horizons)
template <int F16, bool CKPT, bool NZS>
DI void duo_step_fwd(const KArgs& a, const Smem& sm, const WaveW& ww, int t, int h, int lane, bool hasB, const float* x, const float* xi_reg, float* xn,
                     StepAux& A, float* acA, float* acB) {
    const float* ust = sm.ust + t * UST;
    float z[NN], zA[NN], zB[NN];
    fwd_head(x, A.Rm, z);
#pragma unroll
    for (int k = 0; k < NN; ++k) half_split(z[k], zA[k], zB[k]);
    SCHED_PHASE();
    float PA[7], PB[7];
    fwd_mlp_partials<F16, false, CKPT ? 4 : 7>(a, sm, ww, ust, h, lane, zA, A, PA);
    if constexpr (CKPT) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(acA + (q * 64 + lane) * 4) = make_float4(A.h2[4 * q], A.h2[4 * q + 1], A.h2[4 * q + 2], A.h2[4 * q + 3]);
    }
    SCHED_PHASE();
    if (hasB) {
        fwd_mlp_partials<F16, false, CKPT ? 4 : 7>(a, sm, ww, ust, h, lane, zB, A, PB);
        if constexpr (CKPT) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(acB + (q * 64 + lane) * 4) = make_float4(A.h2[4 * q], A.h2[4 * q + 1], A.h2[4 * q + 2], A.h2[4 * q + 3]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 7; ++i) PB[i] = PA[i];
    }
    SCHED_PHASE();
    float o[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) o[i] = half_join_sum(PA[i], PB[i]) + a.M.b3[i];
    const float eta = sigmoid_spec(half_join_sum(PA[6], PB[6]) + a.M.b3n);
    SCHED_PHASE();
    if constexpr (CKPT) duo_reform_rotation(x, A.Rm);   // gradient sweep: nine registers less across both MLP passes
    float xi[NN];
    if constexpr (NZS) duo_noise_take<CKPT ? 4 : 0>(sm.nzs, lane, xi);     // (gradient sweep: pass A's four checkpoint stores at least are younger than the request)
    else {
#pragma unroll
        for (int i = 0; i < NN; ++i) xi[i] = xi_reg[i];
    }
    fwd_tail(a, sm, ust, t, x, xi, A.Rm, o, eta, xn, A);
}

// ------------------------------------------------------------------------------------------------
// team-level rollout: expected cost of control sequence u (LDS). Same contract as block_rollout.
// ------------------------------------------------------------------------------------------------
template <class Team, int F16, bool NZS>
DI float duo_rollout(const KArgs& a, const Smem& sm, const WaveW& ww, const float* u, int b, int tid, bool store_traj, float* xmean_out) {
    b = opaque_s(b); tid = opaque_v(tid);
    const int H = a.H, G = a.G, P = a.P;
    const int lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave-uniform: the pair's base addresses stay in SGPRs
    const bool want_mean = xmean_out != nullptr;
    CLK_BEGIN(t_roll);
    Team::sync();
    block_prepass<Team>(a, sm, u, tid);
    const int PS = part_stride(H);
    const int ws = duo_workspace_slot<Team>();         // workspaces (trajectory, checkpoint, partial sums) belong to the team's SLOT, not to the instance
    float* prows = a.part + (size_t)ws * G * PS;
    float cu = block_ucost<Team>(a, sm, u, tid);
    Team::sync();                                      // prepass table visible to every wave of the team
    const int NPAIR = (G + 1) >> 1;
    for (int gp = wave; gp < NPAIR; gp += Team::NWAVES) {
        const DuoPair pr = duo_pair(gp, G, h);
        const bool valid = pr.own && (pr.g * 32 + j) < P;
        // uniform base of the pair (group 2 gp) + a 32-bit per-lane offset (this lane's group and column): scalar-base addressing
        const unsigned gofs = (unsigned)(pr.g - 2 * gp);
        const float* nzb = a.noise + ((size_t)(b * G + 2 * gp) * H) * NN * 32;         // uniform (SGPR) bases of the pair ...
        float* tjb = a.traj + ((size_t)(ws * G + 2 * gp) * (H + 1)) * NX * 32;
        const unsigned nzo = gofs * (unsigned)(H * NN * 32) + (unsigned)j;              // ... + 32-bit per-lane offsets (group, column)
        const unsigned tjo = gofs * (unsigned)((H + 1) * NX * 32) + (unsigned)j;
        float* xm = prows + (size_t)pr.g * PS;          // this group's row of per-step particle sums (SPEC.md §6.1/§6.3)
        float x[NX], xn[NX], xi[NN];
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = a.x0[b * NX + i];      // (read here, per pair: nothing of it stays live across the pair loop)
        if constexpr (NZS) {
            duo_noise_request(nzb, nzo, 0, sm.nzs);
            // the initial state is in registers BEFORE the step loop: left to the compiler its wait sits in the loop header (first use),
            // where every iteration it would also wait for the DMAs requested at the end of the previous step
#pragma unroll
            for (int i = 0; i < NX; ++i) asm volatile("" : "+v"(x[i]));
        } else {
#pragma unroll
            for (int i = 0; i < NN; ++i) xi[i] = nzb[nzo + (unsigned)(i * 32)];
        }
        if (store_traj && pr.own) {
#pragma unroll
            for (int i = 0; i < NX; ++i) tjb[tjo + (unsigned)(i * 32)] = x[i];
        }
        if (want_mean) {
#pragma unroll
            for (int i = 0; i < NX; ++i) { float s = group_bfly32(valid ? x[i] : 0.0f); if (j == 0 && pr.own) xm[i] = s; }
        }
        float J = 0.0f;
        StepAux A;
        CLK_BEGIN(t_loop);
        for (int t = 0; t < H; ++t) {
            duo_rotate_priority();
            // !NZS: next step's rows requested a whole step ahead into registers (a just-in-time load at the top of the step measured 33 %
            // slower: the compiler sinks it to its use)
            float xin[NN];
            if constexpr (!NZS) {
                if (t + 1 < H) {
#pragma unroll
                    for (int i = 0; i < NN; ++i) xin[i] = nzb[nzo + (unsigned)(((t + 1) * NN + i) * 32)];
                }
            }
            duo_step_fwd<F16, false, NZS>(a, sm, ww, t, h, lane, pr.hasB, x, xi, xn, A, nullptr, nullptr);
            if constexpr (NZS) { if (t + 1 < H) duo_noise_request(nzb, nzo, t + 1, sm.nzs); }     // a whole step ahead of its use
            float l = stage_cost<false>(a, xn, sm.xref + (t + 1) * NX, nullptr);
            l = FMA(a.C.res_mult * A.eta, A.eta, l);
            J = FMA(sm.disc[t], l, J);
#pragma unroll
            for (int i = 0; i < NX; ++i) x[i] = xn[i];
            if constexpr (!NZS) {
                if (t + 1 < H) {
#pragma unroll
                    for (int i = 0; i < NN; ++i) xi[i] = xin[i];
                }
            }
            if (store_traj && pr.own) {
#pragma unroll
                for (int i = 0; i < NX; ++i) tjb[tjo + (unsigned)(((t + 1) * NX + i) * 32)] = x[i];
            }
            if (want_mean) {
#pragma unroll
                for (int i = 0; i < NX; ++i) { float s = group_bfly32(valid ? x[i] : 0.0f); if (j == 0 && pr.own) xm[(t + 1) * NX + i] = s; }
            }
        }
        CLK_END(2, t_loop);
        const float T = group_bfly32(valid ? J : 0.0f);
        if (j == 0 && pr.own) xm[PS - 1] = T;          // group total of the particle costs (last element of the group's row)
    }
    Team::sync();
    const float tot = group_ordered_sum(prows, G, PS, PS - 1);
    if (want_mean) {
        for (int i = tid; i < (H + 1) * NX; i += Team::NT) xmean_out[i] = group_ordered_sum(prows, G, PS, i) * a.invP;
    }
    CLK_END(1, t_roll);
    return FMA(tot, a.invP, cu);
}

// ------------------------------------------------------------------------------------------------
// team-level cost + gradient (forward sweep with trajectory / checkpoint store, adjoint sweep). Same contract as block_cost_grad;
// no register prefetch buffer: built for three waves per SIMD, which hide the latency of the adjoint's loads.
// ------------------------------------------------------------------------------------------------
template <class Team, int M, int F16, bool NZS>
DI float duo_cost_grad(const KArgs& a, const Smem& sm, const WaveW& ww, const float* y, float* gout, int b, int tid) {
    b = opaque_s(b); tid = opaque_v(tid);
    const int H = a.H, G = a.G, P = a.P;
    constexpr int nq = M + 4;
    const int lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave-uniform: the pair's base addresses stay in SGPRs
    CLK_BEGIN(t_grad);
    Team::sync();
    block_prepass<Team>(a, sm, y, tid);
    const int PS = part_stride(H);
    const int ws = duo_workspace_slot<Team>();
    float* prows = a.part + (size_t)ws * G * PS;
    float cu = block_ucost<Team>(a, sm, y, tid);
    Team::sync();
    const int NPAIR = (G + 1) >> 1;
    for (int gp = wave; gp < NPAIR; gp += Team::NWAVES) {
        const DuoPair pr = duo_pair(gp, G, h);
        const bool valid = pr.own && (pr.g * 32 + j) < P;
        const int gA = 2 * gp, gB = pr.hasB ? 2 * gp + 1 : gA;
        const unsigned gofs = (unsigned)(pr.g - gA);
        const float* nzb = a.noise + ((size_t)(b * G + gA) * H) * NN * 32;             // uniform (SGPR) bases of the pair ...
        float* tjb = a.traj + ((size_t)(ws * G + gA) * (H + 1)) * NX * 32;
        const unsigned nzo = gofs * (unsigned)(H * NN * 32) + (unsigned)j;              // ... + 32-bit per-lane offsets (group, column)
        const unsigned tjo = gofs * (unsigned)((H + 1) * NX * 32) + (unsigned)j;
        float* acA = a.act + ((size_t)(ws * G + gA) * H) * ACT_STRIDE;      // checkpoint rows of the two groups (tiles: whole wave)
        float* acB = a.act + ((size_t)(ws * G + gB) * H) * ACT_STRIDE;
        const unsigned aso = gofs * (unsigned)(H * ACT_STRIDE) + 1024u + (unsigned)j * 8u;   // this lane's particle: step scalars, relative to acA
        float* Sq = prows + (size_t)pr.g * PS;                               // this group's row of per-step adjoint sums (SPEC.md §6.1)
        float x[NX], xn[NX], xi[NN];
        StepAux A;
        CLK_BEGIN(t_fwd);
        // ---- forward sweep, x_t and the second hidden layer streamed to HBM ----
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = a.x0[b * NX + i];      // (read here, per pair: nothing of it stays live across the pair loop)
        if (pr.own) {
#pragma unroll
            for (int i = 0; i < NX; ++i) tjb[tjo + (unsigned)(i * 32)] = x[i];
        }
        if constexpr (NZS) {
            duo_noise_request(nzb, nzo, 0, sm.nzs);
#pragma unroll
            for (int i = 0; i < NX; ++i) asm volatile("" : "+v"(x[i]));      // (as in duo_rollout: no wait for the initial state inside the loop)
        } else {
#pragma unroll
            for (int i = 0; i < NN; ++i) xi[i] = nzb[nzo + (unsigned)(i * 32)];
        }
        float J = 0.0f;
        for (int t = 0; t < H; ++t) {
            duo_rotate_priority();
            float xin[NN];
            if constexpr (!NZS) {
                if (t + 1 < H) {
#pragma unroll
                    for (int i = 0; i < NN; ++i) xin[i] = nzb[nzo + (unsigned)(((t + 1) * NN + i) * 32)];
                }
            }
            duo_step_fwd<F16, true, NZS>(a, sm, ww, t, h, lane, pr.hasB, x, xi, xn, A, acA + (size_t)CKPT_T(t) * ACT_STRIDE, acB + (size_t)CKPT_T(t) * ACT_STRIDE);
            if constexpr (NZS) { if (t + 1 < H) duo_noise_request(nzb, nzo, t + 1, sm.nzs); }   // before this step's scalar / trajectory stores: older than they are
            if (pr.own) {
                const unsigned so = aso + (unsigned)(t * ACT_STRIDE);
                *reinterpret_cast<float4*>(acA + so) = make_float4(A.eta, A.Fb[0], A.Fb[1], A.Fb[2]);
                acA[so + 4] = A.rn;
            }
            float l = stage_cost<false>(a, xn, sm.xref + (t + 1) * NX, nullptr);
            l = FMA(a.C.res_mult * A.eta, A.eta, l);
            J = FMA(sm.disc[t], l, J);
#pragma unroll
            for (int i = 0; i < NX; ++i) x[i] = xn[i];
            if constexpr (!NZS) {
                if (t + 1 < H) {
#pragma unroll
                    for (int i = 0; i < NN; ++i) xi[i] = xin[i];
                }
            }
            if (pr.own) {
#pragma unroll
                for (int i = 0; i < NX; ++i) tjb[tjo + (unsigned)(((t + 1) * NX + i) * 32)] = x[i];
            }
        }
        { const float T = group_bfly32(valid ? J : 0.0f); if (j == 0 && pr.own) Sq[PS - 1] = T; }
        CLK_END(4, t_fwd);
        CLK_BEGIN(t_adj);
        // ---- adjoint sweep: x (registers) currently holds x_H ----
        float lam[NX], xt[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) lam[i] = 0.0f;
        // this wave's own stores of x_t / activations must be visible to its loads (same CU: workgroup scope)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        for (int t = H - 1; t >= 0; --t) {
            duo_rotate_priority();                   // (requesting this step's loads at top priority instead: no gain)
            // (the 25 loads below are requested at the top of their own step. A register prefetch a step ahead spills; the LDS staging rows
            // have room for 6 of the 24 dwords. A timing-only build that redirected these loads to cache-resident rows bounds what a perfect
            // prefetch could buy: 12 % of the adjoint sweep, 4 % of the launch.)
            const float* apA = acA + (size_t)CKPT_T(t) * ACT_STRIDE;
            const float* apB = acB + (size_t)CKPT_T(t) * ACT_STRIDE;
            float4 hA[4], hB[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) hA[q] = *reinterpret_cast<const float4*>(apA + (q * 64 + lane) * 4);
            float touched = 0.0f;
            {
                // Pass B's checkpoint tile (4 KB = 32 lines of 128 bytes) is only TOUCHED here — one dword per line, into a register nobody reads — and loaded right before
                // pass B needs it (an L2 hit by then). The compiler does not know the touch: an unknown OLDER load only makes its vmcnt waits longer, never shorter.
                // The destination register must stay reserved until the load has certainly returned — the compiler believes the asm is over at once and would hand the
                // register to someone else, whose value the returning dword then overwrites (a memory fault within the first launch, round 5): `touched` is kept alive
                // by an empty asm BEHIND pass B, which has waited for its own, younger checkpoint loads by then (loads return in order).
                if (pr.hasB) {
                    const float* tp = apB + (lane & 31) * 32;
                    asm volatile("global_load_dword %0, %1, off" : "=v"(touched) : "v"(tp) : "memory");
                }
            }
            const unsigned so = aso + (unsigned)(t * ACT_STRIDE);
            const float4 ns4 = *reinterpret_cast<const float4*>(acA + so);
            const float nrn = acA[so + 4];
            {
#pragma unroll
                for (int i = 0; i < NX; ++i) xt[i] = tjb[tjo + (unsigned)((t * NX + i) * 32)];
#pragma unroll
                for (int i = 0; i < NN; ++i) xi[i] = nzb[nzo + (unsigned)((t * NN + i) * 32)];
            }
            // x = x_{t+1}: fold the stage-cost gradient into the incoming adjoint
            const float dsc = sm.disc[t];
            {
                float gx[NX];
                stage_cost<true>(a, x, sm.xref + (t + 1) * NX, gx);
#pragma unroll
                for (int i = 0; i < NX; ++i) lam[i] = FMA(dsc, gx[i], lam[i]);
            }
            SCHED_PHASE();
            const float* ust = sm.ust + t * UST;
            float z[NN], zA[NN], zB[NN];
            fwd_head(xt, A.Rm, z);
#pragma unroll
            for (int k = 0; k < NN; ++k) half_split(z[k], zA[k], zB[k]);
            A.eta = ns4.x; A.Fb[0] = ns4.y; A.Fb[1] = ns4.z; A.Fb[2] = ns4.w; A.rn = nrn;
#pragma unroll
            for (int i = 0; i < 3; ++i) A.Jom[i] = a.M.J[i] * xt[10 + i];
#pragma unroll
            for (int i = 0; i < 4; ++i) A.qn[i] = x[6 + i];   // q_{t+1}
            const float ebc = dsc * ((2.0f * a.C.res_mult) * A.eta);
            VjpTmp T;
            float gq[12], lamn[NX];
            vjp_head<M>(a, sm, t, xt, xi, A, lam, ebc, T, gq);
            float eA, eB, obA[6], obB[6];
            float adj_inv = 1.0f;
            ObLimbs OA, OB;
            if constexpr (FAST && F16 != 0) {       // SPEC.md §10e: one power of two per particle, applied before the broadcast (once for both passes)
                const AdjScale S = adj_scale(a, T.ebraw, T.ob);
                adj_inv = S.inv;
                half_split(T.ebraw * S.s2, eA, eB);
                float obs[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) obs[i] = T.ob[i] * S.s2;
                const ObLimbs O = ob_limbs(obs);          // per particle, once for both passes: the limbs are what is broadcast
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    float la, lb;
                    half_split(__builtin_bit_cast(float, O.r1[i]), la, lb); OA.r1[i] = __builtin_bit_cast(unsigned, la); OB.r1[i] = __builtin_bit_cast(unsigned, lb);
                    half_split(__builtin_bit_cast(float, O.r2[i]), la, lb); OA.r2[i] = __builtin_bit_cast(unsigned, la); OB.r2[i] = __builtin_bit_cast(unsigned, lb);
                }
            } else {
            half_split(T.ebraw, eA, eB);
#pragma unroll
            for (int i = 0; i < 6; ++i) half_split(T.ob[i], obA[i], obB[i]);
            }
            SCHED_PHASE();
            float zb[NN];
            if constexpr (FAST && F16 != 0) {
                // (only AdjRegs<M>::N registers of a pass's result tile hold existing rows)
                constexpr int NR = AdjRegs<M>::N;
                float rA[NR], rB[NR];
                // (pass B's six broadcast inputs wait in this wave's noise staging rows — idle during the adjoint sweep — while pass A runs)
                if constexpr (NZS) {
#pragma unroll
                    for (int k = 0; k < NN; ++k) sm.nzs[k * 64 + lane] = zB[k];
                }
                adj_mlp_pass_mp<NR, false, F16>(sm, ww, ust, h, lane, zA, hA, eA, OA, rA, []() {});
                SCHED_PHASE();
                if constexpr (NZS) {
#pragma unroll
                    for (int k = 0; k < NN; ++k) zB[k] = sm.nzs[k * 64 + lane];
                }
                // (parking pass A's result rows there in turn while pass B runs was tried: the allocator answered with MORE spills in three of four kernels)
                if (pr.hasB) adj_mlp_pass_mp<NR, true, F16>(sm, ww, ust, h, lane, zB, hB, eB, OB, rB, [&]() {
#pragma unroll
                    for (int q = 0; q < 4; ++q) hB[q] = *reinterpret_cast<const float4*>(apB + (q * 64 + lane) * 4);       // (touched at the top of the step: an L2 hit)
                });
                else {
#pragma unroll
                    for (int r = 0; r < NR; ++r) asm volatile("v_mov_b32 %0, %1" : "=v"(rB[r]) : "v"(rA[r]));      // (a register of its own: the swap below aliases otherwise)
                }
                SCHED_PHASE();
                // rows of pass A belong to lanes < 32, rows of pass B to lanes >= 32: ONE swap per register pair puts row rowmap(r, 0) of either pass into the
                // first result and row rowmap(r, 1) into the second, each in the lane half that owns the particle
                float lo[8], hi[8];
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, rA[r]), __builtin_bit_cast(unsigned, rB[r]), false, false);
                    const unsigned s0 = sw[0], s1 = sw[1];
                    lo[r] = __builtin_bit_cast(float, s0); hi[r] = __builtin_bit_cast(float, s1);
                }
#pragma unroll
                for (int k = 0; k < NN; ++k) zb[k] = (adj_row_upper(k) ? hi[adj_row_reg(k)] : lo[adj_row_reg(k)]) * adj_inv;
#pragma unroll
                for (int jj = 0; jj < M; ++jj) gq[jj] = (adj_row_upper(NN + jj) ? hi[adj_row_reg(NN + jj)] : lo[adj_row_reg(NN + jj)]) * adj_inv;
            } else {
            float PzA[NN], PuA[M], PzB[NN], PuB[M];
            // per pass: layer 1 recomputed tile by tile (6 MFMAs, 32 tanh), layer 2 from the checkpoint (adj_mlp_pass)
            adj_mlp_pass<M, F16>(sm, ww, ust, h, lane, zA, hA, eA, obA, PzA, PuA);
            SCHED_PHASE();
            if (pr.hasB) {
                // pass B's second-layer checkpoint: touched at the top of the step, loaded here (until round 5 it was requested before pass A and held in registers
                // across it — where the allocator, short of registers, spilled one or two of its quads right behind the load: a synchronous HBM round trip each)
#pragma unroll
                for (int q = 0; q < 4; ++q) hB[q] = *reinterpret_cast<const float4*>(apB + (q * 64 + lane) * 4);
                SCHED_PHASE();
                adj_mlp_pass<M, F16>(sm, ww, ust, h, lane, zB, hB, eB, obB, PzB, PuB);
            } else {
#pragma unroll
                for (int k = 0; k < NN; ++k) PzB[k] = PzA[k];
#pragma unroll
                for (int jj = 0; jj < M; ++jj) PuB[jj] = PuA[jj];
            }
            SCHED_PHASE();
#pragma unroll
            for (int k = 0; k < NN; ++k) zb[k] = half_join_sum(PzA[k], PzB[k]);
#pragma unroll
            for (int jj = 0; jj < M; ++jj) gq[jj] = half_join_sum(PuA[jj], PuB[jj]);
            }
            asm volatile("" :: "v"(touched));     // (see the touch at the top of the step)
            duo_reform_rotation(xt, A.Rm);       // nine registers less across both MLP passes
            vjp_tail(sm, t, xt, A, lam, T, zb, lamn);
#pragma unroll
            for (int i = 0; i < NX; ++i) { lam[i] = lamn[i]; x[i] = xt[i]; }
            // particle sums of the nq per-step adjoint outputs: each lane half reduces its own group (halves never mix below xor 32)
#pragma unroll
            for (int k = 0; k < nq; ++k) {
                const float s = group_bfly32(valid ? gq[k] : 0.0f);
                if (j == 0 && pr.own) Sq[t * 12 + k] = s;
            }
        }
        CLK_END(5, t_adj);
    }
    Team::sync();
    const float tot = group_ordered_sum(prows, G, PS, PS - 1);
    assemble_gradient<Team, M>(a, sm, y, gout, tid, [&](int q) { return group_ordered_sum(prows, G, PS, q); });
    Team::sync();
    CLK_END(3, t_grad);
    return FMA(tot, a.invP, cu);
}
