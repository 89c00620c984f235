// sdempc_spec.inc.h — speculative cooperative solve kernel (state machine)
// Fragment of sdempc_kernels.hip: included inside namespace sdempc::{exact|fastm} (it is compiled twice, see there); not a
// stand-alone header.
// ================================================================================================
// Speculative cooperative solve (smallest batches): up to SEVEN groups of ceil(P/4) workgroups per instance. While groups 0 and 1
// (and 5: the third trial, with group 6 evaluating the gradient behind it) evaluate the first line-search trials of iteration k side
// by side, groups 2, 3 and 4 already evaluate the gradient of
// iteration k+1 at the three points the optimiser can move to: where it goes if it ends on trial 1 resp. trial 2 with an
// improvement, and xk (no improvement). The step sizes of the trials are known before any of them is evaluated (s, s*dec, ...),
// so are the restart tests. One grid barrier per SEQUENTIAL phase; the parallel and the reduction phases hand over through tagged words (below). The optimiser itself is unchanged and runs redundantly in every
// workgroup: a gradient is a pure function of its point, so using the pre-computed one (when the speculation hits — the trial
// costs decide that) gives the same bits as computing it afterwards; on a miss the iteration falls back to the sequential order.
// Written as a state machine with ONE call site of the particle work, so that the rollout and the gradient sweep are each
// instantiated once (the first version inlined them at seven sites: 140 KB of code, 2x slower sweeps).
// ================================================================================================
constexpr int SPEC_GROUPS = 7, SPEC_CKS = 4;      // (SPEC_SLOTS = 9: sdempc_kernels.hip, with the workspace layout)
constexpr int SLOT_SEQ = 6, SLOT_GRAD = 5;        // slots 0..4 and 7, 8: the items of a parallel phase
constexpr int SLOT_T3 = 7, SLOT_Y3 = 8;           // third parallel trial and the candidate gradient behind it (groups 5, 6)

// Streamed hand-off: every per-particle output of this kernel is a 64-bit word {value, tag} (tag = number of the phase that wrote it, unique within a
// launch; the words are zeroed by the host before the launch), written by ONE aligned store — a reader that finds the tag has the value, whatever else
// the writer still has in flight. So the parallel phase ends WITHOUT a grid barrier: every workgroup polls the (few) trial costs it needs, the reduction
// phase polls the words of the gradient it reduces, and the only thing a barrier still ordered — a fast workgroup overwriting (two iterations later) words
// a slow one is still reading — is kept by the arrival counter (coop_arrive / coop_arrived_wait). The sequential phases keep their barrier.
DI const unsigned long long* spec_words(const CoopCtx& C, int PS, unsigned par, int slot) {
    return reinterpret_cast<const unsigned long long*>(C.pp) + (size_t)(par * SPEC_SLOTS + slot) * PS * C.Ppad;
}
DI Lane2IO lane_io_slot(const KArgs& a, const CoopCtx& C, int b, int p, unsigned par, int slot, unsigned tag) {
    const int H = a.H;
    Lane2IO io;
    io.x0 = a.x0 + (size_t)b * NX;
    io.ck = C.ck + (size_t)p * (H + 1) * COOP_ROW;
    io.out = C.pp + 2 * ((size_t)(par * SPEC_SLOTS + slot) * part_stride(H) * C.Ppad + p); io.os = 2 * C.Ppad; io.tag = tag;
    return io;
}
// in every workgroup: expected cost of control sequence u whose particle outputs sit in (par, slot), written by phase `tag`
DI float spec_cost(const KArgs& a, const Smem& sm, CoopCtx& C, int tid, unsigned par, const float* u, int slot, unsigned tag) {
    const int lane = tid & 63, wave = tid >> 6, PS = part_stride(a.H);
    const float cu = block_ucost<TeamBlock>(a, sm, u, tid);
    __syncthreads();
    if (wave == 0) {
        const float t0 = coop_total_tagged(C, spec_words(C, PS, par, slot) + (size_t)(PS - 1) * C.Ppad, a.P, a.G, lane, tag, __builtin_amdgcn_s_memrealtime());
        if (lane == 0) sm.red[12] = t0;
    }
    __syncthreads();
    return FMA(sm.red[12], a.invP, cu);
}
// K reductions of SPEC.md §6.2 with one pair of barriers (each one the arithmetic of team_reduce256: chain e = tid, tid + 256, ...;
// 64-lane butterflies; ((w0+w1)+w2)+w3) — a lone workgroup pays ~1.4 us per separate reduction, and an iteration has nine of them
template <int K, class F>
DI void spec_reduce_n(float* mred, int N, int tid, float (&out)[K], F&& elem) {
    float acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = 0.0f;
    for (int e = tid; e < N; e += 256) elem(e, acc);
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = wave_bfly64(acc[k]);
    __syncthreads();
    if ((tid & 63) == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) mred[4 * k + (tid >> 6)] = acc[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) out[k] = ((mred[4 * k] + mred[4 * k + 1]) + mred[4 * k + 2]) + mred[4 * k + 3];
}
constexpr int SPEC_MRED = 48 + 64, SPEC_XV = 6;  // extra control vectors xn1..3, y1..3; floats of reduction scratch (40 used) + the per-motor constants (SpecMot) behind them
// The per-motor constants of KArgs the head of an iteration indexes by motor, copied to LDS once: indexed reads of the kernel argument are memory loads the
// compiler cannot hoist out of the state machine's loop, and they sat in the dependent chain of the ten reductions between the totals and the next phase.
// The functions below are the statements of ucost_elem / slew_dw / assemble_gradient (sdempc_kernels.hip, sdempc_lane.inc.h) on those copies.
struct SpecMot { const float *uref, *ulo, *uhi, *shi, *slo, *dir, *ry, *rx; };
DI float ucost_elem_k(const KArgs& a, const Smem& sm, const SpecMot& K, const float* u, int e, int t, int j, int m) {
    float du = u[e] - K.uref[j];
    float c = (a.C.uerr * du) * du;
    if (t >= 1) {
        float ds = u[e] - u[e - m];
        c = FMA(a.C.slew * ds, ds, c);
        if (a.C.has_sc) {
            float hi = ds - K.shi[j]; hi = hi < 0.0f ? 0.0f : hi;
            float lo = K.slo[j] - ds; lo = lo < 0.0f ? 0.0f : lo;
            c = FMA(a.C.slew_cc * hi, hi, c);
            c = FMA(a.C.slew_cc * lo, lo, c);
        }
    }
    return sm.disc[t] * c;
}
DI float slew_dw_k(const KArgs& a, const SpecMot& K, const float* u, int t, int j, int m) {
    float ds = u[t * m + j] - u[(t - 1) * m + j];
    float d = (2.0f * a.C.slew) * ds;
    if (a.C.has_sc) {
        float hi = ds - K.shi[j]; hi = hi < 0.0f ? 0.0f : hi;
        float lo = K.slo[j] - ds; lo = lo < 0.0f ? 0.0f : lo;
        d = FMA(2.0f * a.C.slew_cc, hi - lo, d);
    }
    return d;
}
DI float assemble_elem_k(const KArgs& a, const Smem& sm, const SpecMot& K, const float* y, int e, int t, int jj, int m, const float* S) {
    const int H = a.H;
    float uj = y[e];
    float dT = FMA(2.0f * a.M.ct2, uj, a.M.ct1);
    float dM = K.dir[jj] * FMA(2.0f * a.M.cm2, uj, a.M.cm1);
    float acc = S[0];
    acc = FMA(S[1], dT, acc);
    acc = FMA(S[2], K.ry[jj] * dT, acc);
    acc = FMA(S[3], -(K.rx[jj] * dT), acc);
    acc = FMA(S[4], dM, acc);
    float du = uj - K.uref[jj];
    float dw = 0.0f;
    if (t >= 1) dw = slew_dw_k(a, K, y, t, jj, m);
    float gcu = sm.disc[t] * FMA(2.0f * a.C.uerr, du, dw);
    if (t + 1 < H) { float dwn = slew_dw_k(a, K, y, t + 1, jj, m); gcu = FMA(-sm.disc[t + 1], dwn, gcu); }
    return FMA(acc, a.invP, gcu);
}
// one workgroup per CU (512 registers per lane: what does not fit the 256 VGPRs spills to AGPRs, not to scratch memory — with two
// workgroups per CU the adjoint loop carried 43 scratch accesses per step and ran 3x slower)
// DIRECT: single-particle instances (P == 1, every MPC YAML the reference ships): a particle is its own total, no reduction phase
template <int M, bool DIRECT>
__global__ void __launch_bounds__(256, 1) sdempc_solve_spec_kernel(KArgs a) {
    using Team = TeamBlock;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // a.coop_ngrp groups (2..7) per instance take the roles T1, T2, S(y2), S(xk), S(y1), T3, S(y3) in this order of usefulness
    // (measured at C2: the line search ends on trial 2 in 60 %, there is no improvement in 31 %, it ends on trial 1 in 28 % of the iterations)
    // (the instantiations for four and six motors are launched for exactly those counts: sdempc_kernels.hip, launch_solve_spec — a compile-time m turns the
    // e / m, e % m of every elementwise loop of the optimiser into multiplications)
    const int nwg = a.coop_nwg, ng = a.coop_ngrp, per = ng * nwg, H = a.H, m = M == 8 ? a.m : M, N = H * m, PS = part_stride(H);
    const int b_ = blockIdx.x / per, r_ = blockIdx.x - b_ * per, grp = r_ / nwg;
    const bool have_y2 = ng >= 3, have_xk = ng >= 4, have_y1 = ng >= 5, have_t3 = ng >= 6, have_y3 = ng >= 7;
    const int grad_grp = ng >= 3 ? 2 : 0;          // who evaluates a gradient outside the parallel phase
    const int b = __builtin_amdgcn_readfirstlane(b_);
    CoopCtx C;
    C.nwg = per; C.wgi = r_ - grp * nwg; C.Ppad = a.G * 32; C.epoch = 0u; C.spin_limit = a.coop_spin; C.fence = a.opt.coop_fence;
    C.bar = a.coop_bar + COOP_BAR_WORDS * b;
    C.pp = a.coop_pp + (size_t)b * coop_pp_stride(H, C.Ppad);
    C.ck = a.coop_ck + ((size_t)b * SPEC_CKS + (grp == 6 ? 3 : grp >= 2 && grp <= 4 ? grp - 2 : 0)) * a.P * (H + 1) * COOP_ROW;    // gradients run on groups 2..4 and 6 (or 0 when there are only two)
    Smem sm = carve(smem, H, m, 0, true);
    WaveW ww;
    load_weights(a, sm, ww, tid, Team::BNT);
    __syncthreads();
    if (b >= a.B) return;
    if ((int)blockIdx.x == a.opt.absent_wg) return;      // fault injection (SDEMPC_OPT_TEST_ABSENT_WG): a workgroup that never became resident
    load_common<Team>(a, sm, b, tid);
    __syncthreads();
    lane2_stage<Team>(a, sm, b, C.wgi, tid);
    const int nv = (N + 3) & ~3;
    float* ex = sm.cend;
    float *xn1 = ex, *xn2 = ex + nv, *y1 = ex + 2 * nv, *y2 = ex + 3 * nv, *xn3 = ex + 4 * nv, *y3 = ex + 5 * nv, *mred = ex + SPEC_XV * nv;
    float *xk = sm.v[0], *yk = sm.v[1], *xn = sm.v[2], *g = sm.v[3], *d1 = sm.v[4], *d2 = sm.v[5];
    SpecMot MK;
    {
        float* kc = mred + 48;
        MK.uref = kc; MK.ulo = kc + 8; MK.uhi = kc + 16; MK.shi = kc + 24; MK.slo = kc + 32; MK.dir = kc + 40; MK.ry = kc + 48; MK.rx = kc + 56;
        if (tid < 8) {
            kc[tid] = a.C.uref[tid]; kc[8 + tid] = a.C.ulo[tid]; kc[16 + tid] = a.C.uhi[tid]; kc[24 + tid] = a.C.slew_hi[tid]; kc[32 + tid] = a.C.slew_lo[tid];
            kc[40 + tid] = a.M.dir[tid]; kc[48 + tid] = a.M.ry[tid]; kc[56 + tid] = a.M.rx[tid];
        }
    }
    for (int e = tid; e < N; e += Team::NT) {
        int jj = e % m;
        float v = clampf(a.u[(size_t)b * N + e], a.C.ulo[jj], a.C.uhi[jj]);
        xk[e] = v; yk[e] = v;
    }
    // PH_RED: the particle sums of a gradient (H*nq totals + the cost total) are reduced ONCE, by the waves of the trial groups (two totals each),
    // published as tagged words and polled by every workgroup — every workgroup reducing everything itself took ~45 us per iteration
    enum { PH_INIT, PH_GRAD, PH_PAR, PH_SEQ, PH_RED, PH_FINAL, PH_DONE };
#ifndef SDEMPC_VAR_SPEC_CLK
#define SDEMPC_VAR_SPEC_CLK 0
#endif
    // [2][PS] published totals behind the per-particle slots, each a 64-bit word {value, tag}: the tag is the number of the reduction phase that
    // produced it, so a consumer that reads the pair in one aligned 64-bit load knows the value is the one it waits for — the reduction phase needs
    // no grid barrier of its own (a barrier costs 5 - 6 us of round trips across the XCDs: 5 % of an iteration), the hand-off is the datum itself
    unsigned long long* gtot_base = reinterpret_cast<unsigned long long*>(C.pp + coop_gtot_offset(H, C.Ppad));
    C.gtot = gtot_base;
    unsigned red_cnt = 0u, red_par = 0u, red_tag = 0u; int red_slot = 0;
    unsigned phc = 0u, par_tag = 0u, arr_cnt = 0u;      // phases so far (the tag of a phase's outputs is its number), tag of the last parallel phase, arrivals of this workgroup
    // Polling costs the polled (a store whose line hundreds of waves keep reading is acknowledged late, and gfx9 counts loads and stores in one in-order
    // counter: sdempc_lane2.inc.h, adjoint loop), and 55 us of it per iteration buy nothing. So a reducer SLEEPS (no memory traffic) until shortly before
    // the time the totals took to arrive in the iteration before, counted from the start of the parallel phase, and polls only then (measured, C2 single solve:
    // 19.9 ms; reducers polling from the moment they can 21.1 ms; a wave waking when ITS words were last seen arriving 20.8 ms — the early ones then poll the
    // TOTALS for the rest of the phase). That time contains the sleepers' own lateness: were the chain from the words to the totals (4.5 us at C2, 7 at C3) ever
    // longer than the lead, every iteration would start later than the one before. Hence the guard: the shortest time seen in this solve is remembered, and an
    // iteration whose predecessor took an eighth longer than that does not sleep (it polls from the start, re-measures, and the lateness is gone).
    uint64_t t_par = 0; unsigned d_tot = 0u, d_min = 0u;      // s_memrealtime at the start of the last parallel phase; ticks from there to the totals: last streamed reduction, shortest so far
    constexpr unsigned RED_LEAD = 900u;                 // start polling this many 10-ns ticks before the totals are due
    // tags of an earlier launch must not be taken for this one's: the instance's first workgroup clears the words (the first reduction phase lies
    // behind at least two grid barriers; since the streamed hand-off the host zeroes the whole workspace before the launch as well)
    if (r_ == 0)
        for (int i = tid; i < 2 * PS; i += Team::NT) __hip_atomic_store(gtot_base + i, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int phase = PH_INIT;
    float c_init = 0.0f, c_x = 0.0f, s = a.stepsize_in[b], gsq = 0.0f, sum_ls = 0.0f, sum_s = 0.0f, c_y = 0.0f, c_n = 0.0f;
    float gd_1 = 0.0f, gd_2 = 0.0f, rs_1 = 0.0f, rs_2 = 0.0f, cu_1 = 0.0f, cu_2 = 0.0f;      // of the parallel trials, known before they run
    float gd_3 = 0.0f, rs_3 = 0.0f, cu_3 = 0.0f;
    int k = 0, kr = 0, noimp = 0, nit = 0, nls_tot = 0, plain = 1, nls = 0, jsel = 0, jl = 0;
    unsigned par_cnt = 0u, par_spec = 0u;
    bool spec = false, two = false, three = false;
    const bool has_ls = a.A.maxls > 0;
#if SDEMPC_VAR_SPEC_CLK
    // diagnostic build (tools/spec_clock.py): 10 ns ticks by phase kind and section, workgroup (group 3, first) -> over the mean trajectory
    unsigned long long ck_acc[PH_DONE][3] = {}, ck_n[PH_DONE] = {};
    unsigned long long hw_last = 0, hw_hit = 0, hw_nhit = 0, hw_n = 0; int hw_slot = -1;      // work time of a parallel phase when this group's gradient was the one used / when not
    unsigned long long hk[6] = {}, hk_t = 0;      // sections of a polled head: control cost of yk | wait for the totals + g + trial points | barrier + arrive | ten reductions | candidate points
#define SPEC_HK(i) { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); hk[i] += t_ - hk_t; hk_t = t_; }
#else
#define SPEC_HK(i)
#endif
    while (phase != PH_DONE) {
#if SDEMPC_VAR_SPEC_CLK
        const int ck_phase = phase;
        const unsigned long long ck_t0 = __builtin_amdgcn_s_memrealtime();
#endif
        // ---- this workgroup's work item of the phase ----
        const float* iu = xk; int islot = SLOT_SEQ; bool igrad = false, imean = false, iact = false;
        unsigned par = C.epoch & 1u;
        const unsigned wtag = ++phc;
        if (phase == PH_RED) {
            constexpr int nq = M + 4;
            const unsigned long long* pw = spec_words(C, PS, red_par, red_slot);
            unsigned long long* gt = gtot_base + (size_t)(red_cnt & 1u) * PS;
            const unsigned long long tag = (unsigned long long)(red_cnt + 1u) << 32;
            // H * nq + 1 totals (the last item: the cost total), at most two per wave and those two polled side by side (a wave that reduced several of
            // them one after the other paid one cross-XCD round trip per item). The waves of the TRIAL groups reduce when they can take it all: their rollouts
            // end long before the gradient sweeps do (42 of 100 us at C2), so they have read the trial costs, know which candidate gradient the optimiser
            // will use and are already polling its words when the last of them is written — the gradient groups, which end the phase, only wait for the totals.
            const int items = H * nq + 1, nt = have_t3 ? 3 : 2, tw = nt * nwg * 4;
            const bool trial_grp = grp < 2 || grp == 5;
            const int gi = trial_grp ? (grp == 5 ? 2 : grp) : nt + (grp == 6 ? 3 : grp - 2);
            const int rank = __builtin_amdgcn_readfirstlane((gi * nwg + C.wgi) * 4 + wave);
            const int nredw = 2 * tw >= items ? tw : per * 4;
            if (rank < nredw && red_tag == par_tag && d_tot > RED_LEAD && d_tot <= d_min + (d_min >> 3)) {
                const uint64_t due = t_par + (uint64_t)(d_tot - RED_LEAD);
                while (__builtin_amdgcn_s_memrealtime() < due) __builtin_amdgcn_s_sleep(8);
            }
            const uint64_t t0r = __builtin_amdgcn_s_memrealtime();
            if (rank < nredw)
                for (int i0 = rank; i0 < items; i0 += 2 * nredw) {
                    const int i1 = i0 + nredw;
                    const int t0 = i0 / nq, q0 = i0 < H * nq ? t0 * 12 + (i0 - t0 * nq) : PS - 1;
                    const int t1 = i1 / nq, q1 = i1 < H * nq ? t1 * 12 + (i1 - t1 * nq) : PS - 1;
                    float s0, s1;
                    coop_total_tagged2(C, pw + (size_t)q0 * C.Ppad, i1 < items ? pw + (size_t)q1 * C.Ppad : nullptr, a.P, a.G, lane, red_tag, t0r, s0, s1);
                    if (lane == 0) {
                        __hip_atomic_store(gt + q0, tag | (unsigned long long)__float_as_uint(s0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (i1 < items) __hip_atomic_store(gt + q1, tag | (unsigned long long)__float_as_uint(s1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
        }
        else if (phase == PH_INIT) { iact = grp == 0; }
        else if (phase == PH_GRAD) { iact = grp == grad_grp; iu = yk; islot = SLOT_GRAD; igrad = true; }
        else if (phase == PH_SEQ) { iact = grp == 0; iu = xn; }
        else if (phase == PH_FINAL) { iact = grp == 0; imean = true; }
        else {   // PH_PAR
            par = par_cnt & 1u; par_spec = par; par_cnt += 1u; par_tag = wtag;
            t_par = __builtin_amdgcn_s_memrealtime();
            if (grp == 0) { iact = true; iu = xn1; islot = 0; }
            else if (grp == 1) { iact = two; iu = xn2; islot = 1; }
            else if (grp == 2) { iact = spec && two; iu = y2; igrad = true; islot = 3; }     // when there is one trial only, y1 takes this group
            else if (grp == 3) { iact = spec; iu = xk; igrad = true; islot = 4; }
            else if (grp == 4) { iact = spec; iu = y1; igrad = true; islot = 2; }
            else if (grp == 5) { iact = three; iu = xn3; islot = SLOT_T3; }
            else { iact = spec && three; iu = y3; igrad = true; islot = SLOT_Y3; }
            if (grp == 2 && spec && !two) { iact = true; iu = y1; igrad = true; islot = 2; }
        }
        // The optimiser's scalars are identical in every lane (and made provably so where they are produced, so that every branch of
        // the state machine is uniform for the compiler); pinning them to SGPRs across the particle work keeps them out of the
        // VGPR allocator's way (with 343 registers in use LLVM saved two of them to AGPRs under the partial EXEC mask of a preceding
        // divergent block and restored them under the full mask: wrong telemetry in lane 0; SGPR spills are whole-wave and safe)
        c_init = uni_f(c_init); c_x = uni_f(c_x); s = uni_f(s); gsq = uni_f(gsq); sum_ls = uni_f(sum_ls); sum_s = uni_f(sum_s);
        c_y = uni_f(c_y); c_n = uni_f(c_n);
        gd_1 = uni_f(gd_1); gd_2 = uni_f(gd_2); rs_1 = uni_f(rs_1); rs_2 = uni_f(rs_2); cu_1 = uni_f(cu_1); cu_2 = uni_f(cu_2);
        gd_3 = uni_f(gd_3); rs_3 = uni_f(rs_3); cu_3 = uni_f(cu_3);
        if (iact) {      // the only call site of the particle work
            __syncthreads();
            lane2_prepass<Team>(a, sm, iu, tid);
            __syncthreads();
            const int p = C.wgi * 4 + wave;
            if (p < a.P) {
                const Lane2IO io = lane_io_slot(a, C, b, p, par, islot, wtag);
                const Lane2Lds L = lane2_lds(a, sm, wave);
                if (igrad) lane2_grad<M, true>(a, sm, L, io, lane);
                else if (imean) lane2_rollout<true, true>(a, sm, L, io, lane);
                else lane2_rollout<false, true>(a, sm, L, io, lane);
            }
        }
#if SDEMPC_VAR_SPEC_CLK
        const unsigned long long ck_t1 = __builtin_amdgcn_s_memrealtime();
#endif
        if (phase != PH_RED && phase != PH_PAR) coop_barrier(C, tid);      // (the outputs of a parallel phase and the totals of a reduction phase are their own hand-off)
#if SDEMPC_VAR_SPEC_CLK
        const unsigned long long ck_t2 = __builtin_amdgcn_s_memrealtime();
#endif
        // ---- the optimiser (SPEC.md §8), advanced as far as the data of this phase allows ----
        bool head = false, tail = false, fin = false, head_poll = false;
        if (phase == PH_INIT) {
            c_init = uni_f(spec_cost(a, sm, C, tid, par, xk, SLOT_SEQ, wtag));
            c_x = c_init;
            phase = a.A.max_iter > 0 ? PH_GRAD : PH_FINAL;
        } else if (phase == PH_GRAD) {
            red_slot = SLOT_GRAD; red_par = par; red_tag = wtag;
            if constexpr (DIRECT) fin = true;
            else phase = PH_RED;
        } else if (phase == PH_RED) {
            red_cnt += 1u;
            head = true; head_poll = true;      // the totals are polled in the head, by the thread that needs them
        } else if (phase == PH_PAR) {
            // the trial costs at once: particle totals by waves 0 .. 2 (polled: no barrier behind this phase), control costs and g.d known since the
            // head of the iteration; the last wave makes sure every workgroup has left the iteration before (coop_arrive)
            const int PSs = part_stride(H);
            __syncthreads();
            {
                const uint64_t t0w = __builtin_amdgcn_s_memrealtime();
                if (wave < (three ? 3 : two ? 2 : 1)) {
                    const float t0 = coop_total_tagged(C, spec_words(C, PSs, par, wave == 2 ? SLOT_T3 : wave) + (size_t)(PSs - 1) * C.Ppad, a.P, a.G, lane, wtag, t0w);
                    if (lane == 0) sm.red[12 + wave] = t0;
                } else if (tid == 255) coop_arrived_wait(C, arr_cnt * (unsigned)per, t0w);
            }
            __syncthreads();
            c_n = uni_f(FMA(sm.red[12], a.invP, cu_1));
            nls = 1; jsel = 1;
            bool done = !has_ls;
            if (has_ls) {
                if (c_n <= FMA(a.A.coef, gd_1, c_y)) done = true;
                else if (0 < a.A.maxls - 1) s = s * a.A.dec;
            }
            if (!done && two) {
                c_n = uni_f(FMA(sm.red[13], a.invP, cu_2));
                nls = 2; jsel = 2;
                if (c_n <= FMA(a.A.coef, gd_2, c_y)) done = true;
                else if (1 < a.A.maxls - 1) s = s * a.A.dec;
            }
            if (!done && three) {
                c_n = uni_f(FMA(sm.red[14], a.invP, cu_3));
                nls = 3; jsel = 3;
                if (c_n <= FMA(a.A.coef, gd_3, c_y)) done = true;
                else if (2 < a.A.maxls - 1) s = s * a.A.dec;
            }
            {
                const float* xj = jsel == 3 ? xn3 : jsel == 2 ? xn2 : xn1;
                for (int e = tid; e < N; e += Team::NT) xn[e] = xj[e];
            }
            const int npar = three ? 3 : 2;
            if (!done && npar < a.A.maxls) {     // further trials one at a time
                jl = npar;
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
            c_n = uni_f(spec_cost(a, sm, C, tid, par, xn, SLOT_SEQ, wtag));
            const float gd = uni_f(block_dot<Team>(sm, g, d1, N, tid));
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
                const unsigned long long* pw = spec_words(C, PS, par, SLOT_SEQ);
                float* xmean_out = a.xmean + (size_t)b * (H + 1) * NX;
#if !SDEMPC_VAR_SPEC_CLK
                const uint64_t t0w = __builtin_amdgcn_s_memrealtime();
                for (int q = wave; q < (H + 1) * NX; q += 8) {
                    float s0, s1;
                    coop_total_tagged2(C, pw + (size_t)q * C.Ppad, q + 4 < (H + 1) * NX ? pw + (size_t)(q + 4) * C.Ppad : nullptr, a.P, a.G, lane, wtag, t0w, s0, s1);
                    if (lane == 0) { xmean_out[q] = s0 * a.invP; if (q + 4 < (H + 1) * NX) xmean_out[q + 4] = s1 * a.invP; }
                }
#endif
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
                float rs = jsel == 1 ? rs_1 : jsel == 2 ? rs_2 : rs_3;          // restart test of a parallel trial: known since the head
                if (jsel == 0) {
                    for (int e = tid; e < N; e += Team::NT) { d1[e] = yk[e] - xn[e]; d2[e] = xn[e] - xk[e]; }
                    rs = uni_f(block_dot<Team>(sm, d1, d2, N, tid));
                }
                if (rs > 0.0f) {
                    kr = 0; plain = 1;
                    for (int e = tid; e < N; e += Team::NT) { yk[e] = xn[e]; xk[e] = xn[e]; }
                } else {
                    float bt = a.beta[kr];
                    for (int e = tid; e < N; e += Team::NT) { int jj = e % m; const float dd = xn[e] - xk[e]; yk[e] = clampf(FMA(bt, dd, xn[e]), a.C.ulo[jj], a.C.uhi[jj]); xk[e] = xn[e]; }
                    kr = kr + 1; plain = 0;
                }
                c_x = c_n; noimp = 0;
                // the new yk is exactly y_jsel: was its gradient among the speculated ones?
                if (spec && jsel == 2 && have_y2) hit_slot = 3;
                if (spec && jsel == 3 && have_y3) hit_slot = SLOT_Y3;
                if (spec && jsel == 1 && (have_y1 || (have_y2 && !two))) hit_slot = 2;
            } else {
                if (!plain) stop = 0;
                kr = 0; plain = 1;
                for (int e = tid; e < N; e += Team::NT) yk[e] = xk[e];
                noimp = noimp + 1;
                if (spec && have_xk) hit_slot = 4;               // the new yk is xk, unchanged since the parallel phase
            }
#if SDEMPC_VAR_SPEC_CLK
            if (hw_slot >= 0) { if (hit_slot == hw_slot) { hw_hit += hw_last; hw_n += 1; } else hw_nhit += hw_last; }
#endif
            if (noimp >= a.A.max_noimp) stop = 1;
            k += 1;
            if (stop || k >= a.A.max_iter) phase = PH_FINAL;
            else if (hit_slot >= 0) {
                red_slot = hit_slot; red_par = par_spec; red_tag = par_tag;
                if constexpr (DIRECT) fin = true;
                else phase = PH_RED;
            }
            else phase = PH_GRAD;
        }
        if constexpr (DIRECT) {
            // (c_y, g) straight from the particle's outputs in (red_par, red_slot): v + 0 is what the butterflies and the slot order of SPEC.md §6.1 make of
            // one value and zeros, so no reduction phase and no barrier for it — the head polls the particle's words as it polls the totals otherwise
            if (fin) { head = true; head_poll = true; }
        }
        if (head) {      // start of iteration k with (c_y, g) in hand
            float sn = s;
            if (has_ls) {
                if (k > 0 && a.A.reset_inc) sn = sn * a.A.inc;
                if (sn > a.A.smax) sn = a.A.smax;
            } else {
                sn = a.A.stepsize;
            }
            const float s1 = sn, s2 = sn * a.A.dec, s3 = s2 * a.A.dec;
            if (head_poll) {
                // (c_y, g) from the published totals: every thread polls the five sums its element of the gradient is made of (and the cost total) itself and
                // goes on to the trial points of that element — no staging of the totals in LDS, no barrier between totals, gradient and trial points; the
                // control cost of yk needs no totals and is reduced BEFORE the wait (its barriers also order this iteration's writes of g / xn_j behind the last reads)
                const unsigned long long* gt = DIRECT ? spec_words(C, PS, red_par, red_slot) : gtot_base + (size_t)((red_cnt - 1u) & 1u) * PS;
                const int ws = DIRECT ? C.Ppad : 1;                        // words from one quantity to the next (P == 1: the particle's own outputs, particle-minor rows)
                const unsigned ptag = DIRECT ? red_tag : red_cnt;
#if SDEMPC_VAR_SPEC_CLK
                hk_t = __builtin_amdgcn_s_memrealtime();
#endif
                const float cu = block_ucost<Team>(a, sm, yk, tid);
                SPEC_HK(0)
                const uint64_t t0w = __builtin_amdgcn_s_memrealtime();
                float ctot = 0.0f;
                for (int e0 = 0; e0 < N; e0 += 2 * Team::NT) {      // two elements per thread and polling round (N > 256: a second round would be a second round trip)
                    const int ea = e0 + tid, eb = ea + Team::NT;
                    const bool ha = ea < N, hb = eb < N;
                    const int ta = ha ? ea / m : 0, ja = ha ? ea - ta * m : 0, tb = hb ? eb / m : 0, jb = hb ? eb - tb * m : 0;
                    const unsigned long long *ga = gt + (size_t)(ta * 12) * ws, *gb = gt + (size_t)(tb * 12) * ws;
                    const unsigned long long* const pw[11] = {ga + ja * ws, ga + M * ws, ga + (M + 1) * ws, ga + (M + 2) * ws, ga + (M + 3) * ws,
                                                              gb + jb * ws, gb + M * ws, gb + (M + 1) * ws, gb + (M + 2) * ws, gb + (M + 3) * ws, gt + (size_t)(PS - 1) * ws};
                    const bool have[11] = {ha, ha, ha, ha, ha, hb, hb, hb, hb, hb, e0 == 0};
                    float v[11];
                    tagged_wait_n<11>(C, pw, have, ptag, t0w, v);
                    if constexpr (DIRECT) {
#pragma unroll
                        for (int i = 0; i < 11; ++i) v[i] = v[i] + 0.0f;
                    }
                    if (e0 == 0) ctot = v[10];
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const int e = half ? eb : ea, t = half ? tb : ta, jj = half ? jb : ja;
                        if (half ? hb : ha) {
                            const float ge = assemble_elem_k(a, sm, MK, yk, e, t, jj, m, v + 5 * half);
                            g[e] = ge;
                            xn1[e] = clampf(FMA(-s1, ge, yk[e]), MK.ulo[jj], MK.uhi[jj]);
                            xn2[e] = clampf(FMA(-s2, ge, yk[e]), MK.ulo[jj], MK.uhi[jj]);
                            xn3[e] = clampf(FMA(-s3, ge, yk[e]), MK.ulo[jj], MK.uhi[jj]);
                        }
                    }
                }
                if (red_tag == par_tag) { d_tot = (unsigned)(__builtin_amdgcn_s_memrealtime() - t_par); if (d_min == 0u || d_tot < d_min) d_min = d_tot; }
                c_y = uni_f(FMA(ctot, a.invP, cu));
                SPEC_HK(1)
            } else {
                __syncthreads();
                for (int e = tid; e < N; e += Team::NT) {
                    int jj = e % m;
                    xn1[e] = clampf(FMA(-s1, g[e], yk[e]), a.C.ulo[jj], a.C.uhi[jj]);
                    xn2[e] = clampf(FMA(-s2, g[e], yk[e]), a.C.ulo[jj], a.C.uhi[jj]);
                    xn3[e] = clampf(FMA(-s3, g[e], yk[e]), a.C.ulo[jj], a.C.uhi[jj]);
                }
            }
            __syncthreads();
            arr_cnt += 1u; coop_arrive(C, tid);      // everything the iteration before produced has been read by this workgroup
            SPEC_HK(2)
            // everything the optimiser will ask about the two parallel trials except their particle costs, in one reduction:
            // |g|^2; g.(xn_j - yk) of the Armijo tests; the restart tests (yk - xn_j).(xn_j - xk); the control costs of xn_j
            float r7[10];
            spec_reduce_n<10>(mred, N, tid, r7, [&](int e, float (&acc)[10]) {
                const float ge = g[e], ye = yk[e], xe = xk[e], x1 = xn1[e], x2 = xn2[e], x3 = xn3[e];
                acc[0] = FMA(ge, ge, acc[0]);
                acc[1] = FMA(ge, x1 - ye, acc[1]);
                acc[2] = FMA(ge, x2 - ye, acc[2]);
                acc[3] = FMA(ye - x1, x1 - xe, acc[3]);
                acc[4] = FMA(ye - x2, x2 - xe, acc[4]);
                const int t = e / m, j = e - t * m;
                acc[5] = FMA(ucost_elem_k(a, sm, MK, xn1, e, t, j, m), 1.0f, acc[5]);
                acc[6] = FMA(ucost_elem_k(a, sm, MK, xn2, e, t, j, m), 1.0f, acc[6]);
                acc[7] = FMA(ge, x3 - ye, acc[7]);
                acc[8] = FMA(ye - x3, x3 - xe, acc[8]);
                acc[9] = FMA(ucost_elem_k(a, sm, MK, xn3, e, t, j, m), 1.0f, acc[9]);
            });
            gsq = uni_f(r7[0]);
            SPEC_HK(3)
            if (!(gsq < __builtin_inff())) { phase = PH_FINAL; continue; }      // (s keeps the value of the last completed iteration)
            s = sn;
            gd_1 = uni_f(r7[1]); gd_2 = uni_f(r7[2]); rs_1 = uni_f(r7[3]); rs_2 = uni_f(r7[4]); cu_1 = uni_f(r7[5]); cu_2 = uni_f(r7[6]);
            gd_3 = uni_f(r7[7]); rs_3 = uni_f(r7[8]); cu_3 = uni_f(r7[9]);
            two = (has_ls ? a.A.maxls : 1) > 1;
            three = have_t3 && has_ls && a.A.maxls > 2;          // third trial side by side with the first two
            spec = (k + 1 < a.A.max_iter);
            // where the optimiser moves if it ends on trial j with an improvement (the expressions of the tail above)
            const float bt = a.beta[kr];
            for (int e = tid; e < N; e += Team::NT) {
                int jj = e % m;
                const float x1 = xn1[e], x2 = xn2[e], x3 = xn3[e], xe = xk[e];
                y1[e] = (rs_1 > 0.0f) ? x1 : clampf(FMA(bt, x1 - xe, x1), MK.ulo[jj], MK.uhi[jj]);
                if (two) y2[e] = (rs_2 > 0.0f) ? x2 : clampf(FMA(bt, x2 - xe, x2), MK.ulo[jj], MK.uhi[jj]);
                if (three) y3[e] = (rs_3 > 0.0f) ? x3 : clampf(FMA(bt, x3 - xe, x3), MK.ulo[jj], MK.uhi[jj]);
            }
            __syncthreads();
            SPEC_HK(4)
            phase = PH_PAR;
        }
#if SDEMPC_VAR_SPEC_CLK
        {
            const unsigned long long ck_t3 = __builtin_amdgcn_s_memrealtime();
            if (ck_phase == PH_PAR) { hw_last = ck_t1 - ck_t0; hw_slot = grp >= 2 && grp != 5 ? (grp == 2 ? 3 : grp == 3 ? 4 : grp == 4 ? 2 : SLOT_Y3) : -1; }
            ck_acc[ck_phase][0] += ck_t1 - ck_t0; ck_acc[ck_phase][1] += ck_t2 - ck_t1; ck_acc[ck_phase][2] += ck_t3 - ck_t2; ck_n[ck_phase] += 1;
        }
#endif
    }
#if SDEMPC_VAR_SPEC_CLK
    if (tid == 0 && r_ < 250) {      // every workgroup: mean work time of its parallel phases (us) and where it ran (XCC id * 100 + group)
        float* o = a.xmean + (size_t)b * (H + 1) * NX + 100;
        o[2 * r_] = (float)ck_acc[PH_PAR][0] * 0.01f / (float)(ck_n[PH_PAR] ? ck_n[PH_PAR] : 1);
        o[2 * r_ + 1] = (float)((__builtin_amdgcn_s_getreg(63508) & 15) * 100 + grp);      // HW_REG_XCC_ID
    }
    if (grp >= 2 && grp != 5 && C.wgi == 0 && tid == 0 && ng == 7) {      // per gradient group: mean work time (us) when hit / when not, and how often it was hit
        float* o = a.xmean + (size_t)b * (H + 1) * NX + 560 + 4 * (grp == 6 ? 3 : grp - 2);
        const unsigned long long nn = ck_n[PH_PAR] - hw_n;
        o[0] = hw_n ? (float)hw_hit * 0.01f / (float)hw_n : 0.0f; o[1] = nn ? (float)hw_nhit * 0.01f / (float)nn : 0.0f; o[2] = (float)hw_n; o[3] = (float)nn;
    }
    if (grp == (ng > 3 ? 3 : 0) && C.wgi == 0 && tid == 0) {
        float* o = a.xmean + (size_t)b * (H + 1) * NX + 64;
        for (int ph = 0; ph < PH_DONE; ++ph) { for (int k = 0; k < 3; ++k) o[ph * 4 + k] = (float)ck_acc[ph][k]; o[ph * 4 + 3] = (float)ck_n[ph]; }
        for (int i = 0; i < 5; ++i) o[PH_DONE * 4 + i] = (float)hk[i];
    }
#endif
}

