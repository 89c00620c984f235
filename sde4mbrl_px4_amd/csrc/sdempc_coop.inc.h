// sdempc_coop.inc.h — cooperative latency path: one instance over several workgroups, grid barrier, particle reductions
// Fragment of sdempc_kernels.hip: included inside namespace sdempc::{exact|fastm} (it is compiled twice, see there); not a
// stand-alone header.
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
    unsigned* bar;          // this instance's COOP_BAR_WORDS words (zeroed by the host before the launch): [0] grid-barrier counter, [1] error flag, [2] arrivals of the streamed hand-off
    unsigned epoch, spin_limit;   // spin_limit: ticks of the 100 MHz s_memrealtime clock one barrier may wait (KArgs::coop_spin)
    int fence;              // KArgs::opt.coop_fence: agent-scope release / acquire fences around the barrier
    float* pp;              // [2][PS][Ppad] per-particle outputs, double-buffered by rollout parity
    unsigned long long* gtot;   // [2][PS] published totals {value, tag} of the distributed reductions (behind the per-particle slots of the instance)
    float* ck;              // [P][H+1][COOP_ROW] checkpoint rows
};

// Grid barrier of the cooperative layouts. Memory ordering of the hand-off ("grid barrier"): every handed-off word is
// written with an agent-scope (sc1, write-through) store and read with an agent-scope (sc1, L1-bypassing) global load; every storing
// wave drains its stores (s_waitcnt vmcnt(0)) before the workgroup barrier behind which ONE lane adds to the counter; the polling
// lane's workgroup loads only after the workgroup barrier that follows its poll. That is the fence-free form listed as valid (and
// measured on gfx950 / ROCm 7.2, not an architectural guarantee) in MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement &
// inter-workgroup visibility", Consumer conditions (1)-(4), first table row. C.fence adds the language-level release / acquire pair
// of the HIP memory model on top (buffer_wbl2 sc1 / buffer_inv sc1): SDEMPC_OPT_COOP_FENCE, same results, measured cost in DESIGN.md.
// The wait is bounded in TIME (s_memrealtime, 100 MHz): a grid that is not fully resident gives up instead of hanging the GPU.
DI void coop_barrier(CoopCtx& C, int tid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's (sc1) stores of the handed-off values have completed
    __syncthreads();
    C.epoch += 1;
    if (tid == 0) {
        if (C.fence) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the compiler may drop the wait behind buffer_wbl2 (ROCm 7.2): keep it explicit
        }
        __hip_atomic_fetch_add(C.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = C.epoch * (unsigned)C.nwg;
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        // counter and error flag are the two halves of one aligned 64-bit word: ONE load per poll (two dependent round trips across the XCDs
        // per poll made the poll period, and with it the mean time to notice the last arrival, twice as long)
        const unsigned long long* both = reinterpret_cast<const unsigned long long*>(C.bar);
        for (;;) {
            const unsigned long long w = __hip_atomic_load(both, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)w >= target) break;                                   // everyone arrived
            if ((unsigned)(w >> 32) != 0u) { C.spin_limit = 0; break; }          // another workgroup gave up: no further waiting in this kernel
            if (__builtin_amdgcn_s_memrealtime() - t0 >= (uint64_t)C.spin_limit) { __hip_atomic_store(C.bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); C.spin_limit = 0; break; }
        }
        if (C.fence) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the invalidate completes asynchronously: hold the workgroup barrier until it has
        }
    }
    __syncthreads();
}

// Wait of a tagged hand-off (a 64-bit word {value, tag}; the datum is its own flag: sdempc_spec.inc.h, coop_cost_grad): until the word carries
// `tag`, bounded like a grid barrier — and, like coop_barrier, over as soon as ANY workgroup of the grid has given up: the error flag is looked at
// on every eighth unsuccessful poll (a wait of a healthy launch ends within a handful of polls, so it never pays for that load), and a thread that
// saw the flag or ran out of budget itself stops waiting for the rest of the kernel (C.spin_limit = 0: every later wait of this thread ends on its
// first unsuccessful poll). Without it every reduction phase behind a missing producer waited a full budget: up to 200 x 100 ms per C2 solve
// instead of one budget (the results are invalid from the first give-up on: the telemetry is poisoned with NaN, sdempc_solve_status reports it).
DI float tagged_wait(CoopCtx& C, const unsigned long long* p, unsigned tag, uint64_t t0w) {
    unsigned long long v;
    unsigned polls = 0;
    while ((unsigned)((v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != tag) {
        if (__builtin_amdgcn_s_memrealtime() - t0w >= (uint64_t)C.spin_limit) {
            __hip_atomic_store(C.bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            C.spin_limit = 0;
            break;
        }
        if ((++polls & 7u) == 0u && __hip_atomic_load(C.bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { C.spin_limit = 0; break; }
    }
    return __uint_as_float((unsigned)v);
}

// ---- streamed hand-off (speculative kernel): per-particle outputs are 64-bit words {value, tag} too, polled instead of waited for behind a grid barrier ----
// N tagged words per lane, polled side by side (one round trip per poll for all of them); an absent word (have == false) counts as
// arrived with the value + 0.0. Bounded like tagged_wait.
template <int N>
DI void tagged_wait_n(CoopCtx& C, const unsigned long long* const (&p)[N], const bool (&have)[N], unsigned tag, uint64_t t0w, float (&v)[N]) {
    unsigned long long w[N];
#pragma unroll
    for (int i = 0; i < N; ++i) w[i] = (unsigned long long)tag << 32;
    unsigned polls = 0;
    for (;;) {
#pragma unroll
        for (int i = 0; i < N; ++i) if (have[i]) w[i] = __hip_atomic_load(p[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool all = true;
#pragma unroll
        for (int i = 0; i < N; ++i) all = all && (unsigned)(w[i] >> 32) == tag;
        if (all) break;
        if (__builtin_amdgcn_s_memrealtime() - t0w >= (uint64_t)C.spin_limit) {
            __hip_atomic_store(C.bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            C.spin_limit = 0;
            break;
        }
        if ((++polls & 7u) == 0u && __hip_atomic_load(C.bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { C.spin_limit = 0; break; }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = __uint_as_float((unsigned)w[i]);
}
// coop_total of up to TWO quantities at once from tagged words (pq1 may be null, wave-uniformly): the same butterflies, slots and order. Four words per lane and
// polling round: two quantities x one chunk of four particle groups, or one quantity x two chunks (P > 128 — chunk after chunk cost one more round trip each)
DI void coop_total_tagged2(CoopCtx& C, const unsigned long long* pq0, const unsigned long long* pq1, int P, int G, int lane, unsigned tag, uint64_t t0w, float& s0, float& s1) {
    const int hh = lane >> 5, j = lane & 31;
    const bool second = pq1 != nullptr;
    float Sa0 = 0.0f, Sb0 = 0.0f, Sa1 = 0.0f, Sb1 = 0.0f;
    if (second) {
        for (int g0 = 0; g0 < G; g0 += 4) {
            const int pa = 32 * (g0 + hh) + j, pb = 32 * (g0 + 2 + hh) + j;
            const bool ha = (g0 + hh < G) && pa < P, hb = (g0 + 2 + hh < G) && pb < P;
            const unsigned long long* const pw[4] = {pq0 + pa, pq0 + pb, pq1 + pa, pq1 + pb};
            const bool have[4] = {ha, hb, ha, hb};
            float v[4];
            tagged_wait_n<4>(C, pw, have, tag, t0w, v);
            Sa0 = Sa0 + group_bfly32(v[0]);
            Sa1 = Sa1 + group_bfly32(v[2]);
            if (g0 + 2 < G) { Sb0 = Sb0 + group_bfly32(v[1]); Sb1 = Sb1 + group_bfly32(v[3]); }
        }
    } else {
        for (int g0 = 0; g0 < G; g0 += 8) {
            const int pa = 32 * (g0 + hh) + j, pb = 32 * (g0 + 2 + hh) + j, pc = 32 * (g0 + 4 + hh) + j, pd = 32 * (g0 + 6 + hh) + j;
            const unsigned long long* const pw[4] = {pq0 + pa, pq0 + pb, pq0 + pc, pq0 + pd};
            const bool have[4] = {(g0 + hh < G) && pa < P, (g0 + 2 + hh < G) && pb < P, (g0 + 4 + hh < G) && pc < P, (g0 + 6 + hh < G) && pd < P};
            float v[4];
            tagged_wait_n<4>(C, pw, have, tag, t0w, v);
            Sa0 = Sa0 + group_bfly32(v[0]);
            if (g0 + 2 < G) Sb0 = Sb0 + group_bfly32(v[1]);
            if (g0 + 4 < G) {
                Sa0 = Sa0 + group_bfly32(v[2]);
                if (g0 + 6 < G) Sb0 = Sb0 + group_bfly32(v[3]);
            }
        }
    }
    s0 = ((readlane_f(Sa0, 0) + readlane_f(Sa0, 32)) + readlane_f(Sb0, 0)) + readlane_f(Sb0, 32);
    s1 = ((readlane_f(Sa1, 0) + readlane_f(Sa1, 32)) + readlane_f(Sb1, 0)) + readlane_f(Sb1, 32);
}
DI float coop_total_tagged(CoopCtx& C, const unsigned long long* pq, int P, int G, int lane, unsigned tag, uint64_t t0w) {
    float s0, s1;
    coop_total_tagged2(C, pq, nullptr, P, G, lane, tag, t0w, s0, s1);
    return s0;
}
// Arrivals: what keeps a slow workgroup's READS of iteration k ahead of a fast one's WRITES of iteration k + 2 into the same (double-buffered) words now
// that no grid barrier separates them. A workgroup arrives (one fire-and-forget atomic) when it has read everything iteration k produced; before it leaves
// the parallel phase of iteration k + 1 it makes sure everyone has — a whole phase later, so the word it polls has long had its value (one load, in the
// shadow of the trial-cost polls of the other waves).
// (The tagged words themselves need no fence under any memory model: value and flag are ONE atomic location. The arrival counter orders accesses to
// OTHER locations — reads before the add, writes behind the wait — so SDEMPC_OPT_COOP_FENCE puts its release / acquire pair here as well.)
DI void coop_arrive(CoopCtx& C, int tid) {
    // The compiler's wait-count model still carries the polling loops' loads as "maybe outstanding" (their timeout exits leave without waiting), and the first
    // write to one of their registers behind the atomic then gets a vmcnt(0) — which on gfx9 also waits for the ATOMIC (one counter, in order): a 2 us round
    // trip across the XCDs in one wave, with the other three waiting for it at the next barrier. Those loads returned long ago: a wait the compiler can see,
    // in front of the atomic, costs nothing and clears its books, so that nothing behind the atomic waits for the counter until real loads are due.
    __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0), the other counters untouched (gfx9 encoding)
    if (tid == 0) {
        if (C.fence) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __hip_atomic_fetch_add(C.bar + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
DI void coop_arrived_wait(CoopCtx& C, unsigned target, uint64_t t0w) {
    unsigned polls = 0;
    while (__hip_atomic_load(C.bar + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (__builtin_amdgcn_s_memrealtime() - t0w >= (uint64_t)C.spin_limit) {
            __hip_atomic_store(C.bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            C.spin_limit = 0;
            break;
        }
        if ((++polls & 7u) == 0u && __hip_atomic_load(C.bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { C.spin_limit = 0; break; }
    }
    if (C.fence) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
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

DI Lane2IO lane_io_coop(const KArgs& a, const CoopCtx& C, int b, int p) {
    const int H = a.H;
    Lane2IO io;
    io.x0 = a.x0 + (size_t)b * NX;
    io.ck = C.ck + (size_t)p * (H + 1) * COOP_ROW;
    io.out = C.pp + (size_t)(C.epoch & 1u) * part_stride(H) * C.Ppad + p; io.os = C.Ppad; io.tag = 0u;
    return io;
}

template <class Team>
DI float coop_rollout(const KArgs& a, const Smem& sm, const LaneW& W, CoopCtx& C, const float* u, int b, int tid, float* xmean_out) {
    b = opaque_s(b); tid = opaque_v(tid);
    const int H = a.H, P = a.P, G = a.G, lane = tid & 63, wave = tid >> 6, PS = part_stride(H);
    const bool want_mean = xmean_out != nullptr;
    Team::sync();
    lane2_prepass<Team>(a, sm, u, tid);
    float cu = block_ucost<Team>(a, sm, u, tid);
    const int p = C.wgi * 4 + wave;
    const float* pbuf = C.pp + (size_t)(C.epoch & 1u) * PS * C.Ppad;
    if (p < P) {
        const Lane2IO io = lane_io_coop(a, C, b, p);
        const Lane2Lds L = lane2_lds(a, sm, wave);
        if (want_mean) lane2_rollout<true>(a, sm, L, io, lane);
        else lane2_rollout<false>(a, sm, L, io, lane);
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
    lane2_prepass<Team>(a, sm, y, tid);
    float cu = block_ucost<Team>(a, sm, y, tid);
    const int p = C.wgi * 4 + wave;
    const float* pbuf = C.pp + (size_t)(C.epoch & 1u) * PS * C.Ppad;
    if (p < P) {
        const Lane2IO io = lane_io_coop(a, C, b, p);
        lane2_grad<M>(a, sm, lane2_lds(a, sm, wave), io, lane);
    }
    coop_barrier(C, tid);
    // particle sums of the nq adjoint outputs of every step (+ the cost total): one total per wave, spread over every wave of the instance's
    // workgroups, published as 64-bit words {value, tag = number of this barrier} and read back by everyone — no second barrier: the datum is
    // the hand-off (sdempc_spec.inc.h). Every workgroup reducing all H * nq totals itself cost 18 % of a C5 gradient evaluation (H = 200, 32 groups).
    {
        unsigned long long* gt = C.gtot + (size_t)(C.epoch & 1u) * PS;
        const unsigned long long tag = (unsigned long long)C.epoch << 32;
        for (int item = C.wgi * 4 + wave; item <= H * nq; item += C.nwg * 4) {
            const int t = item / nq, q = item < H * nq ? t * 12 + (item - t * nq) : PS - 1;
            const float sv = coop_total(pbuf + (size_t)q * C.Ppad, P, G, lane);
            if (lane == 0) __hip_atomic_store(gt + q, tag | (unsigned long long)__float_as_uint(sv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const uint64_t t0w = __builtin_amdgcn_s_memrealtime();
        auto tagged = [&](const unsigned long long* pw) { return tagged_wait(C, pw, C.epoch, t0w); };
        for (int q = tid; q < H * 12; q += Team::NT)
            if ((q % 12) < nq) sm.tot[q] = tagged(gt + q);
        if (tid == 0) sm.red[12] = tagged(gt + PS - 1);
    }
    Team::sync();
    const float tot = sm.red[12];
    assemble_gradient<Team, M>(a, sm, y, gout, tid, [&](int q) { return sm.tot[q]; });
    Team::sync();
    return FMA(tot, a.invP, cu);
}

