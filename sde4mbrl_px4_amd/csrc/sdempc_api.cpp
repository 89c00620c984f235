// sdempc_api.cpp — C ABI (include/sdempc.h) over the HIP kernels in sdempc_kernels.hip.
//
// Reference-side counterpart: the solver objects built in SDEControlROS.load_single_mpc
// (sde4mbrl_px4/mpc_controller/sde_control.py:681-721) and used by mpc_process_fn (:328-450).
// Host logic only: argument checking, table construction (time grid, discount, momentum), lazy HIP
// context creation (the reference builds its solvers before fork(), sde_control.py:69-75), staging
// buffers for the host-pointer entry points and layout conversion. No arithmetic of the hot path
// runs on the host and there is no CPU fallback.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <cxxabi.h>

#include <memory>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/sdempc.h"
#include "sdempc_kernels.h"

using namespace sdempc;
static_assert(blob::FLOATS == SDEMPC_BLOB_FLOATS, "blob layout: sdempc_kernels.h and include/sdempc.h");

namespace {
std::string g_create_error;

// SPEC.md §9: round toward zero to the nearest IEEE binary16 value; overflow saturates at 65504.
float f16_rtz_host(float x) {
    uint32_t u; memcpy(&u, &x, 4);
    const uint32_t sign = u & 0x80000000u, mag = u & 0x7FFFFFFFu;
    if (mag >= 0x7F800000u) return x;
    const int e = (int)(mag >> 23) - 127;
    float r;
    if (e > 15) r = 65504.0f;
    else if (e >= -14) { uint32_t t = mag & ~0x1FFFu; memcpy(&r, &t, 4); }
    else { float a; memcpy(&a, &mag, 4); r = floorf(a * 16777216.0f) / 16777216.0f; }
    uint32_t ru; memcpy(&ru, &r, 4); ru |= sign; memcpy(&r, &ru, 4);
    return r;
}

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};
}  // namespace

struct sdempc_handle {
    sdempc_cfg cfg;
    std::vector<float> time_steps;
    std::vector<float> blob_f;  // float payload of the model blob
    int m = 0, H = 0, P = 0, G = 0, max_batch = 0;
    int device = 0;
    bool dev_ready = false;
    mutable std::string err;
    KArgs base;
    unsigned ticket_total = 0;   // running value of the device ticket word (KArgs::ticket_host points here; sdempc_kernels.hip, launch_persistent)
    int ws_rows = 0;             // rows (instances or team slots) the trajectory / checkpoint / partial-sum / control-table workspaces hold
    int traj_batch = 0;          // instances whose particle x horizon tensor the trajectory workspace holds (last rollout with store_traj); 0: none —
                                 // any later launch that writes the workspace (gradient, solve) or a reallocation of it resets this
    bool last_ticketed = false;  // the last solve launch handed its instances out by ticket: sdempc_solve_status compares the word with the mirror
    // host tables
    std::vector<float> h_sdt, h_disc, h_beta;
    // device tables
    DevBuf d_dt, d_sdt, d_disc, d_beta, d_wts, d_sctab;
    // workspace + staging (sized for max_batch)
    DevBuf d_ustg;            // per-step control table [B][H][36] of the solve kernel's long-horizon instantiation (KArgs::ustg)
    DevBuf d_part, d_act, d_traj, d_x0, d_u, d_xref, d_noise, d_step, d_cost, d_grad, d_xmean, d_uopt, d_info;
    // canonical-layout staging of the host-pointer entry points (allocated on their first use)
    DevBuf d_noise_canon, d_traj_canon, d_keys;
    DevBuf d_work;            // u64[4] work counters (KArgs::work)
    // cooperative latency path of the solve (allocated on its first use, sized for coop_cap instances)
    DevBuf d_coop_bar, d_coop_pp, d_coop_ck;
    int coop_cap = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    const void* last_fn = nullptr;   // host function pointer of the kernel the last *_dev launch started (sdempc_last_kernel_name)
    int spin_us = -1;         // SDEMPC_OPT_COOP_SPIN_US: budget of one grid barrier in microseconds; -1 = derived (coop_spin_ticks)
    float coop_ms_last = 0.0f;   // duration of the last cooperative-layout solve that completed (0: none measured yet)
    int last_coop_B = 0;      // > 0: the last solve launch took the cooperative path with this many instances (error flags to check)
    bool coop_off = false;    // a grid barrier timed out once: this handle stays on the one-workgroup-per-instance layouts
    int layout_fallbacks = 0; // how often that happened (sdempc_layout_fallbacks)
};

namespace {

int fail(sdempc_handle* h, int code, const char* fmt, const char* detail = "") {
    char buf[512];
    snprintf(buf, sizeof buf, fmt, detail);
    if (h) h->err = buf; else g_create_error = buf;
    return code;
}

// No exception may cross the C ABI (include/sdempc.h: "never aborts or throws"; an exception in the reference's forked worker would
// end mpc_process silently, sde_control.py:365-419): every entry point runs inside guarded(), which turns std::bad_alloc into
// SDEMPC_ENOMEM and anything else into SDEMPC_EINVAL. The handlers themselves allocate nothing that can throw past them.
void set_error_nothrow(const sdempc_handle* h, const char* msg) noexcept {
    try { if (h) h->err = msg; else g_create_error = msg; } catch (...) { /* no room even for the message: the code still reports it */ }
}
template <class F>
int guarded(const sdempc_handle* h, F&& f) noexcept {
    try { return f(); }
    catch (const std::bad_alloc&) { set_error_nothrow(h, "out of host memory"); return SDEMPC_ENOMEM; }
    catch (const std::length_error&) { set_error_nothrow(h, "a table size derived from the arguments exceeds what the host can allocate"); return SDEMPC_ENOMEM; }
    catch (const std::exception& e) { set_error_nothrow(h, e.what()); return SDEMPC_EINVAL; }
    catch (...) { set_error_nothrow(h, "unexpected exception inside libsdempc"); return SDEMPC_EINVAL; }
}

#define HIPCHK(h, call)                                                                 \
    do {                                                                                \
        hipError_t e__ = (call);                                                        \
        if (e__ != hipSuccess) {                                                        \
            char b__[400];                                                              \
            snprintf(b__, sizeof b__, "%s failed: %s", #call, hipGetErrorString(e__)); \
            (h)->err = b__;                                                             \
            return SDEMPC_EDEVICE;                                                      \
        }                                                                               \
    } while (0)

// Environment variables give the DEFAULTS of a new handle's options, read once here (sdempc_create); the launch path never reads the
// environment. Unset or malformed -> the built-in default.
int env_int(const char* name, int dflt, int lo, int hi) {
    const char* e = getenv(name);
    if (!e || !*e) return dflt;
    char* end = nullptr;
    const long v = strtol(e, &end, 10);
    if (end == e || v < lo || v > hi) return dflt;
    return (int)v;
}
void default_options(sdempc_handle* h) {
    LaunchOpts& o = h->base.opt;
    o.cus = 256;                                       // replaced by the device's count when the device is bound (ensure_device)
    o.lane = env_int("SDEMPC_LANE", 1, 0, 1);
    o.coop = env_int("SDEMPC_COOP", 1, 0, 1);
    o.spec = env_int("SDEMPC_SPEC", 1, 0, 1);
    o.pk = env_int("SDEMPC_PK", -1, -1, 1);
    o.ustg = env_int("SDEMPC_USTG", -1, -1, 1);
    o.duo = env_int("SDEMPC_DUO", -1, -1, 1);
    o.coop_launch = env_int("SDEMPC_COOP_LAUNCH", 0, 0, 1);
    o.coop_fence = env_int("SDEMPC_COOP_FENCE", 0, 0, 1);
    o.hex = env_int("SDEMPC_HEX", 1, 0, 1);
    o.absent_wg = -1;                                  // fault injection is a per-handle option only (SDEMPC_OPT_TEST_ABSENT_WG): never from the environment
    h->spin_us = env_int("SDEMPC_COOP_SPIN_US", -1, -1, 10 * 1000 * 1000);
}
// Budget of one grid barrier of the cooperative layouts, in 10 ns ticks (KArgs::coop_spin). A barrier is passed ~780 times per C2
// solve, microseconds each; one that waits this long means the grid is not fully resident (the GPU is shared) and the launch gives
// up so that the caller's control tick falls back to the one-workgroup-per-instance layout instead of stalling. Derived: five times
// the last completed cooperative solve of this handle, between 2 ms and 100 ms; 100 ms before the first one.
unsigned coop_spin_ticks(const sdempc_handle* h) {
    if (h->spin_us >= 0) return (unsigned)((uint64_t)h->spin_us * 100u > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)h->spin_us * 100u);
    float ms = h->coop_ms_last > 0.0f ? 5.0f * h->coop_ms_last : 100.0f;
    if (ms < 2.0f) ms = 2.0f;
    if (ms > 100.0f) ms = 100.0f;
    return (unsigned)(ms * 1e5f);
}

int dev_alloc(sdempc_handle* h, DevBuf& b, size_t bytes) {
    if (bytes == 0) bytes = 16;
    HIPCHK(h, hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    return 0;
}
void dev_free(DevBuf& b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
}

size_t noise_floats(const sdempc_handle* h, int B) { return (size_t)B * h->G * h->H * SDEMPC_NNOISE * 32; }
size_t traj_floats(const sdempc_handle* h, int B) { return (size_t)B * h->G * (h->H + 1) * SDEMPC_NX * 32; }

int ensure_device_impl(sdempc_handle* h) {
    if (h->dev_ready) {
        HIPCHK(h, hipSetDevice(h->device));
        return 0;
    }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(h, SDEMPC_EDEVICE, "no HIP device available (%s); sdempc has no CPU fallback", hipGetErrorString(e));
    if (h->device >= n) return fail(h, SDEMPC_EDEVICE, "device ordinal out of range%s");
    HIPCHK(h, hipSetDevice(h->device));
    {   // compute units of THIS handle's device (the layout heuristics count workgroups against it)
        int cus = 0;
        HIPCHK(h, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device));
        if (cus > 0) h->base.opt.cus = cus;
    }
    HIPCHK(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    HIPCHK(h, hipEventCreate(&h->ev0));
    HIPCHK(h, hipEventCreate(&h->ev1));
    const int H = h->H, m = h->m, B = h->max_batch;
    int rc;
    if ((rc = dev_alloc(h, h->d_dt, sizeof(float) * H))) return rc;
    if ((rc = dev_alloc(h, h->d_sdt, sizeof(float) * H * SDEMPC_NNOISE))) return rc;
    if ((rc = dev_alloc(h, h->d_disc, sizeof(float) * (H + 1)))) return rc;
    if ((rc = dev_alloc(h, h->d_beta, sizeof(float) * h->h_beta.size()))) return rc;
    if ((rc = dev_alloc(h, h->d_wts, sizeof(float) * h->blob_f.size()))) return rc;
    HIPCHK(h, hipMemcpy(h->d_dt.p, h->time_steps.data(), sizeof(float) * H, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->d_sdt.p, h->h_sdt.data(), sizeof(float) * H * SDEMPC_NNOISE, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->d_disc.p, h->h_disc.data(), sizeof(float) * (H + 1), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->d_beta.p, h->h_beta.data(), sizeof(float) * h->h_beta.size(), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->d_wts.p, h->blob_f.data(), sizeof(float) * h->blob_f.size(), hipMemcpyHostToDevice));
    if (h->cfg.num_state_constr > 0) {       // state_constr table (SPEC.md §5.3)
        CostK::StateBound tab[SDEMPC_NX];
        for (int k = 0; k < h->cfg.num_state_constr; ++k)
            tab[k] = CostK::StateBound{h->cfg.state_id[k], h->cfg.state_w[k], h->cfg.state_lo[k], h->cfg.state_hi[k]};
        if ((rc = dev_alloc(h, h->d_sctab, sizeof(CostK::StateBound) * h->cfg.num_state_constr))) return rc;
        HIPCHK(h, hipMemcpy(h->d_sctab.p, tab, sizeof(CostK::StateBound) * h->cfg.num_state_constr, hipMemcpyHostToDevice));
        h->base.C.sc_n = h->cfg.num_state_constr;
        h->base.C.sc_tab = (const CostK::StateBound*)h->d_sctab.p;
    }
    if ((rc = dev_alloc(h, h->d_x0, sizeof(float) * B * SDEMPC_NX))) return rc;
    if ((rc = dev_alloc(h, h->d_u, sizeof(float) * B * H * m))) return rc;
    if ((rc = dev_alloc(h, h->d_xref, sizeof(float) * B * (H + 1) * SDEMPC_NX))) return rc;
    if ((rc = dev_alloc(h, h->d_noise, sizeof(float) * noise_floats(h, B)))) return rc;
    if ((rc = dev_alloc(h, h->d_step, sizeof(float) * B))) return rc;
    if ((rc = dev_alloc(h, h->d_cost, sizeof(float) * B))) return rc;
    if ((rc = dev_alloc(h, h->d_grad, sizeof(float) * B * H * m))) return rc;
    if ((rc = dev_alloc(h, h->d_xmean, sizeof(float) * B * (H + 1) * SDEMPC_NX))) return rc;
    if ((rc = dev_alloc(h, h->d_uopt, sizeof(float) * B * H * m))) return rc;
    if ((rc = dev_alloc(h, h->d_info, sizeof(float) * B * 8))) return rc;
    if ((rc = dev_alloc(h, h->d_work, sizeof(unsigned long long) * 5))) return rc;      // 4 counters + the persistent launches' instance ticket
    HIPCHK(h, hipMemset(h->d_work.p, 0, h->d_work.bytes));
    h->base.work = (unsigned long long*)h->d_work.p;
    h->base.ticket_host = &h->ticket_total;
    h->ticket_total = 0;                                   // matches the zeroed ticket word
    h->base.dt = (const float*)h->d_dt.p;
    h->base.sdt = (const float*)h->d_sdt.p;
    h->base.disc = (const float*)h->d_disc.p;
    h->base.beta = (const float*)h->d_beta.p;
    h->base.wts = (const float*)h->d_wts.p;
    h->dev_ready = true;
    return 0;
}

// The kernels' own workspaces — particle x horizon tensor, activation checkpoint, per-group partial sums, control table in global memory —
// hold `rows` instances: one row per instance for the one-workgroup-per-instance layouts, one per TEAM SLOT for the persistent throughput
// launches, which is what keeps a large batch small (C2: 1.5 MB per row; 1,536 slots = 2.3 GB whatever the batch). Allocated on the first
// launch that needs them and grown when a later launch needs more rows (never shrunk).
int ensure_workspace(sdempc_handle* h, int rows) {
    if (rows <= h->ws_rows) return 0;
    HIPCHK(h, hipStreamSynchronize(h->stream));          // nothing may still be using the old rows (caller streams: the caller's business, as for every _dev entry point)
    for (DevBuf* b : {&h->d_traj, &h->d_act, &h->d_part, &h->d_ustg}) dev_free(*b);
    h->ws_rows = 0; h->base.ws_rows = 0; h->traj_batch = 0;
    h->base.traj = h->base.act = h->base.part = h->base.ustg = nullptr;
    const int H = h->H;
    int rc;
    if ((rc = dev_alloc(h, h->d_traj, sizeof(float) * traj_floats(h, rows)))) return rc;
    HIPCHK(h, hipMemset(h->d_traj.p, 0, h->d_traj.bytes));
    if ((rc = dev_alloc(h, h->d_act, sizeof(float) * (size_t)rows * h->G * H * ACT_STRIDE))) return rc;
    if ((rc = dev_alloc(h, h->d_part, sizeof(float) * (size_t)rows * h->G * part_stride(H)))) return rc;
    if ((rc = dev_alloc(h, h->d_ustg, sizeof(float) * (size_t)rows * H * 36))) return rc;
    h->base.traj = (float*)h->d_traj.p;
    h->base.act = (float*)h->d_act.p;
    h->base.part = (float*)h->d_part.p;
    h->base.ustg = (float*)h->d_ustg.p;
    h->ws_rows = rows;
    h->base.ws_rows = rows;
    return 0;
}

void release_device(sdempc_handle* h);

// Lazy device initialisation; a failure half-way (e.g. out of HBM) releases what was allocated so that a later call starts clean.
int ensure_device(sdempc_handle* h) {
    const bool was_ready = h->dev_ready;
    const int rc = ensure_device_impl(h);
    if (rc != 0 && !was_ready) { const std::string keep = h->err; release_device(h); h->err = keep; }
    return rc;
}

int check_batch(sdempc_handle* h, int B) {
    if (!h) return SDEMPC_EINVAL;
    if (B < 1) return fail(h, SDEMPC_EINVAL, "batch must be >= 1%s");
    if (B > h->max_batch) return fail(h, SDEMPC_ECAPACITY, "batch exceeds max_batch given to sdempc_create%s");
    return 0;
}

void noise_to_dev_layout(const sdempc_handle* h, int B, const float* in, float* out) {
    const int P = h->P, H = h->H, G = h->G;
    memset(out, 0, sizeof(float) * noise_floats(h, B));
    for (int b = 0; b < B; ++b)
        for (int p = 0; p < P; ++p) {
            const int g = p / 32, l = p % 32;
            for (int t = 0; t < H; ++t)
                for (int i = 0; i < SDEMPC_NNOISE; ++i)
                    out[((((size_t)b * G + g) * H + t) * SDEMPC_NNOISE + i) * 32 + l] = in[(((size_t)b * P + p) * H + t) * SDEMPC_NNOISE + i];
        }
}

int stage_common(sdempc_handle* h, int B, const float* x0, const float* u, const float* xref, const float* noise) {
    const int H = h->H, m = h->m;
    int rc;
    if (!h->d_noise_canon.p && (rc = dev_alloc(h, h->d_noise_canon, sizeof(float) * (size_t)h->max_batch * h->P * H * SDEMPC_NNOISE))) return rc;
    HIPCHK(h, hipMemcpyAsync(h->d_x0.p, x0, sizeof(float) * B * SDEMPC_NX, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_u.p, u, sizeof(float) * B * H * m, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_xref.p, xref, sizeof(float) * B * (H + 1) * SDEMPC_NX, hipMemcpyHostToDevice, h->stream));
    // the caller's canonical [B][P][H][6] tensor goes up as it is; the particle-minor layout is made on the device
    HIPCHK(h, hipMemcpyAsync(h->d_noise_canon.p, noise, sizeof(float) * (size_t)B * h->P * H * SDEMPC_NNOISE, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, launch_relayout(true, (const float*)h->d_noise_canon.p, (float*)h->d_noise.p, B, h->P, h->G, H * SDEMPC_NNOISE, h->stream));
    return 0;
}

// keys (host u32[B][2]) -> device staging -> noise in the device layout at out_dev, on stream st
int noise_from_keys(sdempc_handle* h, int B, const uint32_t* keys, float* out_dev, hipStream_t st) {
    int rc;
    if (!h->d_keys.p && (rc = dev_alloc(h, h->d_keys, sizeof(uint32_t) * 2 * (size_t)h->max_batch))) return rc;
    HIPCHK(h, hipMemcpyAsync(h->d_keys.p, keys, sizeof(uint32_t) * 2 * (size_t)B, hipMemcpyHostToDevice, st));
    HIPCHK(h, launch_noise_from_keys((const uint32_t*)h->d_keys.p, out_dev, B, h->P, h->G, h->H, st));
    return 0;
}

// after a synchronised cooperative solve: did any grid barrier give up? (telemetry is NaN in that case as well)
// Did a grid barrier of the last (cooperative-layout) solve launch give up? Call after the launch's stream has been synchronised.
int coop_timed_out(sdempc_handle* h, bool* timed_out) {
    *timed_out = false;
    if (h->last_coop_B <= 0) return 0;
    std::vector<unsigned> f(COOP_BAR_WORDS * (size_t)h->last_coop_B);
    HIPCHK(h, hipMemcpy(f.data(), h->d_coop_bar.p, sizeof(unsigned) * f.size(), hipMemcpyDeviceToHost));
    for (int b = 0; b < h->last_coop_B; ++b)
        if (f[COOP_BAR_WORDS * b + 1] != 0u) *timed_out = true;
    if (*timed_out) {                    // the workgroups were not all resident (GPU shared with other work): no second try on this handle
        h->coop_off = true;
        h->layout_fallbacks += 1;
        h->last_coop_B = 0;
    } else if (h->timed) {               // completed: its duration scales the next launch's barrier budget (coop_spin_ticks)
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, h->ev0, h->ev1) == hipSuccess && ms > 0.0f) h->coop_ms_last = ms;
    }
    return 0;
}

// A ticketed persistent launch (sdempc_kernels.hip, launch_persistent) draws its instances relative to the value the device's ticket word
// had at launch, which the host mirrors (ticket_total) on the assumption that every launch advances the word by exactly its batch size.
// Anything that breaks the assumption — a kernel that was aborted, two launches of one handle overlapping on different streams, a
// captured graph replaying a launch with its baked-in base — would leave instances unsolved with stale outputs. Call after the launch's
// stream has been synchronised: compares the word with the mirror, re-synchronises the mirror and reports the launch as failed.
int tickets_consistent(sdempc_handle* h) {
    if (!h->last_ticketed || !h->dev_ready) return 0;
    h->last_ticketed = false;
    unsigned word = 0;
    HIPCHK(h, hipMemcpy(&word, (const char*)h->d_work.p + 4 * sizeof(unsigned long long), sizeof word, hipMemcpyDeviceToHost));
    if (word == h->ticket_total) return 0;
    char msg[256];
    snprintf(msg, sizeof msg, "ticketed launch: the device's ticket word reads %u where the host expects %u (aborted kernel, overlapping launches of one "
                              "handle, or a replayed graph); instances of that launch may be unsolved", word, h->ticket_total);
    h->ticket_total = word;
    return fail(h, SDEMPC_EDEVICE, "%s", msg);
}

template <class F>
int timed_launch(sdempc_handle* h, hipStream_t st, F&& f) {
    HIPCHK(h, hipEventRecord(h->ev0, st));
    hipError_t e = f();
    if (e != hipSuccess) return fail(h, SDEMPC_EDEVICE, "kernel launch failed: %s", hipGetErrorString(e));
    HIPCHK(h, hipEventRecord(h->ev1, st));
    h->timed = true;
    h->last_fn = last_launched_kernel();
    return 0;
}


int solve_staged(sdempc_handle* h, int32_t B, float* uopt, float* xevol, sdempc_info* info);
}  // namespace

extern "C" {

int sdempc_abi_version(void) { return SDEMPC_ABI_VERSION; }
int sdempc_build_flags(void) { return SDEMPC_ALL_VARIANTS ? SDEMPC_BUILD_ALL_VARIANTS : 0; }

const char* sdempc_last_error(const sdempc_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int sdempc_create(const sdempc_cfg* cfg, const void* model_blob, size_t blob_bytes, int32_t max_batch, sdempc_handle** out) {
    return guarded(nullptr, [&]() -> int {
    if (!out) return fail(nullptr, SDEMPC_EINVAL, "out is NULL%s");
    *out = nullptr;
    if (!cfg || cfg->struct_size != (int32_t)sizeof(sdempc_cfg)) return fail(nullptr, SDEMPC_EINVAL, "cfg NULL or struct_size mismatch%s");
    if (!model_blob || blob_bytes < sizeof(int32_t) * SDEMPC_BLOB_HEADER_INTS + sizeof(float) * SDEMPC_BLOB_FLOATS)
        return fail(nullptr, SDEMPC_EBLOB, "model blob too small%s");
    const int32_t* hd = (const int32_t*)model_blob;
    if (hd[0] != SDEMPC_BLOB_MAGIC || hd[1] != 1 || hd[3] != SDEMPC_HID || hd[4] != 6 || hd[5] != SDEMPC_NNOISE)
        return fail(nullptr, SDEMPC_EBLOB, "model blob header mismatch%s");
    const int m = hd[2];
    if (m < 1 || m > SDEMPC_MAX_MOTORS || m != cfg->num_motors) return fail(nullptr, SDEMPC_EINVAL, "num_motors of cfg and model blob differ or out of range%s");
    if (cfg->horizon < 1 || cfg->horizon > 4096 || cfg->num_particles < 1 || !cfg->time_steps || max_batch < 1)
        return fail(nullptr, SDEMPC_EINVAL, "horizon/num_particles/time_steps/max_batch invalid%s");
    if (cfg->max_iter < 0 || cfg->ls_maxls < 0 || cfg->max_no_improvement_iter < 1) return fail(nullptr, SDEMPC_EINVAL, "apg parameters invalid%s");
    if (cfg->num_state_constr < 0 || cfg->num_state_constr > SDEMPC_NX) return fail(nullptr, SDEMPC_EINVAL, "num_state_constr out of range%s");
    for (int k = 0; k < cfg->num_state_constr; ++k)
        if (cfg->state_id[k] < 0 || cfg->state_id[k] >= SDEMPC_NX || (k > 0 && cfg->state_id[k] <= cfg->state_id[k - 1]))
            return fail(nullptr, SDEMPC_EINVAL, "state_id must be strictly ascending indices into the 13-state%s");
    for (int t = 0; t < cfg->horizon; ++t)
        if (!(cfg->time_steps[t] > 0.0f)) return fail(nullptr, SDEMPC_EINVAL, "time_steps must be positive%s");
    if (smem_bytes(cfg->horizon, m, team_ipb((cfg->num_particles + 31) / 32, cfg->horizon, m)) > 160 * 1024) return fail(nullptr, SDEMPC_EINVAL, "horizon too large for one workgroup's LDS (160 KiB)%s");
    if (cfg->max_iter > 10 * 1000 * 1000) return fail(nullptr, SDEMPC_EINVAL, "max_iter beyond 10^7%s");
    std::unique_ptr<sdempc_handle> hp(new (std::nothrow) sdempc_handle());      // released into *out on success only
    sdempc_handle* h = hp.get();
    if (!h) return fail(nullptr, SDEMPC_ENOMEM, "out of memory%s");
    h->cfg = *cfg;
    h->H = cfg->horizon; h->P = cfg->num_particles; h->m = m; h->G = (h->P + 31) / 32; h->max_batch = max_batch;
    h->time_steps.assign(cfg->time_steps, cfg->time_steps + h->H);
    h->cfg.time_steps = h->time_steps.data();
    const float* f = (const float*)(hd + SDEMPC_BLOB_HEADER_INTS);
    h->blob_f.assign(f, f + SDEMPC_BLOB_FLOATS);
    if (cfg->mlp_dtype < 0 || cfg->mlp_dtype > 2) return fail(nullptr, SDEMPC_EINVAL, "mlp_dtype must be 0 (f32), 1 (f16) or 2 (f32x3)%s");
    if (cfg->math_mode != 0 && cfg->math_mode != 1) return fail(nullptr, SDEMPC_EINVAL, "math_mode must be 0 (exact) or 1 (fast)%s");
    if (cfg->mlp_dtype == 1) {   // layer-1 state-input weights and layer-2 weights live in fp16 (forward and adjoint alike)
        for (int i = 0; i < 64 * 6; ++i) h->blob_f[blob::W1Z + i] = f16_rtz_host(h->blob_f[blob::W1Z + i]);
        for (int i = 0; i < 32 * 32; ++i) h->blob_f[blob::W2 + i] = f16_rtz_host(h->blob_f[blob::W2 + i]);
    }
    if (cfg->math_mode == 1) {
        // SPEC.md §10b: the hardware tanh is evaluated as r = rcp(1 + exp2(a')), a' = (2 log2 e) a, tanh(a) = 1 - 2 r. The pre-scale goes into the weights
        // and biases that feed a tanh (one rounding each), the affine map 1 - 2 r into the weights and biases that consume one (exact factors, biases by
        // sequential sums), and the derivative 1 - tanh^2 = 4 (r - r^2) leaves its factor 4 in the transposed weights (exact). The device blob becomes
        // two blocks of the same layout: [forward weights][weights of the vector-Jacobian products]; float32 host arithmetic, no contraction.
        const float c = 2.885390043258667f;
        const std::vector<float> o = h->blob_f;
        std::vector<float> F = o, V = o;
        for (int i = 0; i < 64 * 6; ++i) { float w = c * o[blob::W1Z + i]; F[blob::W1Z + i] = cfg->mlp_dtype == 1 ? f16_rtz_host(w) : w; }
        for (int i = 0; i < 64; ++i) F[blob::B1 + i] = c * o[blob::B1 + i];
        for (int i = 0; i < 32 * 8; ++i) F[blob::W1U + i] = c * o[blob::W1U + i];
        for (int j = 0; j < 32; ++j) {
            float sum = c * o[blob::B2 + j];
            for (int k = 0; k < 32; ++k) {
                float w = c * o[blob::W2 + j * 32 + k];
                if (cfg->mlp_dtype == 1) w = f16_rtz_host(w);
                sum = sum + w;
                F[blob::W2 + j * 32 + k] = -2.0f * w;
            }
            F[blob::B2 + j] = sum;
        }
        for (int i = 0; i < 6; ++i) {
            float sum = o[blob::B3 + i];
            for (int k = 0; k < 32; ++k) { sum = sum + o[blob::W3 + i * 32 + k]; F[blob::W3 + i * 32 + k] = -2.0f * o[blob::W3 + i * 32 + k]; }
            F[blob::B3 + i] = sum;
        }
        {
            float sum = o[blob::B3N];
            for (int k = 0; k < 32; ++k) { sum = sum + o[blob::W3N + k]; F[blob::W3N + k] = -2.0f * o[blob::W3N + k]; }
            F[blob::B3N] = sum;
        }
        for (int i = 0; i < 32 * 32; ++i) V[blob::W2 + i] = 4.0f * o[blob::W2 + i];
        for (int i = 0; i < 6 * 32; ++i) V[blob::W3 + i] = 4.0f * o[blob::W3 + i];
        for (int k = 0; k < 32; ++k) V[blob::W3N + k] = 4.0f * o[blob::W3N + k];
        if (cfg->mlp_dtype == 1) {
            // mlp_dtype f16 quantises c * w once more (toward zero), so the forward pass evaluates the weights F / c, not the Wq the blob held:
            // the vector-Jacobian products differentiate what was evaluated (SPEC.md §10b, last item of "Adjoint")
            for (int i = 0; i < 64 * 6; ++i) V[blob::W1Z + i] = F[blob::W1Z + i] / c;
            for (int i = 0; i < 32 * 32; ++i) V[blob::W2 + i] = (-2.0f * F[blob::W2 + i]) / c;
        }
        h->blob_f = F;
        h->blob_f.insert(h->blob_f.end(), V.begin(), V.end());
    }
    // tables (SPEC.md §5: float32 host arithmetic)
    const float* sigma = f + blob::SIGMA;
    h->h_sdt.resize((size_t)h->H * SDEMPC_NNOISE);
    for (int t = 0; t < h->H; ++t) {
        float sq = sqrtf(h->time_steps[t]);
        for (int i = 0; i < SDEMPC_NNOISE; ++i) h->h_sdt[(size_t)t * SDEMPC_NNOISE + i] = sigma[i] * sq;
    }
    h->h_disc.resize(h->H + 1);
    float d = 1.0f / (float)h->H;
    for (int t = 0; t <= h->H; ++t) { h->h_disc[t] = d; d = d * cfg->discount; }
    h->h_beta.resize(cfg->max_iter + 2);
    for (int i = 0; i < cfg->max_iter + 2; ++i) {
        float b = (i == 0) ? cfg->beta_init : (float)(i + 1) / (float)(i + 4);
        if (i > 0 && cfg->use_moment_scale) b = cfg->moment_scale * b;
        h->h_beta[i] = b;
    }
    // kernel argument block
    KArgs& a = h->base;
    memset(&a, 0, sizeof a);
    a.H = h->H; a.P = h->P; a.m = m; a.G = h->G;
    a.invP = 1.0f / (float)h->P;
    a.f16 = cfg->mlp_dtype;              // 0 f32, 1 fp16 operands (SPEC.md §9), 2 three-limb bf16 split of the layer-2 contractions (§9b)
    a.fast = cfg->math_mode == 1;
    a.M.inv_mass = f[0]; a.M.grav = f[1];
    for (int i = 0; i < 3; ++i) { a.M.J[i] = f[2 + i]; a.M.iJ[i] = f[5 + i]; }
    a.M.ct2 = f[8]; a.M.ct1 = f[9]; a.M.ct0 = f[10]; a.M.cm2 = f[11]; a.M.cm1 = f[12];
    for (int j = 0; j < 8; ++j) { a.M.rx[j] = f[16 + j]; a.M.ry[j] = f[24 + j]; a.M.dir[j] = f[32 + j]; }
    for (int i = 0; i < 3; ++i) { a.M.sF[i] = f[40 + i]; a.M.sT[i] = f[43 + i]; }
    for (int i = 0; i < 6; ++i) a.M.b3[i] = h->blob_f[blob::B3 + i];      // (math_mode fast: the forward block's, SPEC.md §10b)
    a.M.b3n = h->blob_f[blob::B3N];
    a.M.adj_s0 = -2.0f; a.M.adj_i0 = 1.0f;
    if (cfg->math_mode == 1 && cfg->mlp_dtype != 0) {
        // SPEC.md §10e: the scale offset of the adjoint's binary16 contractions. With the largest output adjoint of a particle scaled into [2^eoff, 2^(eoff+1)),
        // |abar2| < 2^(eoff+1) B3/4, |abar1n| < 2^(eoff+1) Bn/4, |abar1d| < 2^(eoff+1) C2 B3/16 (|r - r^2| <= 1/4; B3, Bn: absolute column sums of the forward
        // output weights, C2: of 4 W2): eoff = min(10, 14 - e) with 2^e > the largest of the three bounds keeps all of them below 2^15 (binary16 ends at 65504).
        // float32 host arithmetic, sums in ascending index order; the oracle derives the same number by the same statements.
        const float* F = h->blob_f.data();
        const float* V = F + SDEMPC_BLOB_FLOATS;
        float B3 = 0.0f, Bn = 0.0f, C2 = 0.0f;
        for (int k = 0; k < 32; ++k) {
            float s3 = 0.0f, s2 = 0.0f;
            for (int i = 0; i < 6; ++i) s3 = s3 + fabsf(F[blob::W3 + i * 32 + k]);
            for (int j = 0; j < 32; ++j) s2 = s2 + fabsf(V[blob::W2 + j * 32 + k]);
            B3 = fmaxf(B3, s3); C2 = fmaxf(C2, s2); Bn = fmaxf(Bn, fabsf(F[blob::W3N + k]));
        }
        const float bound = fmaxf(fmaxf(B3 * 0.25f, Bn * 0.25f), (C2 * (B3 * 0.25f)) * 0.25f);
        int e = 0, eoff = 10;
        if (bound > 0.0f && bound < INFINITY) { (void)frexpf(bound, &e); if (14 - e < eoff) eoff = 14 - e; }
        if (eoff < -40) eoff = -40;
        a.M.adj_s0 = ldexpf(-2.0f, eoff); a.M.adj_i0 = ldexpf(1.0f, -eoff);
    }
    for (int i = 0; i < 3; ++i) { a.C.perr[i] = cfg->perr[i]; a.C.verr[i] = cfg->verr[i]; a.C.qerr[i] = cfg->qerr[i]; a.C.werr[i] = cfg->werr[i]; }
    a.C.res_mult = cfg->res_mult; a.C.uerr = cfg->uerr; a.C.slew = cfg->u_slew_coeff; a.C.slew_cc = cfg->u_slew_constr_coeff;
    a.C.has_sc = cfg->has_slew_constr;
    for (int j = 0; j < 8; ++j) {
        a.C.slew_lo[j] = cfg->u_slew_lo[j]; a.C.slew_hi[j] = cfg->u_slew_hi[j]; a.C.uref[j] = cfg->uref[j];
        a.C.ulo[j] = cfg->u_lo[j]; a.C.uhi[j] = cfg->u_hi[j];
    }
    a.C.sc_n = 0; a.C.sc_tab = nullptr;       // the table goes to the device with the other tables (ensure_device)
    a.A.max_iter = cfg->max_iter; a.A.max_noimp = cfg->max_no_improvement_iter; a.A.maxls = cfg->ls_maxls;
    a.A.reset_inc = cfg->ls_reset_option == 1;
    a.A.atol = cfg->atol; a.A.rtol = cfg->rtol; a.A.stepsize = cfg->stepsize; a.A.smax = cfg->ls_max_stepsize;
    a.A.coef = cfg->ls_coef; a.A.dec = cfg->ls_decrease_factor; a.A.inc = cfg->ls_increase_factor;
    default_options(h);
    *out = hp.release();
    return SDEMPC_OK;
    });
}

void sdempc_destroy(sdempc_handle* h) {
    if (!h) return;
    release_device(h);
    delete h;
}

}  // extern "C"

namespace {
void release_device(sdempc_handle* h) {
    if (h->dev_ready || h->stream || h->d_dt.p) {
        (void)hipSetDevice(h->device);
        for (DevBuf* b : {&h->d_sctab, &h->d_ustg, &h->d_part, &h->d_act, &h->d_dt, &h->d_sdt, &h->d_disc, &h->d_beta, &h->d_wts, &h->d_traj, &h->d_x0, &h->d_u, &h->d_xref, &h->d_noise, &h->d_noise_canon, &h->d_traj_canon, &h->d_keys, &h->d_work, &h->d_coop_bar, &h->d_coop_pp, &h->d_coop_ck,
                          &h->d_step, &h->d_cost, &h->d_grad, &h->d_xmean, &h->d_uopt, &h->d_info})
            dev_free(*b);
        if (h->ev0) (void)hipEventDestroy(h->ev0);
        if (h->ev1) (void)hipEventDestroy(h->ev1);
        if (h->stream) (void)hipStreamDestroy(h->stream);
        h->ev0 = h->ev1 = nullptr; h->stream = nullptr; h->coop_cap = 0; h->ws_rows = 0;
    }
    h->dev_ready = false; h->timed = false;
}
}  // namespace

extern "C" {

int sdempc_set_device(sdempc_handle* h, int32_t device) {
    return guarded(h, [&]() -> int {
    if (!h) return SDEMPC_EINVAL;
    if (h->dev_ready) return fail(h, SDEMPC_EINVAL, "sdempc_set_device must precede the first device call%s");
    if (device < 0) return fail(h, SDEMPC_EINVAL, "negative device ordinal%s");
    h->device = device;
    return SDEMPC_OK;
    });
}

int sdempc_device_ready(const sdempc_handle* h) { return h && h->dev_ready ? 1 : 0; }

int sdempc_set_option(sdempc_handle* h, int32_t key, int32_t value) {
    return guarded(h, [&]() -> int {
    if (!h) return SDEMPC_EINVAL;
    LaunchOpts& o = h->base.opt;
    auto flag = [&](int& dst) { if (value != 0 && value != 1) return fail(h, SDEMPC_EINVAL, "option value must be 0 or 1%s"); dst = value; return (int)SDEMPC_OK; };
    auto tri = [&](int& dst) { if (value < -1 || value > 1) return fail(h, SDEMPC_EINVAL, "option value must be -1 (auto), 0 or 1%s"); dst = value; return (int)SDEMPC_OK; };
    switch (key) {
        case SDEMPC_OPT_LANE: return flag(o.lane);
        case SDEMPC_OPT_COOP: { int rc = flag(o.coop); if (rc == SDEMPC_OK && value == 1) h->coop_off = false; return rc; }
        case SDEMPC_OPT_SPEC: return flag(o.spec);
        case SDEMPC_OPT_PK:
            if (value == 1 && !SDEMPC_ALL_VARIANTS) return fail(h, SDEMPC_EINVAL, "this build carries no packed-tanh instantiations (make EXTRA=-DSDEMPC_ALL_VARIANTS=1)%s");
            return tri(o.pk);
        case SDEMPC_OPT_USTG: return tri(o.ustg);
        case SDEMPC_OPT_DUO: return tri(o.duo);
        case SDEMPC_OPT_COOP_LAUNCH: return flag(o.coop_launch);
        case SDEMPC_OPT_COOP_FENCE: return flag(o.coop_fence);
        case SDEMPC_OPT_HEX: return flag(o.hex);
        case SDEMPC_OPT_TEST_ABSENT_WG:
            if (value < -1) return fail(h, SDEMPC_EINVAL, "absent workgroup must be -1 (none) or a workgroup index%s");
            o.absent_wg = value;
            return SDEMPC_OK;
        case SDEMPC_OPT_COOP_SPIN_US:
            if (value < -1) return fail(h, SDEMPC_EINVAL, "spin budget must be -1 (derived) or >= 0 microseconds%s");
            h->spin_us = value;
            return SDEMPC_OK;
        default: return fail(h, SDEMPC_EINVAL, "unknown option key%s");
    }
    });
}

int sdempc_get_option(const sdempc_handle* h, int32_t key, int32_t* value) {
    return guarded(h, [&]() -> int {
    if (!h || !value) return SDEMPC_EINVAL;
    const LaunchOpts& o = h->base.opt;
    switch (key) {
        case SDEMPC_OPT_LANE: *value = o.lane; break;
        case SDEMPC_OPT_COOP: *value = o.coop && !h->coop_off; break;
        case SDEMPC_OPT_SPEC: *value = o.spec; break;
        case SDEMPC_OPT_PK: *value = o.pk; break;
        case SDEMPC_OPT_USTG: *value = o.ustg; break;
        case SDEMPC_OPT_DUO: *value = o.duo; break;
        case SDEMPC_OPT_COOP_LAUNCH: *value = o.coop_launch; break;
        case SDEMPC_OPT_COOP_FENCE: *value = o.coop_fence; break;
        case SDEMPC_OPT_HEX: *value = o.hex; break;
        case SDEMPC_OPT_TEST_ABSENT_WG: *value = o.absent_wg; break;
        case SDEMPC_OPT_COOP_SPIN_US: *value = h->spin_us >= 0 ? h->spin_us : (int32_t)(coop_spin_ticks(h) / 100u); break;
        case SDEMPC_OPT_DEVICE_CUS: *value = o.cus; break;
        default: return SDEMPC_EINVAL;
    }
    return SDEMPC_OK;
    });
}

int sdempc_reset(sdempc_handle* h, const float* x, const float* xdes, float* yk, sdempc_info* info) {
    return guarded(h, [&]() -> int {
    if (!h || !yk || !info) return SDEMPC_EINVAL;
    (void)x; (void)xdes;
    for (int t = 0; t < h->H; ++t)
        for (int j = 0; j < h->m; ++j) yk[t * h->m + j] = h->cfg.uref[j];
    memset(info, 0, sizeof *info);
    info->stepsize = h->cfg.ls_maxls > 0 ? h->cfg.ls_init_stepsize : h->cfg.stepsize;
    return SDEMPC_OK;
    });
}

size_t sdempc_noise_dev_floats(const sdempc_handle* h, int32_t B) { return h ? noise_floats(h, B) : 0; }
size_t sdempc_traj_dev_floats(const sdempc_handle* h, int32_t B) { return h ? traj_floats(h, B) : 0; }

int sdempc_noise_to_device_layout(const sdempc_handle* h, int32_t B, const float* noise_host, float* out_host) {
    return guarded(h, [&]() -> int {
    if (!h || !noise_host || !out_host || B < 1) return SDEMPC_EINVAL;
    noise_to_dev_layout(h, B, noise_host, out_host);
    return SDEMPC_OK;
    });
}

int sdempc_noise_to_device_layout_dev(sdempc_handle* h, int32_t B, const void* noise_canonical_dev, void* noise_out_dev, void* stream) {
    return guarded(h, [&]() -> int {
    int rc = check_batch(h, B);
    if (rc) return rc;
    if (!noise_canonical_dev || !noise_out_dev) return fail(h, SDEMPC_EINVAL, "NULL device pointer%s");
    if ((rc = ensure_device(h))) return rc;
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    HIPCHK(h, launch_relayout(true, (const float*)noise_canonical_dev, (float*)noise_out_dev, B, h->P, h->G, h->H * SDEMPC_NNOISE, st));
    return SDEMPC_OK;
    });
}

int sdempc_traj_to_canonical_dev(sdempc_handle* h, int32_t B, void* traj_out_dev, void* stream) {
    return guarded(h, [&]() -> int {
    int rc = check_batch(h, B);
    if (rc) return rc;
    if (!traj_out_dev) return fail(h, SDEMPC_EINVAL, "NULL device pointer%s");
    if ((rc = ensure_device(h))) return rc;
    if (B > h->traj_batch)
        return fail(h, SDEMPC_EINVAL, h->traj_batch ? "the trajectory workspace holds fewer instances than asked for (last rollout with store_traj was smaller)%s"
                                                     : "the trajectory workspace holds no rollout: run a rollout with store_traj first (a gradient evaluation, a solve or a "
                                                       "workspace reallocation since then has overwritten it)%s");
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    HIPCHK(h, launch_relayout(false, (const float*)h->d_traj.p, (float*)traj_out_dev, B, h->P, h->G, (h->H + 1) * SDEMPC_NX, st));
    return SDEMPC_OK;
    });
}

int sdempc_rollout_batch_dev(sdempc_handle* h, int32_t B, const void* x0_dev, const void* u_dev, const void* xref_dev, const void* noise_dev,
                             void* cost_dev, void* xmean_dev, int32_t store_traj, void* stream) {
    return guarded(h, [&]() -> int {
    int rc = check_batch(h, B);
    if (rc) return rc;
    if (!x0_dev || !u_dev || !xref_dev || !noise_dev || !cost_dev) return fail(h, SDEMPC_EINVAL, "NULL device pointer%s");
    if ((rc = ensure_device(h))) return rc;
    if ((rc = ensure_workspace(h, B))) return rc;
    KArgs a = h->base;
    a.x0 = (const float*)x0_dev; a.u = (const float*)u_dev; a.xref = (const float*)xref_dev; a.noise = (const float*)noise_dev;
    a.cost = (float*)cost_dev; a.xmean = (float*)xmean_dev; a.store_traj = store_traj;
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    h->traj_batch = 0;
    rc = timed_launch(h, st, [&] { return a.fast ? launch_rollout_fast(a, B, st) : launch_rollout(a, B, st); });
    if (rc == SDEMPC_OK && store_traj) h->traj_batch = B;
    return rc;
    });
}

int sdempc_grad_batch_dev(sdempc_handle* h, int32_t B, const void* x0_dev, const void* u_dev, const void* xref_dev, const void* noise_dev,
                          void* cost_dev, void* grad_dev, void* stream) {
    return guarded(h, [&]() -> int {
    int rc = check_batch(h, B);
    if (rc) return rc;
    if (!x0_dev || !u_dev || !xref_dev || !noise_dev || !cost_dev || !grad_dev) return fail(h, SDEMPC_EINVAL, "NULL device pointer%s");
    if ((rc = ensure_device(h))) return rc;
    if ((rc = ensure_workspace(h, B))) return rc;
    KArgs a = h->base;
    a.x0 = (const float*)x0_dev; a.u = (const float*)u_dev; a.xref = (const float*)xref_dev; a.noise = (const float*)noise_dev;
    a.cost = (float*)cost_dev; a.grad = (float*)grad_dev;
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    h->traj_batch = 0;       // (the gradient's forward sweep streams x_t through the trajectory workspace)
    return timed_launch(h, st, [&] { return a.fast ? launch_grad_fast(a, B, st) : launch_grad(a, B, st); });
    });
}

int sdempc_solve_batch_dev(sdempc_handle* h, int32_t B, const void* x0_dev, const void* xref_dev, const void* noise_dev, const void* u_init_dev,
                           const void* stepsize_dev, void* uopt_dev, void* xevol_dev, void* info_dev, void* stream) {
    return guarded(h, [&]() -> int {
    int rc = check_batch(h, B);
    if (rc) return rc;
    if (!x0_dev || !xref_dev || !noise_dev || !u_init_dev || !stepsize_dev || !uopt_dev || !xevol_dev || !info_dev)
        return fail(h, SDEMPC_EINVAL, "NULL device pointer%s");
    if ((rc = ensure_device(h))) return rc;
    {   // workspace rows of this launch: the batch, or the team slots of a persistent throughput launch
        KArgs probe = h->base; probe.B = B;
        if ((rc = ensure_workspace(h, probe.fast ? solve_workspace_rows_fast(probe, B) : solve_workspace_rows(probe, B)))) return rc;
    }
    h->traj_batch = 0;       // (a solve's gradient evaluations stream through the trajectory workspace, indexed by team slot)
    KArgs a = h->base;
    a.x0 = (const float*)x0_dev; a.u = (const float*)u_init_dev; a.xref = (const float*)xref_dev; a.noise = (const float*)noise_dev;
    a.stepsize_in = (const float*)stepsize_dev; a.uopt = (float*)uopt_dev; a.xmean = (float*)xevol_dev; a.info = (float*)info_dev;
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    // Small batches of multi-particle instances: one instance over ceil(P/4) workgroups, one particle per wave (latency path).
    // Same results bit for bit; only taken when every workgroup of the grid is resident at once.
    const bool coop_ok = !a.f16 && !h->coop_off && a.C.sc_n == 0;     // lane layouts: f32 contractions (either math mode), built without the state-bound terms
    const int smax = coop_ok ? spec_max_instances(h->P, h->H, h->m, a.opt) : 0;
    int cmax = coop_ok ? coop_max_instances(h->P, h->H, h->m, a.opt) : 0;
    if (smax > cmax) cmax = smax;
    if (B <= cmax) {
        if (!h->d_coop_bar.p) {
            const int cap = cmax < h->max_batch ? cmax : h->max_batch;
            if ((rc = dev_alloc(h, h->d_coop_bar, sizeof(unsigned) * COOP_BAR_WORDS * (size_t)cap))) return rc;
            if ((rc = dev_alloc(h, h->d_coop_pp, sizeof(float) * coop_pp_floats(h->H, h->G) * cap))) return rc;
            if ((rc = dev_alloc(h, h->d_coop_ck, sizeof(float) * coop_ck_floats(h->H, h->P) * cap))) return rc;
            h->coop_cap = cap;
        }
        if (B <= h->coop_cap) {
            HIPCHK(h, hipMemsetAsync(h->d_coop_bar.p, 0, sizeof(unsigned) * COOP_BAR_WORDS * (size_t)B, st));
            a.coop_bar = (unsigned*)h->d_coop_bar.p; a.coop_pp = (float*)h->d_coop_pp.p; a.coop_ck = (float*)h->d_coop_ck.p;
            a.coop_spin = coop_spin_ticks(h);
            h->last_coop_B = B; h->last_ticketed = false;
            if (B <= smax) {
                // the speculative kernel's outputs are tagged words {value, tag} that readers poll (streamed hand-off, sdempc_spec.inc.h): no tag of an
                // earlier launch may survive (12 MB per C2 instance, a few microseconds of the 20 ms the launch takes)
                HIPCHK(h, hipMemsetAsync(h->d_coop_pp.p, 0, sizeof(float) * coop_pp_floats(h->H, h->G) * (size_t)B, st));
                return timed_launch(h, st, [&] { return a.fast ? launch_solve_spec_fast(a, B, st) : launch_solve_spec(a, B, st); });
            }
            if (B <= coop_max_instances(h->P, h->H, h->m, a.opt)) return timed_launch(h, st, [&] { return a.fast ? launch_solve_coop_fast(a, B, st) : launch_solve_coop(a, B, st); });
            h->last_coop_B = 0;
        }
    }
    h->last_coop_B = 0;
    const unsigned tickets_before = h->ticket_total;
    rc = timed_launch(h, st, [&] { return a.fast ? launch_solve_fast(a, B, st) : launch_solve(a, B, st); });
    h->last_ticketed = rc == SDEMPC_OK && h->ticket_total != tickets_before;
    return rc;
    });
}

int sdempc_noise_from_keys_dev(sdempc_handle* h, int32_t B, const uint32_t* keys, void* noise_out_dev, void* stream) {
    return guarded(h, [&]() -> int {
    int rc = check_batch(h, B);
    if (rc) return rc;
    if (!keys || !noise_out_dev) return fail(h, SDEMPC_EINVAL, "NULL pointer%s");
    if ((rc = ensure_device(h))) return rc;
    return noise_from_keys(h, B, keys, (float*)noise_out_dev, stream ? (hipStream_t)stream : h->stream);
    });
}

int sdempc_noise_from_keys(sdempc_handle* h, int32_t B, const uint32_t* keys, float* noise) {
    return guarded(h, [&]() -> int {
    int rc = check_batch(h, B);
    if (rc) return rc;
    if (!keys || !noise) return fail(h, SDEMPC_EINVAL, "NULL pointer%s");
    if ((rc = ensure_device(h))) return rc;
    const size_t nf = (size_t)h->P * h->H * SDEMPC_NNOISE;
    if (!h->d_noise_canon.p && (rc = dev_alloc(h, h->d_noise_canon, sizeof(float) * (size_t)h->max_batch * nf))) return rc;
    if ((rc = noise_from_keys(h, B, keys, (float*)h->d_noise.p, h->stream))) return rc;
    HIPCHK(h, launch_relayout(false, (const float*)h->d_noise.p, (float*)h->d_noise_canon.p, B, h->P, h->G, h->H * SDEMPC_NNOISE, h->stream));
    HIPCHK(h, hipMemcpyAsync(noise, h->d_noise_canon.p, sizeof(float) * B * nf, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return SDEMPC_OK;
    });
}

int sdempc_last_kernel_name(const sdempc_handle* h, char* buf, size_t n) {
    return guarded(h, [&]() -> int {
    if (!h || !buf || n < 2) return SDEMPC_EINVAL;
    buf[0] = 0;
    if (!h->last_fn) return SDEMPC_OK;
    const char* mangled = hipKernelNameRefByPtr(h->last_fn, h->stream);
    if (!mangled) return SDEMPC_OK;
    int status = 0;
    char* dem = abi::__cxa_demangle(mangled, nullptr, nullptr, &status);
    std::string name = (status == 0 && dem) ? dem : mangled;
    free(dem);
    if (name.rfind("void ", 0) == 0) name = name.substr(5);
    const size_t par = name.rfind("(sdempc::KArgs)");
    if (par != std::string::npos) name = name.substr(0, par);
    snprintf(buf, n, "%s", name.c_str());
    return SDEMPC_OK;
    });
}

float sdempc_last_kernel_ms(const sdempc_handle* h) {
    if (!h || !h->timed) return -1.0f;
    if (hipEventSynchronize(h->ev1) != hipSuccess) return -1.0f;
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, h->ev0, h->ev1) != hipSuccess) return -1.0f;
    return ms;
}

int sdempc_rollout_batch(sdempc_handle* h, int32_t B, const float* x0, const float* u, const float* xref, const float* noise, float* cost,
                         float* traj, float* xmean) {
    return guarded(h, [&]() -> int {
    int rc = check_batch(h, B);
    if (rc) return rc;
    if (!x0 || !u || !xref || !noise || !cost) return fail(h, SDEMPC_EINVAL, "NULL host pointer%s");
    if ((rc = ensure_device(h))) return rc;
    if ((rc = stage_common(h, B, x0, u, xref, noise))) return rc;
    rc = sdempc_rollout_batch_dev(h, B, h->d_x0.p, h->d_u.p, h->d_xref.p, h->d_noise.p, h->d_cost.p, xmean ? h->d_xmean.p : nullptr, traj ? 1 : 0, h->stream);
    if (rc) return rc;
    const int H = h->H, P = h->P;
    HIPCHK(h, hipMemcpyAsync(cost, h->d_cost.p, sizeof(float) * B, hipMemcpyDeviceToHost, h->stream));
    if (xmean) HIPCHK(h, hipMemcpyAsync(xmean, h->d_xmean.p, sizeof(float) * B * (H + 1) * SDEMPC_NX, hipMemcpyDeviceToHost, h->stream));
    if (traj) {
        const size_t nf = (size_t)P * (H + 1) * SDEMPC_NX;
        if (!h->d_traj_canon.p && (rc = dev_alloc(h, h->d_traj_canon, sizeof(float) * h->max_batch * nf))) return rc;
        if ((rc = sdempc_traj_to_canonical_dev(h, B, h->d_traj_canon.p, h->stream))) return rc;
        HIPCHK(h, hipMemcpyAsync(traj, h->d_traj_canon.p, sizeof(float) * B * nf, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return SDEMPC_OK;
    });
}

int sdempc_grad_batch(sdempc_handle* h, int32_t B, const float* x0, const float* u, const float* xref, const float* noise, float* cost, float* grad) {
    return guarded(h, [&]() -> int {
    int rc = check_batch(h, B);
    if (rc) return rc;
    if (!x0 || !u || !xref || !noise || !cost || !grad) return fail(h, SDEMPC_EINVAL, "NULL host pointer%s");
    if ((rc = ensure_device(h))) return rc;
    if ((rc = stage_common(h, B, x0, u, xref, noise))) return rc;
    rc = sdempc_grad_batch_dev(h, B, h->d_x0.p, h->d_u.p, h->d_xref.p, h->d_noise.p, h->d_cost.p, h->d_grad.p, h->stream);
    if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(cost, h->d_cost.p, sizeof(float) * B, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(grad, h->d_grad.p, sizeof(float) * B * h->H * h->m, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return SDEMPC_OK;
    });
}

int sdempc_solve_batch(sdempc_handle* h, int32_t B, const float* x0, const float* xref, const float* noise, const float* u_init,
                       const float* stepsize_in, float* uopt, float* xevol, sdempc_info* info) {
    return guarded(h, [&]() -> int {
    int rc = check_batch(h, B);
    if (rc) return rc;
    if (!x0 || !xref || !noise || !u_init || !stepsize_in || !uopt || !xevol || !info) return fail(h, SDEMPC_EINVAL, "NULL host pointer%s");
    if ((rc = ensure_device(h))) return rc;
    if ((rc = stage_common(h, B, x0, u_init, xref, noise))) return rc;
    HIPCHK(h, hipMemcpyAsync(h->d_step.p, stepsize_in, sizeof(float) * B, hipMemcpyHostToDevice, h->stream));
    return solve_staged(h, B, uopt, xevol, info);
    });
}

int sdempc_solve_batch_keys(sdempc_handle* h, int32_t B, const float* x0, const float* xref, const uint32_t* keys, const float* u_init,
                            const float* stepsize_in, float* uopt, float* xevol, sdempc_info* info) {
    return guarded(h, [&]() -> int {
    int rc = check_batch(h, B);
    if (rc) return rc;
    if (!x0 || !xref || !keys || !u_init || !stepsize_in || !uopt || !xevol || !info) return fail(h, SDEMPC_EINVAL, "NULL host pointer%s");
    if ((rc = ensure_device(h))) return rc;
    const int H = h->H, m = h->m;
    HIPCHK(h, hipMemcpyAsync(h->d_x0.p, x0, sizeof(float) * B * SDEMPC_NX, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_u.p, u_init, sizeof(float) * B * H * m, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_xref.p, xref, sizeof(float) * B * (H + 1) * SDEMPC_NX, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_step.p, stepsize_in, sizeof(float) * B, hipMemcpyHostToDevice, h->stream));
    if ((rc = noise_from_keys(h, B, keys, (float*)h->d_noise.p, h->stream))) return rc;
    return solve_staged(h, B, uopt, xevol, info);
    });
}

int sdempc_solve_status(sdempc_handle* h) {
    return guarded(h, [&]() -> int {
    if (!h) return SDEMPC_EINVAL;
    bool to = false;
    int rc = coop_timed_out(h, &to);
    if (rc) return rc;
    if ((rc = tickets_consistent(h))) return rc;
    return to ? fail(h, SDEMPC_EDEVICE, "cooperative solve: a grid barrier timed out (workgroups not co-resident); results invalid, "
                                        "the handle now stays on the one-workgroup-per-instance layouts%s") : SDEMPC_OK;
    });
}

int32_t sdempc_layout_fallbacks(const sdempc_handle* h) { return h ? h->layout_fallbacks : 0; }

int sdempc_work_counters(sdempc_handle* h, uint64_t out[4], int32_t reset) {
    return guarded(h, [&]() -> int {
    if (!h || !out) return SDEMPC_EINVAL;
    out[0] = out[1] = out[2] = out[3] = 0;
    if (!h->dev_ready) return SDEMPC_OK;
    HIPCHK(h, hipSetDevice(h->device));
    unsigned long long v[4];
    HIPCHK(h, hipMemcpy(v, h->d_work.p, sizeof v, hipMemcpyDeviceToHost));
    for (int i = 0; i < 4; ++i) out[i] = v[i];
    if (reset) HIPCHK(h, hipMemset(h->d_work.p, 0, sizeof v));
    return SDEMPC_OK;
    });
}

}  // extern "C"

namespace {
// Solve on the staged inputs (handle buffers) and fetch the results. If a grid barrier of the cooperative layouts gave up — their
// workgroups were not all resident, i.e. the GPU is shared with other work — the same batch runs once more in the one-workgroup-per-
// instance layout (bit-identical results by construction) and the handle keeps off the cooperative layouts from then on.
int solve_staged(sdempc_handle* h, int32_t B, float* uopt, float* xevol, sdempc_info* info) {
    for (int attempt = 0;; ++attempt) {
        int rc = sdempc_solve_batch_dev(h, B, h->d_x0.p, h->d_xref.p, h->d_noise.p, h->d_u.p, h->d_step.p, h->d_uopt.p, h->d_xmean.p, h->d_info.p, h->stream);
        if (rc) return rc;
        HIPCHK(h, hipMemcpyAsync(uopt, h->d_uopt.p, sizeof(float) * B * h->H * h->m, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipMemcpyAsync(xevol, h->d_xmean.p, sizeof(float) * B * (h->H + 1) * SDEMPC_NX, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipMemcpyAsync(info, h->d_info.p, sizeof(float) * B * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        bool to = false;
        if ((rc = coop_timed_out(h, &to))) return rc;
        if ((rc = tickets_consistent(h))) return rc;
        if (!to) return SDEMPC_OK;
        if (attempt) return fail(h, SDEMPC_EDEVICE, "cooperative solve: a grid barrier timed out twice%s");
    }
}
}  // namespace
