// sdempc_kernels.h — kernel argument block shared by the HIP kernels and the C-ABI host code.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

// Which instantiations a build carries (DESIGN.md §2 "Instantiation set"). The default build holds what a handle with default options can be
// dispatched to for the reference's vehicles (m = 4 Iris, m = 6 Hexa) in every layout, plus the GENERIC motor count (the 8-slot, zero-padded
// instantiation: any m <= 8) in the one-group-per-wave tile layouts and the single-particle lane layout only. `make EXTRA=-DSDEMPC_ALL_VARIANTS=1`
// adds the generic motor count in the duo / six-team / cooperative / speculative layouts and the packed-f32 tanh instantiations of the exact tile
// layout (SDEMPC_OPT_PK; TeamBlock8): + 91 solve kernels, + 10 MB, + 7 CPU-minutes (tests/tools/soak.py draws them when they are there).
#ifndef SDEMPC_ALL_VARIANTS
#define SDEMPC_ALL_VARIANTS 0
#endif

namespace sdempc {

// float payload of the model blob (SPEC.md §2; include/sdempc.h: SDEMPC_BLOB_FLOATS), offsets in floats — one statement for the kernels and for sdempc_create
namespace blob {
constexpr int SF = 40;                   // sF[3], sT[3]
constexpr int SIGMA = 48;
constexpr int W1Z = 56, B1 = W1Z + 64 * 6, W1U = B1 + 64, W2 = W1U + 32 * 8, B2 = W2 + 32 * 32, W3 = B2 + 32, B3 = W3 + 8 * 32, W3N = B3 + 8, B3N = W3N + 32;
constexpr int FLOATS = B3N + 8;
static_assert(FLOATS == 2120, "include/sdempc.h: SDEMPC_BLOB_FLOATS");
}

struct ModelK {  // physics prior + small output-layer constants (SPEC.md §2), passed in SGPRs
    float inv_mass, grav, J[3], iJ[3], ct2, ct1, ct0, cm2, cm1;
    float rx[8], ry[8], dir[8];
    float sF[3], sT[3];
    float b3[6], b3n;
    float adj_s0, adj_i0;   // SPEC.md §10e (math_mode fast + f32x3): -2 * 2^eoff and 2^-eoff, the handle's scale offset of the adjoint's binary16 contractions
};
struct CostK {
    float perr[3], verr[3], qerr[3], werr[3];
    float res_mult, uerr, slew, slew_cc;
    int has_sc;
    float slew_lo[8], slew_hi[8], uref[8], ulo[8], uhi[8];
    int sc_n;                          // state_constr (SPEC.md §5.3): number of bounded states, 0: none
    struct StateBound { int id; float w, lo, hi; };
    const StateBound* sc_tab;          // device table [sc_n], ascending state index (owned by the handle)
};
struct ApgK {
    int max_iter, max_noimp, maxls, reset_inc;
    float atol, rtol, stepsize, smax, coef, dec, inc;
};
// Host-side dispatch options of one handle (include/sdempc.h: sdempc_set_option / SDEMPC_OPT_*). Travels inside KArgs so that the
// launchers see it; device code reads only coop_fence.
struct LaunchOpts {
    int cus;           // compute units of the handle's device (hipDeviceAttributeMultiprocessorCount), set when the device is bound
    int lane;          // 1: single-particle lane layout for P == 1 (default), 0: tile layout
    int coop;          // 1: cooperative multi-workgroup layouts for small batches (default), 0: off
    int spec;          // 1: speculative variant of the cooperative layout for the smallest batches (default), 0: off
    int pk;            // -1: packed-f32 tanh instantiation when the grid leaves one wave per SIMD (default), 0 / 1: forced
    int ustg;          // -1: per-step control table in global memory when that raises occupancy (default), 0 / 1: forced
    int coop_launch;   // 1: hipLaunchCooperativeKernel for the cooperative layouts, 0: plain launch of a grid sized to be resident (default)
    int duo;           // throughput launches in the duo tile layout (64 particles per wave): -1 auto (= on for multi-group instances), 0 off, 1 on
    int coop_fence;    // 1: agent-scope release / acquire fences around the grid barrier, 0: sc1 write-through hand-off only (default)
    int absent_wg;     // fault injection (tests): >= 0: that workgroup of a cooperative-layout grid leaves at once, as if it had never become resident; -1 (default): none
    int hex;           // 1: launches that fill every two-wave team slot of the device run one six-team workgroup per CU (default), 0: two-team workgroups always
};
constexpr int COOP_BAR_WORDS = 4;
struct KArgs {
    int H, P, m, G;
    int B;                     // instances in this launch (set by the launcher)
    float invP;
    ModelK M;
    CostK C;
    ApgK A;
    // device tables (owned by the handle)
    const float* dt;    // [H]
    const float* sdt;   // [H][6]  sigma_i * sqrt(dt_t)
    const float* disc;  // [H+1]   discount^t / H
    const float* beta;  // [max_iter+2] momentum table
    const float* wts;   // blob float payload
    // per call
    const float* x0;           // [B][13]
    const float* u;            // [B][H][m]   control sequence (rollout/grad) or warm start (solve)
    const float* xref;         // [B][H+1][13]
    const float* noise;        // [B][G][H][6][32]
    const float* stepsize_in;  // [B] (solve)
    float* traj;               // [B][G][H+1][13][32] workspace
    float* act;                // [B][G][H][ACT_STRIDE] activation checkpoint of the gradient's forward sweep
    float* part;               // [B][G][part_stride(H)] per-group particle sums (adjoint outputs per step / mean trajectory), SPEC.md §6.1
    float* cost;               // [B]
    float* grad;               // [B][H][m]
    float* xmean;              // [B][H+1][13] or null
    float* uopt;               // [B][H][m]
    float* info;               // [B][8] raw: sum_ls, stepsize, nit, grad_sqr, sum_s, c_init, c_opt, nls_total
    int store_traj;
    int ws_rows;               // rows (instances or team slots) of traj / act / part / ustg the handle has allocated (launchers refuse a grid that needs more)
    int f16;                   // mlp_dtype: 0 f32; 1 fp16-operand MLP contractions in the forward step (SPEC.md §9); 2 layer-2 / W2^T contractions as three-limb bf16 splits (§9b)
    // cooperative latency path (one instance over coop_nwg workgroups; workspace owned by the handle, see sdempc_api.cpp)
    int coop_nwg;
    int coop_ngrp;             // speculative variant: groups of coop_nwg workgroups per instance (2..5)
    unsigned* coop_bar;        // [B][COOP_BAR_WORDS]: grid-barrier counter, error flag, arrivals of the streamed hand-off, pad (zeroed before every launch)
    float* ustg;               // [B][H][36] per-step control table in global memory (long horizons: keeps it out of LDS), or NULL
    unsigned coop_spin;        // time one grid barrier may wait before it gives up, in ticks of the 100 MHz s_memrealtime clock (10 ns)
    float* coop_pp;            // [B][2][part_stride(H)][G*32]
    float* coop_ck;            // [B][P][H+1][160]
    unsigned long long* work;  // [4] cumulative work of sdempc_solve_kernel launches: solves, gradient evaluations, forward-only rollouts, spare;
                               // [4] (as unsigned) the instance ticket word of the persistent launches (see ticket_base)
    int tickets;               // persistent launches: 1 = instances beyond the grid's first ones are handed out by the ticket word (set by the launcher)
    unsigned ticket_base;      // value of the ticket word when this launch starts (set by the launcher from *ticket_host)
    unsigned* ticket_host;     // HOST memory, owned by the handle: running total of the ticket word (it is never reset: every ticketed launch of B
                               // instances advances it by exactly B — B - S successful draws and one failing draw by each of the S teams)
    int fast;                  // SPEC.md §10: hardware transcendentals (selects the fastm translation unit; host-side switch)
    LaunchOpts opt;
};

// floats per (instance, group) row of KArgs::part: max(H*12 adjoint sums, (H+1)*13 state sums) + the group's cost total in the
// last element; multiple of 4
__host__ __device__ inline int part_stride(int H) { const int a = H * 12, b = (H + 1) * 13; return ((a > b ? a : b) + 1 + 3) & ~3; }
// Activation checkpoint of the gradient's forward sweep, floats per (instance, group, step): h2 tile (4 chunks x 64 lanes x 4) + step
// scalars (32 x 8) [+ SDEMPC_CKPT1 further layer-1 tiles of 1024 floats: 1 = drift h1, 2 = drift and density h1 (build-time experiment)]
#ifndef SDEMPC_CKPT1
#define SDEMPC_CKPT1 0
#endif
constexpr int ACT_STRIDE = 1280 + 1024 * SDEMPC_CKPT1;
size_t smem_bytes(int H, int m, int ipb);   // ipb: instances (teams) per workgroup
// host function pointer of the kernel the calling thread launched last through the launch_* functions below (for sdempc_last_kernel_name)
void note_kernel(const void* host_fn);
const void* last_launched_kernel();
// Cooperative latency path of the solve (exact f32, P >= 2): workgroups per instance, workspace sizes, launcher.
// coop_max_instances: how many instances fit one workgroup per CU on the current device (0 = path unavailable for this shape)
int coop_nwg(int P);
int coop_max_instances(int P, int H, int m, const LaunchOpts& o);
size_t coop_pp_floats(int H, int G);           // per instance
size_t coop_ck_floats(int H, int P);           // per instance
hipError_t launch_solve_coop(const KArgs& a, int B, hipStream_t st);
// speculative variant for the smallest batches (4 groups of workgroups per instance: two trials and two candidate gradients at once)
int spec_max_instances(int P, int H, int m, const LaunchOpts& o);
hipError_t launch_solve_spec(const KArgs& a, int B, hipStream_t st);
int team_ipb(int G, int H, int m);            // 4 when one wave owns an instance (G == 1 and LDS permits), else 1
hipError_t launch_rollout(const KArgs& a, int B, hipStream_t st);
hipError_t launch_grad(const KArgs& a, int B, hipStream_t st);
hipError_t launch_solve(const KArgs& a, int B, hipStream_t st);
// rows of KArgs::traj / act / part / ustg a solve launch of B instances indexes (B, or the team slots of a persistent launch)
int solve_workspace_rows(const KArgs& a, int B);
int solve_workspace_rows_fast(const KArgs& a, int B);
// math_mode fast (SPEC.md §10): the same kernels built with hardware transcendentals (second translation unit)
hipError_t launch_rollout_fast(const KArgs& a, int B, hipStream_t st);
hipError_t launch_grad_fast(const KArgs& a, int B, hipStream_t st);
hipError_t launch_solve_fast(const KArgs& a, int B, hipStream_t st);
hipError_t launch_solve_coop_fast(const KArgs& a, int B, hipStream_t st);      // the cooperative latency layouts in math_mode fast (same grid rules)
hipError_t launch_solve_spec_fast(const KArgs& a, int B, hipStream_t st);
// canonical [B][P][C] <-> device [B][G][C][32] (to_dev: zero-pads particles >= P)
hipError_t launch_relayout(bool to_dev, const float* in, float* out, int B, int P, int G, int C, hipStream_t st);
// SPEC.md §7: noise of B instances from their threefry keys (device u32[B][2]) straight into the device layout [B][G][H][6][32]
hipError_t launch_noise_from_keys(const uint32_t* keys_dev, float* out, int B, int P, int G, int H, hipStream_t st);

}  // namespace sdempc
