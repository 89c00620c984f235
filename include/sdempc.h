/*
 * sdempc.h — C ABI of the MI355X-native MPC inner loop (neural-SDE rollout + APG trajectory optimiser).
 *
 * This is the drop-in boundary for ONE path of wuwushrek/sde4mbrl_px4: the solver that
 * `sde4mbrl_px4/mpc_controller/sde_control.py` obtains from
 * `load_mpc_from_cfgfile(mpc_dir, convert_to_enu=True)` (sde_control.py:685) and calls per control
 * tick as `m_reset(x=, rng=, xdes=)` (sde_control.py:345-346,389-394,706) and
 * `m_mpc(x, rng, opt_state, curr_t=, xdes=)` (sde_control.py:349-350,400-416,717).
 * The reference has no C interface for this path (the arithmetic lives in the un-vendored JAX package
 * sde4mbrl); this header is what a ctypes binding on the reference side binds.
 *
 * Conventions
 *   - plain pointers + sizes, no C++ / torch types; every entry point returns 0 on success or a
 *     negative SDEMPC_E* code, never aborts or throws (an exception would silently kill the
 *     reference's `mpc_process`, sde_control.py:365-419 has no try/except).
 *   - state vector f32[13] = [x,y,z, vx,vy,vz, qw,qx,qy,qz, wx,wy,wz] (sde_control.py:246,747).
 *   - "host" entry points take host pointers (caller owns them) and stage through device buffers
 *     owned by the handle; "_dev" entry points take device pointers already resident in HBM and a
 *     hipStream_t passed as void*.
 *   - canonical tensor layouts (row-major, innermost last):
 *       x0    f32[B][13]            initial states
 *       u     f32[B][H][m]          control sequences (normalised PWM)
 *       xref  f32[B][H+1][13]       reference states at t_0..t_H
 *       noise f32[B][P][H][6]       N(0,1) draws for the 6 noisy state dims (v, omega)
 *       traj  f32[B][P][H+1][13]    particle x horizon tensor
 *       uopt  f32[B][H][m], xevol f32[B][H+1][13] (particle mean), info f32[B][8]
 *   - a handle is single-threaded; the HIP context is created lazily on the first call that needs
 *     the device, in the calling process (safe to create the handle before fork()).
 */
#ifndef SDEMPC_H
#define SDEMPC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Version of this header's struct layouts and entry points; sdempc_abi_version() returns the one the library was built with. A binding
 * compares the two before it passes a struct (2: sdempc_cfg grew the state_constr fields, handle options, work counters). */
#define SDEMPC_ABI_VERSION 2

#define SDEMPC_NX 13          /* state dims */
#define SDEMPC_NNOISE 6       /* noisy state dims: v(3), omega(3) */
#define SDEMPC_MAX_MOTORS 8
#define SDEMPC_HID 32         /* hidden width of the residual / density MLPs */
#define SDEMPC_BLOB_MAGIC 0x31454453 /* "SDE1" */
#define SDEMPC_BLOB_HEADER_INTS 16
#define SDEMPC_BLOB_FLOATS 2120

/* error codes */
#define SDEMPC_OK 0
#define SDEMPC_EINVAL (-1)    /* bad argument / config */
#define SDEMPC_EBLOB (-2)     /* malformed model blob */
#define SDEMPC_EDEVICE (-3)   /* HIP error (message in sdempc_last_error) */
#define SDEMPC_ENOMEM (-4)
#define SDEMPC_ECAPACITY (-5) /* batch larger than max_batch given at create */

/* Hyper-parameters: one-to-one with the reference's MPC YAML schema
 * (launch/iris_sitl_traj_mpc.yaml:8-85, launch/iris_sitl_posctrl_mpc.yaml:6-101). */
typedef struct sdempc_cfg {
    int32_t struct_size;              /* sizeof(sdempc_cfg), ABI check */
    int32_t horizon;                  /* H            (yaml: horizon) */
    int32_t num_particles;            /* P            (yaml: num_particles) */
    int32_t num_motors;               /* m            (len(input_constr.input_id)) */
    const float* time_steps;          /* [H] dt per step (yaml: num_short_dt/short_step_dt/long_step_dt) */
    float discount;                   /* yaml: discount */
    /* cost_params */
    float uref[SDEMPC_MAX_MOTORS];
    float uerr;
    float perr[3], verr[3], qerr[3], werr[3];
    float res_mult;
    float u_slew_coeff;
    int32_t has_slew_constr;          /* 1 if cost_params.u_slew_constr present */
    float u_slew_lo[SDEMPC_MAX_MOTORS], u_slew_hi[SDEMPC_MAX_MOTORS];
    float u_slew_constr_coeff;
    /* input_constr.input_bound (enforce_ubound) */
    float u_lo[SDEMPC_MAX_MOTORS], u_hi[SDEMPC_MAX_MOTORS];
    /* apg_mpc */
    int32_t max_iter;
    int32_t max_no_improvement_iter;
    int32_t use_moment_scale;         /* 0: yaml moment_scale null */
    float moment_scale;
    float beta_init;
    float atol, rtol;
    float stepsize;                   /* used when ls_maxls == 0 */
    float ls_init_stepsize, ls_max_stepsize, ls_coef, ls_decrease_factor, ls_increase_factor;
    int32_t ls_reset_option;          /* 0 conservative, 1 increase */
    int32_t ls_maxls;
    /* extension (not a reference YAML key): arithmetic of the MLP contractions. All three are reproduced bit for bit by the CPU oracle.
     * 0 = f32 fma chains (v_mfma_f32_32x32x2_f32; default);
     * 1 = fp16 operands rounded toward zero, f32 accumulate, in the forward rollout (v_mfma_f32_32x32x16_f16; BASELINE config C5; SPEC.md §9);
     * 2 = f32x3: f32 operands, the two 32x32 contractions of a step (layer 2 of the drift net and its transpose in the adjoint) evaluated as
     *     three-limb bf16 splits of both operands on v_mfma_f32_32x32x16_bf16 — f32-level accuracy on the matrix pipe (SPEC.md §9b). */
    int32_t mlp_dtype;
    /* extension (not a reference YAML key): 0 = exact (default): tanh / sigmoid / reciprocal square root in the bit-reproducible
     * software forms of SPEC.md §3, results identical to the CPU oracle bit for bit. 1 = fast: the same kernels with the
     * hardware transcendentals (v_exp_f32, v_rcp_f32, v_rsq_f32) and the hidden activation kept as 1 / (1 + 2^a') with its affine maps folded into the
     * weights by sdempc_create (SPEC.md §10, §10b); about 1e-7 relative per operation away from the exact path, and — through the oracle's model of the three
     * instructions (SPEC.md §10a) — compared with the CPU oracle bit for bit as well. */
    int32_t math_mode;
    /* state_constr (launch/iris_sitl_traj_mpc.yaml:16-29; commented out in every YAML the reference ships), penalty form
     * (slack_proximal: False): stage cost += sum_k state_w[k] * (max(0, x[id_k] - hi_k)^2 + max(0, lo_k - x[id_k])^2) at x_{t+1},
     * state_w = state_penalty * constr_pen (host float32), ids strictly ascending, in the solver's frame. SPEC.md §5.3. */
    int32_t num_state_constr;         /* 0: none */
    int32_t state_id[SDEMPC_NX];
    float state_w[SDEMPC_NX], state_lo[SDEMPC_NX], state_hi[SDEMPC_NX];
} sdempc_cfg;

/* Optimiser telemetry: the 7 scalars the reference reads from opt_state
 * (sde_control.py:444-450, msg/OptMPCState.msg:6-22) + the line-search trial count. */
typedef struct sdempc_info {
    float avg_linesearch;
    float stepsize;
    float num_steps;
    float grad_sqr;
    float avg_stepsize;
    float init_cost;
    float opt_cost;
    float num_ls_trials;              /* total trial rollouts N_ls (for the roofline accounting) */
} sdempc_info;

typedef struct sdempc_handle sdempc_handle;

/* ---- lifetime ----------------------------------------------------------------------------- */
/* Replaces: construction inside load_mpc_from_cfgfile (sde_control.py:685). Host-only; no HIP call. */
int sdempc_create(const sdempc_cfg* cfg, const void* model_blob, size_t blob_bytes,
                  int32_t max_batch, sdempc_handle** out);
void sdempc_destroy(sdempc_handle* h);
const char* sdempc_last_error(const sdempc_handle* h);   /* h may be NULL: last create error */
int sdempc_abi_version(void);
/* What the library was built with (no reference counterpart). Bit 0 (SDEMPC_BUILD_ALL_VARIANTS): the build carries every kernel instantiation
 * (`make EXTRA=-DSDEMPC_ALL_VARIANTS=1`): the generic motor count (m other than 4 / 6) in the duo / six-team / cooperative / speculative layouts
 * too, and the packed-tanh instantiations behind SDEMPC_OPT_PK = 1. The default build runs a generic motor count in the one-group-per-wave tile
 * layouts (and P = 1 in the lane layout) and refuses SDEMPC_OPT_PK = 1; results never depend on the layout. */
#define SDEMPC_BUILD_ALL_VARIANTS 1
int sdempc_build_flags(void);

/* Binds the handle to a HIP device ordinal (default 0). Must precede the first device call. */
int sdempc_set_device(sdempc_handle* h, int32_t device);

/* 1 once the handle has initialised the GPU (first device call), else 0. Host-only, no HIP call: lets the caller of a forked
 * process (the reference forks mpc_process after building its solvers, sde_control.py:69-75,723-728) tell a handle that is safe
 * to use from one whose HIP state belongs to the parent and must be abandoned without sdempc_destroy. */
int sdempc_device_ready(const sdempc_handle* h);

/* ---- execution options (per handle) ---------------------------------------------------------
 * How a call is laid out on the GPU; none of them changes a bit of any result. No reference counterpart (the reference's
 * solver objects, sde_control.py:681-721, have no such knobs). The environment variables named below only give the DEFAULT of
 * a handle created afterwards (read once inside sdempc_create); nothing on the launch path reads the environment.
 *   key                       values                         default  environment default
 *   SDEMPC_OPT_LANE           0 / 1                          1        SDEMPC_LANE          P == 1 instances in the single-particle lane layout
 *   SDEMPC_OPT_COOP           0 / 1                          1        SDEMPC_COOP          small batches spread over many workgroups (latency layouts);
 *                                                                                          setting 1 also re-arms a handle that fell back (sdempc_layout_fallbacks)
 *   SDEMPC_OPT_SPEC           0 / 1                          1        SDEMPC_SPEC          speculative variant of the cooperative layout (smallest batches)
 *   SDEMPC_OPT_PK             -1 auto / 0 / 1                -1       SDEMPC_PK            packed-f32 tanh instantiation of the tile layout (auto: grid <= CUs)
 *   SDEMPC_OPT_USTG           -1 auto / 0 / 1                -1       SDEMPC_USTG          per-step control table in global memory instead of LDS (auto: long horizons)
 *   SDEMPC_OPT_DUO            -1 auto / 0 / 1                -1       SDEMPC_DUO           throughput launches: 64 particles per wave (two 32-particle groups; auto = on for P > 32)
 *   SDEMPC_OPT_HEX            0 / 1                          1        SDEMPC_HEX           launches that fill every two-wave team slot of the device: one six-team
 *                                                                                          workgroup per CU (weights staged once per CU) instead of three two-team ones
 *   SDEMPC_OPT_COOP_LAUNCH    0 / 1                          0        SDEMPC_COOP_LAUNCH   hipLaunchCooperativeKernel for the cooperative layouts
 *   SDEMPC_OPT_COOP_FENCE     0 / 1                          0        SDEMPC_COOP_FENCE    agent-scope release / acquire fences around the grid barrier and the arrival counter
 *   SDEMPC_OPT_COOP_SPIN_US   -1 derived / >= 0 microseconds -1       SDEMPC_COOP_SPIN_US  how long one grid barrier / polled hand-off of a cooperative layout may wait
 *                                                                                          before the launch gives up (derived: 5 x the handle's last completed
 *                                                                                          cooperative solve, clamped to 2..100 ms; 100 ms before the first)
 *   SDEMPC_OPT_TEST_ABSENT_WG -1 none / >= 0 workgroup index -1       (none)               FAULT INJECTION for the tests of the bounded waits: that workgroup of a
 *                                                                                          cooperative-layout grid leaves at once, as a workgroup that never became
 *                                                                                          resident would; the launch then gives up within its spin budget
 *   SDEMPC_OPT_DEVICE_CUS     read-only                                                    compute units of the handle's device (after the first device call)
 */
#define SDEMPC_OPT_LANE 1
#define SDEMPC_OPT_COOP 2
#define SDEMPC_OPT_SPEC 3
#define SDEMPC_OPT_PK 4
#define SDEMPC_OPT_USTG 5
#define SDEMPC_OPT_COOP_LAUNCH 6
#define SDEMPC_OPT_COOP_FENCE 7
#define SDEMPC_OPT_COOP_SPIN_US 8
#define SDEMPC_OPT_DEVICE_CUS 9
#define SDEMPC_OPT_DUO 10
#define SDEMPC_OPT_HEX 11
#define SDEMPC_OPT_TEST_ABSENT_WG 12
int sdempc_set_option(sdempc_handle* h, int32_t key, int32_t value);
int sdempc_get_option(const sdempc_handle* h, int32_t key, int32_t* value);

/* ---- m_reset (sde_control.py:702-707,345-346,389-394) -------------------------------------- */
/* Host-only. Fills yk[H][m] with the hover guess uref and info with the initial telemetry. */
int sdempc_reset(sdempc_handle* h, const float* x, const float* xdes, float* yk, sdempc_info* info);

/* ---- hot-path pieces, batched, host pointers ----------------------- */
/* A4+A5: Euler–Maruyama rollout + expected cost. traj/xmean may be NULL. */
int sdempc_rollout_batch(sdempc_handle* h, int32_t B, const float* x0, const float* u,
                         const float* xref, const float* noise,
                         float* cost /*[B]*/, float* traj /*[B][P][H+1][13] or NULL*/,
                         float* xmean /*[B][H+1][13] or NULL*/);
/* A6 (gradient): cost and d cost / d u by the adjoint pass. */
int sdempc_grad_batch(sdempc_handle* h, int32_t B, const float* x0, const float* u,
                      const float* xref, const float* noise,
                      float* cost /*[B]*/, float* grad /*[B][H][m]*/);
/* A3: the solve. u_init = warm start (opt_state.yk), stepsize_in = opt_state.stepsize.
 * Execution layout is chosen per call and never changes a bit of the result: P = 1 instances (all YAMLs the reference ships) run in a
 * single-particle layout; batches small enough that all workgroups are resident at once (C2: up to 15 instances) are spread over many
 * workgroups, one particle per wave, with one bounded grid barrier per rollout, and for the smallest batches additionally evaluate up to three
 * line-search trials and the candidate gradients of the next iteration at once, handing the per-particle outputs over as tagged words that the
 * workgroups poll (bounded like the barrier) instead of through a barrier (C2 single solve: 20 ms instead of 158 ms); larger
 * batches run one workgroup per instance in the 32-particle MFMA tile layout (throughput). The cooperative layouts assume that no other
 * kernel occupies the GPU while they run; if their workgroups cannot all become resident the barrier gives up after a bounded time
 * (SDEMPC_OPT_COOP_SPIN_US) and the telemetry of the launch is NaN. The host-pointer entry points then run the same batch once more in the
 * one-workgroup-per-instance layout (identical results) and the handle stays off the cooperative layouts from then on
 * (sdempc_layout_fallbacks counts these events); callers of sdempc_solve_batch_dev ask sdempc_solve_status. */
int sdempc_solve_batch(sdempc_handle* h, int32_t B, const float* x0, const float* xref,
                       const float* noise, const float* u_init /*[B][H][m]*/,
                       const float* stepsize_in /*[B]*/,
                       float* uopt /*[B][H][m]*/, float* xevol /*[B][H+1][13]*/,
                       sdempc_info* info /*[B]*/);

/* ---- device-resident variants (benchmarks, multi-instance serving) -------------------------- */
/* noise_dev uses the device layout produced by sdempc_noise_to_device_layout / _dev:
 *   f32[B][G][H][6][32] with G = ceil(P/32), particle p -> (g = p/32, lane = p%32). */
size_t sdempc_noise_dev_floats(const sdempc_handle* h, int32_t B);
size_t sdempc_traj_dev_floats(const sdempc_handle* h, int32_t B);
int sdempc_noise_to_device_layout(const sdempc_handle* h, int32_t B, const float* noise_host, float* out_host);
/* The same conversion on the device: canonical f32[B][P][H][6] at noise_canonical_dev -> device layout at
 * noise_out_dev (sdempc_noise_dev_floats(h, B) floats; padded particles are written as zeros). */
int sdempc_noise_to_device_layout_dev(sdempc_handle* h, int32_t B, const void* noise_canonical_dev,
                                      void* noise_out_dev, void* stream);
/* Copies the particle x horizon tensor that the last sdempc_rollout_batch_dev(store_traj=1)
 * or sdempc_grad_batch_dev of the first B instances left in the handle to traj_out_dev as canonical
 * f32[B][P][H+1][13]. Enqueue it on the stream of that call. */
int sdempc_traj_to_canonical_dev(sdempc_handle* h, int32_t B, void* traj_out_dev, void* stream);
int sdempc_solve_batch_dev(sdempc_handle* h, int32_t B, const void* x0_dev, const void* xref_dev,
                           const void* noise_dev, const void* u_init_dev, const void* stepsize_dev,
                           void* uopt_dev, void* xevol_dev, void* info_dev /*f32[B][8]*/,
                           void* stream);
int sdempc_rollout_batch_dev(sdempc_handle* h, int32_t B, const void* x0_dev, const void* u_dev,
                             const void* xref_dev, const void* noise_dev, void* cost_dev,
                             void* xmean_dev /*or NULL*/, int32_t store_traj, void* stream);
int sdempc_grad_batch_dev(sdempc_handle* h, int32_t B, const void* x0_dev, const void* u_dev,
                          const void* xref_dev, const void* noise_dev, void* cost_dev,
                          void* grad_dev, void* stream);

/* ---- key-derived noise (SPEC.md §7) ---------------------------------------------------------- */
/* The reference threads a JAX PRNG key through every solver call: jax.random.PRNGKey(seed) and its 3-way split
 * (sde_control.py:338-341), `rng` in and out of m_reset / m_mpc (sde_control.py:345-350,400-416,698,706,717). A key is
 * uint32[2] with JAX's threefry2x32 conventions (PRNGKey(seed) = {seed >> 32, seed & 0xffffffff}; the host-side split lives in
 * sde4mbrl_px4_amd/prng.py). These entry points draw the noise tensor of instance b as normal(keys[b], (P, H, 6)) on the
 * device, so only 8 bytes per instance cross the boundary. keys is a HOST pointer, u32[B][2], in all three. */
int sdempc_noise_from_keys_dev(sdempc_handle* h, int32_t B, const uint32_t* keys, void* noise_out_dev /* device layout,
                               sdempc_noise_dev_floats(h, B) floats */, void* stream);
int sdempc_noise_from_keys(sdempc_handle* h, int32_t B, const uint32_t* keys, float* noise /* host, canonical [B][P][H][6] */);
/* A3 with key-derived noise: what m_mpc(x, rng, opt_state, ...) maps to. */
int sdempc_solve_batch_keys(sdempc_handle* h, int32_t B, const float* x0, const float* xref, const uint32_t* keys,
                            const float* u_init /*[B][H][m]*/, const float* stepsize_in /*[B]*/,
                            float* uopt /*[B][H][m]*/, float* xevol /*[B][H+1][13]*/, sdempc_info* info /*[B]*/);

/* After the stream of the last sdempc_solve_batch_dev call has been synchronised: SDEMPC_OK, or SDEMPC_EDEVICE when a grid barrier of
 * a cooperative layout gave up (results of that call invalid, telemetry NaN). The handle then stays off the cooperative layouts, so
 * repeating the call runs in the one-workgroup-per-instance layout. Also SDEMPC_EDEVICE when a large throughput launch that hands its
 * instances out by ticket (three rounds of the persistent grid or more) ended with a ticket count other than the one the host expects:
 * instances of that launch may then be unsolved (the handle re-synchronises itself; repeat the call). Such launches are issued one at a
 * time per handle, in order, and cannot be captured into a hipGraph (the launch carries the ticket base of its own moment). New in this
 * build (no reference counterpart: the reference's solver call, sde_control.py:405-416, either returns or kills the process). */
int sdempc_solve_status(sdempc_handle* h);
/* Number of times this handle left the cooperative layouts because a grid barrier gave up (0 in normal operation). */
int32_t sdempc_layout_fallbacks(const sdempc_handle* h);

/* Work the solve launches of this handle have actually done since creation (or the last reset), summed over instances:
 * out[0] solves, out[1] gradient evaluations (forward + adjoint sweep), out[2] forward-only rollouts (line-search trials, the
 * initial-cost and the final mean-trajectory rollout), out[3] reserved. An iteration whose extrapolation point did not move re-uses
 * its gradient instead of evaluating it again, so out[1] can be below the sum of the iteration counts. Counted by the
 * one-team-per-instance and plain cooperative layouts (the speculative latency layout evaluates extra, discarded rollouts and is
 * not counted). Call after the stream of the last solve has been synchronised. Roofline accounting of bench.py. */
int sdempc_work_counters(sdempc_handle* h, uint64_t out[4], int32_t reset);

/* Name of the kernel instantiation the last *_dev launch of this handle started, as a profiler prints it (demangled, without the
 * argument list), e.g. "sdempc::exact::sdempc_solve_kernel<sdempc::exact::TeamBlock, 4, false, false, 0, false>"; empty before the first
 * launch. For bench.py's roofline record and for matching rocprofv3 kernel traces. */
int sdempc_last_kernel_name(const sdempc_handle* h, char* buf, size_t n);

/* Times the last *_dev launch on its own stream with HIP events (ms); <0 if unavailable. */
float sdempc_last_kernel_ms(const sdempc_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* SDEMPC_H */
