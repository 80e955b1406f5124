/* sgmcmc_hip.h -- C ABI of libsgmcmc_hip.so, the MI355X (gfx950) SG-MCMC update path.
 *
 * This is the drop-in boundary for the per-parameter update executed by
 * `next(sampler)` in MFreidank/pysgmcmc. The reference has no FFI: its boundary
 * is `session.run([theta_t, cost])` (pysgmcmc/samplers/base_classes.py:298-300,
 * :438-441), which executes ~25 TensorFlow elementwise ops per parameter tensor.
 * Each entry point below replaces one such op chain with ONE fused HIP kernel
 * launch over a flat array of all parameters.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no C++/torch types.
 *   - Every function returns 0 on success, a positive hipError_t value if the
 *     HIP runtime failed, or a negative SGMCMC_E* code. sgmcmc_last_error()
 *     returns a thread-local message for the last failure on this thread.
 *   - The library never allocates, frees or copies caller memory. All array
 *     arguments are DEVICE pointers on the device that owns `stream`, n elements
 *     long. 16-byte-aligned arrays take the vector path (one dwordx4 per lane per
 *     array); any other alignment takes a slower scalar path with identical
 *     results.
 *   - Calls are asynchronous on `stream` (a hipStream_t passed as void*; NULL =
 *     the null stream) and are legal inside hipStreamBeginCapture.
 *   - No global mutable state at all: launch geometry is an optional per-call argument
 *     (`const sgmcmc_launch_t *launch`, NULL = the measured defaults), error text is thread-local.
 *     One host thread per chain / per GPU is safe, and so are several chains per thread.
 *   - Arithmetic: one IEEE rounding per reference op, in the reference's op order
 *     (built with -ffp-contract=off; '/' and sqrt correctly rounded), so with
 *     injected noise (`xi` != NULL) results are bit-identical to the CPU oracle
 *     (oracle/sgmcmc_oracle.c) in both f32 and f64.
 *   - Noise: `xi` == NULL draws xi[i] ~ N(0,1) in registers from Philox4x32-10 with
 *     counter = (step_lo, step_hi, quad_lo, quad_hi), quad = i / 4, key = (seed_lo,
 *     seed_hi), then Box-Muller: words (x0,x1) -> elements 4q (sin), 4q+1 (cos);
 *     (x2,x3) -> 4q+2, 4q+3. Identical to rocRAND's philox4x32_10 stream with
 *     subsequence = quad, offset = 4*step. The stream depends only on
 *     (seed, step, i): never on launch geometry, alignment path or device.
 */
#ifndef SGMCMC_HIP_H
#define SGMCMC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGMCMC_ABI_VERSION 6

#define SGMCMC_EINVAL   (-1)   /* null/invalid argument */
#define SGMCMC_ENODEV   (-2)   /* no HIP device / not gfx950 code object */

typedef void *sgmcmc_stream_t;     /* hipStream_t */

/* ---- Contract map (ABI v6): what a maintainer of the reference binds, and what is this build's own machinery ------------------
 * Every entry point below belongs to exactly one group (tests/test_boundary.py checks the lists against the declarations).
 *
 * [boundary]  SURVEY.md section 8(b): the update path behind `next(sampler)` and its one cross-chain exchange. THIS is the drop-in
 *   contract; INTEGRATION.md binds it from the reference's side.
 *     sgmcmc_abi_version sgmcmc_last_error sgmcmc_device_count
 *     sgmcmc_sghmc_step_f32 sgmcmc_sghmc_step_f64 sgmcmc_sgld_step_f32 sgmcmc_sgld_step_f64
 *     sgmcmc_rsghmc_step_f32 sgmcmc_rsghmc_step_f64
 *     sgmcmc_sghmc_scalars_f32 sgmcmc_sghmc_scalars_f64 sgmcmc_sgld_scalars_f32 sgmcmc_sgld_scalars_f64
 *     sgmcmc_rsghmc_scalars_f32 sgmcmc_rsghmc_scalars_f64 sgmcmc_counter_add_u64
 *     sgmcmc_step_stats_records sgmcmc_step_stats_workspace_bytes sgmcmc_step_stats_finish
 *     sgmcmc_philox_normal_f32 sgmcmc_philox_normal_f64 sgmcmc_philox_bits_u32
 *     sgmcmc_moments_update_f32 sgmcmc_moments_update_f64
 *     sgmcmc_rhat_pack_f32 sgmcmc_rhat_pack_f64 sgmcmc_rhat_finish_f32 sgmcmc_rhat_finish_f64
 *     sgmcmc_summary_workspace_bytes sgmcmc_summary_f32 sgmcmc_summary_f64
 *     sgmcmc_event_create sgmcmc_event_destroy sgmcmc_event_elapsed_ms sgmcmc_event_synchronize
 * [cost-path]  internals of the gradient PRODUCER (row a14: BNNCost's launch plans, the replacement of the TF cost graph of
 *   pysgmcmc/models/bayesian_neural_network.py:28-141,337-388). A maintainer who keeps another gradient producer binds none of
 *   them; every one is launched by at least one reachable plan (tests/test_bnn_dense_gpu.py walks the plans).
 *     sgmcmc_window_gather_f32 sgmcmc_window_gather_f64
 *     sgmcmc_bnn_dense_tanh_dot_parts sgmcmc_bnn_dense_tanh_f32 sgmcmc_bnn_dense_tanh_backward_f32 sgmcmc_colsum_finish_f32
 *     sgmcmc_bnn_planes_bytes sgmcmc_bnn_split_planes_f32 sgmcmc_bnn_gw_planes_f32
 *     sgmcmc_bias_tanh_f32 sgmcmc_bias_tanh_f64 sgmcmc_bias_tanh_rowdot_f32 sgmcmc_bias_tanh_rowdot_f64
 *     sgmcmc_bnn_head_f32 sgmcmc_bnn_head_f64 sgmcmc_bnn_head_last_layer_backward_f32 sgmcmc_bnn_head_last_layer_backward_f64
 *     sgmcmc_bnn_last_layer_backward_f32 sgmcmc_bnn_last_layer_backward_f64
 *     sgmcmc_tanh_backward_f32 sgmcmc_tanh_backward_f64 sgmcmc_tanh_backward_colsum_f32 sgmcmc_tanh_backward_colsum_f64
 * [whole-step]  problems that fit one workgroup or one lane: the whole `next(sampler)` step (or many) in one launch --
 *   BASELINE configs[1]'s 3 x 50 net and the reference's toy targets.
 *     sgmcmc_bnn_fused_sghmc_steps_f32 sgmcmc_bnn_fused_sghmc_steps_f64 sgmcmc_bnn_fused_sgld_steps_f32 sgmcmc_bnn_fused_sgld_steps_f64
 *     sgmcmc_toy_chains_f32 sgmcmc_toy_chains_f64
 * [svgd]  pysgmcmc/samplers/svgd.py -- OUT OF SCOPE of the hot path, built in round 1 and kept as is.
 *     sgmcmc_svgd_workspace_bytes sgmcmc_svgd_max_particles sgmcmc_svgd_step_f32 sgmcmc_svgd_step_f64
 *     sgmcmc_svgd_kernel_f32 sgmcmc_svgd_kernel_f64
 * ---- end of the contract map ---------------------------------------------------------------------------------------------------- */

int sgmcmc_abi_version(void);
const char *sgmcmc_last_error(void);

/* Number of HIP devices visible, or a negative code. Does not create a context. */
int sgmcmc_device_count(void);

/* Launch geometry of ONE call (performance only; results never depend on it). Pass NULL for the
 * defaults; a field set to 0 (nontemporal: -1) keeps its default.
 *   block_threads: 64, 128, 192, 256, or -1 = auto (default): 128 when one launch streams more
 *                than 640 MiB (HBM-resident working set), else 256
 *   quads_per_thread: 1, 2 or 4 float4 groups in flight per lane (default 1)
 *   max_blocks: grid cap, the kernel grid-strides beyond it (default 2^20 = uncapped)
 *   nontemporal: 0 plain, 1 nt loads+stores, 2 auto = nt iff one f32 launch touches more
 *                than 640 MiB, i.e. cannot live in the 256 MiB Infinity Cache; f64 launches
 *                stay plain (measured faster at every size) (default 2)                       */
typedef struct sgmcmc_launch {
    int block_threads;
    int quads_per_thread;
    int max_blocks;
    int nontemporal;
    /* Launch instrumentation (both NULL = none): hipEvent_t handles that receive the KERNEL's own start and stop
     * timestamps (hipExtLaunchKernel: taken from the dispatch packet, the duration rocprofv3 reports), instead of
     * the stream-position timestamps of hipEventRecord brackets, which add ~3 us of barrier-packet and dispatch
     * latency around a launch. Read them with sgmcmc_event_elapsed_ms() after the stream has been synchronised.   */
    void *start_event;
    void *stop_event;
} sgmcmc_launch_t;

/* hipEvent_t helpers for the instrumentation above (timing-enabled events of the CURRENT device).                 */
int sgmcmc_event_create(void **event_out);
int sgmcmc_event_destroy(void *event);
int sgmcmc_event_elapsed_ms(void *start_event, void *stop_event, float *ms_out);
int sgmcmc_event_synchronize(void *event);      /* host waits until the event has completed */

/* Optional extras of ONE step call (K1-K3); NULL = none of them. Unlike sgmcmc_launch_t these change WHAT the launch
 * does (never the update arithmetic of an element).
 *   first_element:  index of theta[0] within the chain's whole parameter vector (a multiple of 4). The Philox counter of
 *                element i is (first_element + i) / 4, so a step issued as several launches over consecutive SLICES of
 *                the arena -- each as soon as its part of the gradient exists, e.g. layer by layer on a second stream
 *                under the remaining backward GEMMs -- gives exactly the chain the single launch gives.
 *   stats_record_base / stats_record_total: with a stats workspace, block b of this launch writes statistics record
 *                stats_record_base + b and the workspace header is set to stats_record_total records (0 = this launch's
 *                own): the slices of one step share one workspace. sgmcmc_step_stats_records() gives a launch's count.
 *   stats_select:   0 = every statistic the operator produces; SGMCMC_STATS_THETA_SQ = only sum theta'^2 (the one the
 *                BNN loss head consumes; the other three are written as 0 and cost no reduction work).
 *   flags:       SGMCMC_STEP_HBM_RESIDENT: the launch is part of a working set larger than the Infinity Cache even if
 *                this slice alone is not (the auto geometry then picks what it picks for the whole arena).
 *                SGMCMC_STEP_SKIP_MINV_STORE (adapt = 1 only): do not write minv this step (44 instead of 48 B/param
 *                for K1). minv is only CONSUMED by frozen steps, which read what the LAST burn-in step wrote, so a
 *                caller may set this on every burn-in step but the last; minv then holds stale values in between.
 *   moments_mean / moments_m2 / moments_count: K4 fused into the step -- theta' is folded into the chain's Welford
 *                moments (arrays of n elements aligned with theta, count includes this sample) in the same pass:
 *                +16 B/param on that launch instead of a separate 20 B/param pass. Same arithmetic as
 *                sgmcmc_moments_update_* bit for bit (paths without a fused form run that kernel after the step).
 *   scalars_dev: NULL, or a DEVICE block of 5 elements of the dtype filled by sgmcmc_{sghmc,sgld,rsghmc}_scalars_*;
 *                the kernel then takes its derived scalars (stepsize etc.) from there instead of the by-value
 *                arguments. A hipGraph replays identical arguments: a captured step follows a stepsize SCHEDULE
 *                through this block (one graph per phase instead of one per stepsize).
 *   gather_*:    (ABI v6) the NEXT step's minibatch window as a side job of this launch -- what sgmcmc_window_gather_* does
 *                (pysgmcmc/data_batches.py:118-123: rows [gather_start, gather_start + gather_batch) of X [n_data][gather_dim] into
 *                gather_x_out [gather_batch][gather_x_out_ld], the same rows of y into gather_y_out), X / y of the step's dtype.
 *                A tiny launch of its own costs ~5 us on the device's timeline (dispatch, ramp); inside the step launch --
 *                the one launch of a graph-stepped chain that takes per-step arguments by value -- the copy is done by a
 *                few extra workgroups in front of the update's and costs nothing measurable. All NULL / 0 = none. Needs
 *                gather_dim * sizeof(T) and gather_x_out_ld * sizeof(T) multiples of 16 and a 16-byte aligned source window
 *                and destination (SGMCMC_EINVAL otherwise: use sgmcmc_window_gather_*); the feed buffers must not be read by
 *                anything still running on the stream behind this launch's predecessor. Launch variants without a fused
 *                form (grid-capped, element-wise) run the copy as a small launch of its own behind the step.          */
#define SGMCMC_STATS_THETA_SQ        1
#define SGMCMC_STEP_HBM_RESIDENT     1u
#define SGMCMC_STEP_SKIP_MINV_STORE  2u
typedef struct sgmcmc_step_opts {
    uint64_t first_element;
    uint32_t stats_record_base;
    uint32_t stats_record_total;
    int stats_select;
    unsigned flags;
    void *moments_mean;
    void *moments_m2;
    uint64_t moments_count;
    const void *scalars_dev;
    const void *gather_x;
    const void *gather_y;
    void *gather_x_out;
    void *gather_y_out;
    uint64_t gather_start;
    uint32_t gather_batch;
    uint32_t gather_dim;
    uint32_t gather_x_out_ld;
    uint32_t reserved0;           /* 0 */
} sgmcmc_step_opts_t;

/* Number of statistics records (one per block = the grid) a vector-path step launch of n elements writes under
 * `launch` (block_threads must be explicit there: 64..256), or 0 on invalid arguments.                          */
size_t sgmcmc_step_stats_records(size_t n, const sgmcmc_launch_t *launch);

/* K1 -- fused SGHMC step. Replaces the op chain pysgmcmc/samplers/sghmc.py:165-251
 * (+ constants :111-117) and the burn-in switch pysgmcmc/samplers/base_classes.py:432-456.
 *   adapt = 1  burn-in step (is_burning_in): reads theta,V,grad,tau,g,v_hat; writes
 *              theta,V,tau,g,v_hat,minv (and r if non-NULL).         48 B/param f32
 *   adapt = 0  frozen step: reads theta,V,grad,minv; writes theta,V. 24 B/param f32
 *              (tau,g,v_hat,r may be NULL)
 *   eps, scale_grad, mdecay: the constructor/schedule scalars; eps_scaled and the
 *              noise-scale constants are derived inside, in the dtype.
 *   grad_decay: the kernel uses grad[i] + grad_decay * theta[i] as the gradient (formed in
 *              registers from values it already loads): a Gaussian weight-prior / weight-decay
 *              term of the cost need not be added by the gradient producer. 0 = off (exactly
 *              the reference's update).
 *   xi:   NULL -> Philox(seed, step); else injected N(0,1) draws, n elements.
 *   step: the sampler's n_iterations at the time of the call.
 *   stats_ws:  NULL, or a device workspace of sgmcmc_step_stats_workspace_bytes(n) bytes: the
 *              kernel also reduces, from the values it already holds in registers (wave shuffles
 *              -> LDS -> ONE 32-byte record per block in stats_ws, block-major), the sums {theta'^2, V'^2
 *              (p'^2; 0 for SGLD), minv, minv^2}. sgmcmc_step_stats_finish() then adds the
 *              partials in a fixed order into 4 doubles. Bit-reproducible for a given launch
 *              geometry; costs no extra pass over HBM.
 *   step_dev: NULL, or a DEVICE counter added to `step` when the kernel starts.
 *              A hipGraph replays identical arguments; a graph-captured chain
 *              advances its noise stream with sgmcmc_counter_add_u64 on it.        */
int sgmcmc_sghmc_step_f32(float *theta, float *V, const float *grad,
                          float *tau, float *g, float *v_hat, float *minv, float *r,
                          size_t n, float eps, float scale_grad, float mdecay, float grad_decay, int adapt,
                          const float *xi, uint64_t seed, uint64_t step, const uint64_t *step_dev,
                          void *stats_ws, const sgmcmc_step_opts_t *opts, const sgmcmc_launch_t *launch,
                          sgmcmc_stream_t stream);
int sgmcmc_sghmc_step_f64(double *theta, double *V, const double *grad,
                          double *tau, double *g, double *v_hat, double *minv, double *r,
                          size_t n, double eps, double scale_grad, double mdecay, double grad_decay, int adapt,
                          const double *xi, uint64_t seed, uint64_t step, const uint64_t *step_dev,
                          void *stats_ws, const sgmcmc_step_opts_t *opts, const sgmcmc_launch_t *launch,
                          sgmcmc_stream_t stream);

/* Host-derived scalars of a step as a DEVICE block (sgmcmc_step_opts_t.scalars_dev): derived exactly as the step call
 * derives them from its by-value arguments, stored by a 1-thread kernel on `stream`.                              */
int sgmcmc_sghmc_scalars_f32(float eps, float scale_grad, float mdecay, void *scalars_dev, sgmcmc_stream_t stream);
int sgmcmc_sghmc_scalars_f64(double eps, double scale_grad, double mdecay, void *scalars_dev, sgmcmc_stream_t stream);
int sgmcmc_sgld_scalars_f32(float eps, float A, float scale_grad, void *scalars_dev, sgmcmc_stream_t stream);
int sgmcmc_sgld_scalars_f64(double eps, double A, double scale_grad, void *scalars_dev, sgmcmc_stream_t stream);
int sgmcmc_rsghmc_scalars_f32(float eps, float mass, float c, float D, float b_hat, void *scalars_dev, sgmcmc_stream_t stream);
int sgmcmc_rsghmc_scalars_f64(double eps, double mass, double c, double D, double b_hat, void *scalars_dev,
                              sgmcmc_stream_t stream);

/* Workspace: 32-byte header {uint64 record count} + one 32-byte record {4 doubles} per block.                  */
size_t sgmcmc_step_stats_workspace_bytes(size_t n);
/* K7 -- stats_out[0..3] (device doubles) = fixed-order sum of the per-block records a step kernel left
 * in stats_ws. One 1024-lane block.                                                               */
int sgmcmc_step_stats_finish(const void *stats_ws, double *stats_out, sgmcmc_stream_t stream);

/* K2 -- fused preconditioned SGLD step. Replaces pysgmcmc/samplers/sgld.py:149-211.
 *   adapt = 1: R{theta,grad,tau,g,v_hat} W{theta,tau,g,v_hat,minv}   40 B/param f32
 *   adapt = 0: R{theta,grad,minv} W{theta}                           16 B/param f32 */
int sgmcmc_sgld_step_f32(float *theta, const float *grad,
                         float *tau, float *g, float *v_hat, float *minv, float *r,
                         size_t n, float eps, float A, float scale_grad, float grad_decay, int adapt,
                         const float *xi, uint64_t seed, uint64_t step, const uint64_t *step_dev,
                          void *stats_ws, const sgmcmc_step_opts_t *opts, const sgmcmc_launch_t *launch,
                          sgmcmc_stream_t stream);
int sgmcmc_sgld_step_f64(double *theta, const double *grad,
                         double *tau, double *g, double *v_hat, double *minv, double *r,
                         size_t n, double eps, double A, double scale_grad, double grad_decay, int adapt,
                         const double *xi, uint64_t seed, uint64_t step, const uint64_t *step_dev,
                          void *stats_ws, const sgmcmc_step_opts_t *opts, const sgmcmc_launch_t *launch,
                          sgmcmc_stream_t stream);

/* K3 -- fused relativistic SGHMC step, per element. Replaces
 * pysgmcmc/samplers/relativistic_sghmc.py:120-140. grad_cost = d cost / d theta
 * (the kernel negates it, as :100-103 differentiates -cost).
 * R{theta,p,grad} W{theta,p}                                         20 B/param f32 */
int sgmcmc_rsghmc_step_f32(float *theta, float *p, const float *grad_cost, size_t n,
                           float eps, float mass, float c, float D, float b_hat, float grad_decay,
                           const float *xi, uint64_t seed, uint64_t step, const uint64_t *step_dev,
                          void *stats_ws, const sgmcmc_step_opts_t *opts, const sgmcmc_launch_t *launch,
                          sgmcmc_stream_t stream);
int sgmcmc_rsghmc_step_f64(double *theta, double *p, const double *grad_cost, size_t n,
                           double eps, double mass, double c, double D, double b_hat, double grad_decay,
                           const double *xi, uint64_t seed, uint64_t step, const uint64_t *step_dev,
                          void *stats_ws, const sgmcmc_step_opts_t *opts, const sgmcmc_launch_t *launch,
                          sgmcmc_stream_t stream);

/* Many steps of many independent chains on the reference's built-in toy targets in ONE launch (one lane per chain; state
 * in registers; the per-element update operators and the Philox stream of K1-K3, so a chain is the chain next(sampler)
 * gives on the same gradients). Replaces the per-step session.run loop of the reference's sampler tests and of its ESS
 * experiment (pysgmcmc/tests/samplers/sampler_testing.py:14-59, docs/source/experiments/compute_ess.py:176-246; targets
 * pysgmcmc/diagnostics/objective_functions.py:49-98), cost = -log_likelihood with analytic gradients.
 *   sampler: 0 SGHMC, 1 preconditioned SGLD, 2 relativistic SGHMC
 *   target:  0 1-D Gaussian mixture, target_params = {mu[k], var[k], w[k]} (dim 1)
 *            1 banana (dim 2, no parameters)
 *            2 2-D isotropic unit-variance equal-weight mixture, target_params = {x_0, y_0, ..., x_{k-1}, y_{k-1}} (dim 2)
 *   theta, mom (V or p; NULL for SGLD), tau, g, v_hat, minv (NULL for relativistic SGHMC): DEVICE arrays [n_chains][dim]
 *   scalars (host): SGHMC {eps, scale_grad, mdecay}; SGLD {eps, A, scale_grad}; relativistic {eps, mass, c, D, b_hat}
 *   seeds: DEVICE array [n_chains], the chains' Philox keys; element i of a chain draws normal i of quad 0 at (seed, step)
 *   steps first_step .. first_step + n_steps - 1; steps < burn_in_steps adapt (burn_in_steps <= 0: always, quirk Q4)
 *   kept: NULL, or DEVICE array [ceil(n_steps / keep_every)][n_chains][dim]: theta after steps first_step + j keep_every */
int sgmcmc_toy_chains_f32(int sampler, int target, const double *target_params, int k, float *theta, float *mom, float *tau,
                          float *g, float *v_hat, float *minv, size_t n_chains, int dim, const double *scalars,
                          const uint64_t *seeds, uint64_t first_step, uint64_t n_steps, int64_t burn_in_steps,
                          uint64_t keep_every, float *kept, sgmcmc_stream_t stream);
int sgmcmc_toy_chains_f64(int sampler, int target, const double *target_params, int k, double *theta, double *mom, double *tau,
                          double *g, double *v_hat, double *minv, size_t n_chains, int dim, const double *scalars,
                          const uint64_t *seeds, uint64_t first_step, uint64_t n_steps, int64_t burn_in_steps,
                          uint64_t keep_every, double *kept, sgmcmc_stream_t stream);

/* K5 -- write the N(0,1) stream itself: out[i] = xi(seed, step, i). Replaces
 * tf.random_normal in pysgmcmc/samplers/base_classes.py:218-220 for callers that
 * want the draws materialised (tests, relativistic momentum initialisation).      */
int sgmcmc_philox_normal_f32(float *out, size_t n, uint64_t seed, uint64_t step, const uint64_t *step_dev,
                             const sgmcmc_launch_t *launch, sgmcmc_stream_t stream);
int sgmcmc_philox_normal_f64(double *out, size_t n, uint64_t seed, uint64_t step, const uint64_t *step_dev,
                             const sgmcmc_launch_t *launch, sgmcmc_stream_t stream);
/* raw Philox words: out[i] = x[i & 3] of quad i >> 2 (bit-exact integer check)     */
int sgmcmc_philox_bits_u32(uint32_t *out, size_t n, uint64_t seed, uint64_t step, const uint64_t *step_dev,
                             sgmcmc_stream_t stream);

/* *counter += inc on the stream (1-thread kernel; the graph-safe step counter).     */
int sgmcmc_counter_add_u64(uint64_t *counter, uint64_t inc, sgmcmc_stream_t stream);

/* K4 -- Welford running moments of one chain, for cross-chain R-hat. Replaces the
 * per-chain mean/variance pass that pysgmcmc/diagnostics/sampler_diagnostics.py:118-194
 * delegates to pymc3. count = samples folded in INCLUDING this one (>= 1).
 * R{theta,mean,m2} W{mean,m2}                                        20 B/param f32 */
int sgmcmc_moments_update_f32(const float *theta, float *mean, float *m2, size_t n,
                              uint64_t count, const sgmcmc_launch_t *launch, sgmcmc_stream_t stream);
int sgmcmc_moments_update_f64(const double *theta, double *mean, double *m2, size_t n,
                              uint64_t count, const sgmcmc_launch_t *launch, sgmcmc_stream_t stream);

/* R-hat exchange step (replaces what pysgmcmc/diagnostics/sampler_diagnostics.py:118-194
 * delegates to pymc3.diagnostics.gelman_rubin). The library owns no communicator: the host issues the collective
 * (torch.distributed, backend "nccl" = RCCL) between pack and finish.
 *   pack:   every chain writes [mean | mean^2 | m2/(count-1)] of its Welford moments.
 *           n_shards = 1, shard_len = n: out3 = 3n elements, for ONE all-reduce(SUM) across the m chains.
 *           n_shards = m, shard_len = ceil(n/m) (any padding): out3 = m chunks of [3][shard_len]; chunk s holds the
 *           three rows of parameters [s*shard_len, (s+1)*shard_len) (zeros beyond n). ONE reduce-scatter(SUM) then
 *           leaves on rank s the summed rows of ITS parameter shard: 3n(m-1)/m elements cross xGMI per rank instead
 *           of the all-reduce's 6n(m-1)/m, and every rank finishes only n/m parameters.
 *   finish: on [S_mean | S_sq | S_var] with row pitch ld, n valid elements:
 *           rhat[i] = sqrt(((W (cnt-1)/cnt) + B/cnt) / W), W = S_var/m, B = cnt (S_sq - S_mean^2/m)/(m-1);
 *           one IEEE rounding per operation in the dtype.
 * summary_out4 / summary_ws (both NULL or both given; ws = sgmcmc_summary_workspace_bytes()):
 * the K6 summary {sum, sum of squares, min, max} of rhat is left in DEVICE memory on the same
 * stream -- the in-loop exchange never synchronises with the host.                              */
int sgmcmc_rhat_pack_f32(const float *mean, const float *m2, size_t n, uint64_t count, size_t n_shards, size_t shard_len,
                         float *out3, sgmcmc_stream_t stream);
int sgmcmc_rhat_pack_f64(const double *mean, const double *m2, size_t n, uint64_t count, size_t n_shards, size_t shard_len,
                         double *out3, sgmcmc_stream_t stream);
int sgmcmc_rhat_finish_f32(const float *sum3, size_t n, size_t ld, int m_chains, uint64_t count,
                           float *rhat, double *summary_out4, void *summary_ws, sgmcmc_stream_t stream);
int sgmcmc_rhat_finish_f64(const double *sum3, size_t n, size_t ld, int m_chains, uint64_t count,
                           double *rhat, double *summary_out4, void *summary_ws, sgmcmc_stream_t stream);

/* K6 -- deterministic summary of an array: out4 (device, 4 doubles) = {sum, sum of
 * squares, min, max}. Wave-shuffle partial sums -> LDS -> per-block partials in
 * `workspace` -> fixed-order final pass; bit-reproducible for a given n.
 * workspace must hold sgmcmc_summary_workspace_bytes() bytes (device).
 * Used for the adapted-preconditioner (minv) statistics at the end of burn-in and
 * for max/mean R-hat.                                                               */
size_t sgmcmc_summary_workspace_bytes(void);
int sgmcmc_summary_f32(const float *x, size_t n, double *out4, void *workspace, sgmcmc_stream_t stream);
int sgmcmc_summary_f64(const double *x, size_t n, double *out4, void *workspace, sgmcmc_stream_t stream);

/* BNN cost path helpers (the gradient producer of the update path; replaces the TF graph of
 * pysgmcmc/models/bayesian_neural_network.py:365-388 after the network's last GEMM).
 *
 * bnn_head: from the network mean output `mean[B]`, targets `y[B]`, the scalar log-variance
 * parameter `log_var` and sum(theta^2) over all parameters, computes in ONE launch:
 *   delta[B] = d NLL / d mean, cost_out = NLL (likelihood / batch_size + both priors / n_examples),
 *   grad_log_var_out = d NLL / d log_var, mse_out, and (if grad_last_bias_out != NULL) the bias
 *   gradient of a single-output last layer = sum_i delta[i] (+ prior term; `last_bias` = that bias).
 * sum(theta^2) comes from `theta_sumsq` (a device double) or, when `stats_ws` != NULL, from the
 * per-block records the previous step kernel left in its statistics workspace (summed here in a
 * fixed order; saves the K7 launch on the critical path). All scalar outputs are device pointers.
 * fold_prior_grad is a bit mask: bit 0 leaves the weight-prior gradient term (wdecay / (n_params * n_examples)) * theta
 * out of the gradients because the caller passes it to the update kernel as grad_decay; bit 1 says
 * `mean` lacks the single-output layer's bias (a plain GEMV produced it) and *last_bias is added here.
 *
 * tanh_backward: delta[i] *= 1 - h[i]^2.
 * tanh_backward_colsum: the same on a row-major [rows][cols] matrix, plus colsum[c] = sum_r delta[r][c]
 * (+ beta * bias[c] if beta != 0): the bias gradient of that layer, deterministic (no atomics).
 * bnn_last_layer_backward: backward of a single-output last layer fused with the tanh backward below it:
 * delta_prev[r][c] = dvec[r] w[c] (1 - h[r][c]^2), colsum[c] = sum_r delta_prev[r][c] (+ beta bias_prev[c]),
 * gw[c] = sum_r h[r][c] dvec[r] (+ beta w[c]) -- replaces an outer product, a GEMV and the colsum kernel.      */
int sgmcmc_bnn_head_f32(const float *mean, const float *y, const float *log_var, const double *theta_sumsq,
                        const void *stats_ws, const float *last_bias, size_t B,
                        double batch_size, double n_examples, double n_params, double wdecay, double prior_mean,
                        double prior_var, int fold_prior_grad, float *delta, float *cost_out, float *grad_log_var_out,
                        float *grad_last_bias_out, float *mse_out, sgmcmc_stream_t stream);
int sgmcmc_bnn_head_f64(const double *mean, const double *y, const double *log_var, const double *theta_sumsq,
                        const void *stats_ws, const double *last_bias, size_t B,
                        double batch_size, double n_examples, double n_params, double wdecay, double prior_mean,
                        double prior_var, int fold_prior_grad, double *delta, double *cost_out, double *grad_log_var_out,
                        double *grad_last_bias_out, double *mse_out, sgmcmc_stream_t stream);
int sgmcmc_tanh_backward_f32(float *delta, const float *h, size_t n, sgmcmc_stream_t stream);
int sgmcmc_tanh_backward_f64(double *delta, const double *h, size_t n, sgmcmc_stream_t stream);
int sgmcmc_bnn_last_layer_backward_f32(const float *dvec, const float *w, const float *h, size_t rows, size_t cols,
                                       const float *bias_prev, float beta, float *delta_prev, float *colsum, float *gw,
                                       sgmcmc_stream_t stream);
int sgmcmc_bnn_last_layer_backward_f64(const double *dvec, const double *w, const double *h, size_t rows, size_t cols,
                                       const double *bias_prev, double beta, double *delta_prev, double *colsum,
                                       double *gw, sgmcmc_stream_t stream);
int sgmcmc_tanh_backward_colsum_f32(float *delta, const float *h, size_t rows, size_t cols, const float *bias, float beta,
                                    float *colsum, sgmcmc_stream_t stream);
int sgmcmc_tanh_backward_colsum_f64(double *delta, const double *h, size_t rows, size_t cols, const double *bias,
                                    double beta, double *colsum, sgmcmc_stream_t stream);

/* Fused small-model path: `n_steps` COMPLETE SGHMC steps of a tanh-MLP BNN with one output unit in ONE
 * launch, one 1024-lane workgroup per chain (the launch-bound regime of the reference's default 3x50
 * net). Per step and chain: minibatch window [start, start+batch) of the resident data set, forward,
 * loss head (pysgmcmc/models/bayesian_neural_network.py:365-388), analytic backward into `grad`, the K1
 * update (pysgmcmc/samplers/sghmc.py:165-251; same operator and Philox stream as
 * sgmcmc_sghmc_step: noise of element i at step s is xi(seed_base + chain, s, i); the weight-prior
 * gradient rides in as grad_decay), sum(theta^2) for the next cost.
 *   rows theta..minv: n_params elements per chain, chain c at + c * chain_stride, 16-B aligned;
 *     parameter order W1, b1, ..., WL, bL, log_var (W row-major in x out).
 *   layer_sizes: HOST array [inputs, h1, ..., 1] of n_layers + 1 ints (n_layers <= 8).
 *   X [n_data][inputs], y [n_data]: device; window_starts: device [n_chains][n_steps].
 *   step index = first_step + t; adapt while step < burn_in_steps (always if burn_in_steps == 0).
 *   xi: NULL or injected noise [n_steps][n_params] for chain 0. cost_out: device [n_chains][n_steps],
 *     cost_out[c][t] = NLL at the parameters BEFORE step t.
 * Activations and a copy of the parameters live in LDS: (2 * batch * sum(layer sizes) + n_params)
 * elements must fit 160 KiB (else SGMCMC_EINVAL: use the GEMM path).                               */
int sgmcmc_bnn_fused_sghmc_steps_f32(float *theta, float *V, float *grad, float *tau, float *g, float *v_hat, float *minv,
                                     size_t n_params, size_t chain_stride, int n_chains, const int *layer_sizes,
                                     int n_layers, const float *X, const float *y, size_t n_data,
                                     const int *window_starts, int batch, double batch_size, double n_examples,
                                     double wdecay, double prior_mean, double prior_var, float eps, float scale_grad,
                                     float mdecay, uint64_t first_step, uint64_t n_steps, uint64_t burn_in_steps,
                                     uint64_t seed_base, const float *xi, float *cost_out, sgmcmc_stream_t stream);
int sgmcmc_bnn_fused_sghmc_steps_f64(double *theta, double *V, double *grad, double *tau, double *g, double *v_hat,
                                     double *minv, size_t n_params, size_t chain_stride, int n_chains,
                                     const int *layer_sizes, int n_layers, const double *X, const double *y,
                                     size_t n_data, const int *window_starts, int batch, double batch_size,
                                     double n_examples, double wdecay, double prior_mean, double prior_var, double eps,
                                     double scale_grad, double mdecay, uint64_t first_step, uint64_t n_steps,
                                     uint64_t burn_in_steps, uint64_t seed_base, const double *xi, double *cost_out,
                                     sgmcmc_stream_t stream);

/* The same with the preconditioned-SGLD update (K2, pysgmcmc/samplers/sgld.py:149-211) instead of SGHMC: no
 * momentum row; `A` as in sgmcmc_sgld_step_*. The reference's BNN accepts both samplers
 * (pysgmcmc/sampling.py:40,64).                                                                    */
int sgmcmc_bnn_fused_sgld_steps_f32(float *theta, float *grad, float *tau, float *g, float *v_hat, float *minv,
                                    size_t n_params, size_t chain_stride, int n_chains, const int *layer_sizes,
                                    int n_layers, const float *X, const float *y, size_t n_data,
                                    const int *window_starts, int batch, double batch_size, double n_examples,
                                    double wdecay, double prior_mean, double prior_var, float eps, float scale_grad,
                                    float A, uint64_t first_step, uint64_t n_steps, uint64_t burn_in_steps,
                                    uint64_t seed_base, const float *xi, float *cost_out, sgmcmc_stream_t stream);
int sgmcmc_bnn_fused_sgld_steps_f64(double *theta, double *grad, double *tau, double *g, double *v_hat, double *minv,
                                    size_t n_params, size_t chain_stride, int n_chains, const int *layer_sizes,
                                    int n_layers, const double *X, const double *y, size_t n_data,
                                    const int *window_starts, int batch, double batch_size, double n_examples,
                                    double wdecay, double prior_mean, double prior_var, double eps, double scale_grad,
                                    double A, uint64_t first_step, uint64_t n_steps, uint64_t burn_in_steps,
                                    uint64_t seed_base, const double *xi, double *cost_out, sgmcmc_stream_t stream);

/* Forward of the last hidden layer fused with a single-output layer above it (models/bayesian_neural_network.py:
 * 48-56): a[rows][cols] = tanh(a + bias[c]) in place (bias NULL = none), out[r] = sum_c a[r][c] * w[c] (without the output
 * unit's bias: the loss head adds it). The forward product before it is then a plain GEMM -- at batch 256 the library's plain
 * product is 1.4-2.1 us faster than its bias-epilogue one.
 * stats_ws / tsq_parts (both NULL or both given): the first min(16, rows) workgroups also add up one slice each of
 * the sum(theta^2) partials of the previous step kernel's statistics workspace into tsq_parts[0..15] (doubles), for
 * sgmcmc_bnn_head_last_layer_backward_*. (ABI v6 dropped sgmcmc_tanh_rowdot_*, the bias == NULL form as an entry point of its
 * own: no plan of the cost path launched it any more.)                                                              */
int sgmcmc_bias_tanh_rowdot_f32(float *a, const float *bias, const float *w, size_t rows, size_t cols, float *out,
                                const void *stats_ws, double *tsq_parts, sgmcmc_stream_t stream);
int sgmcmc_bias_tanh_rowdot_f64(double *a, const double *bias, const double *w, size_t rows, size_t cols, double *out,
                                const void *stats_ws, double *tsq_parts, sgmcmc_stream_t stream);
/* Hidden-layer activation (bayesian_neural_network.py:28-56, fully_connected with tanh): a[rows][cols] =
 * tanh(a + bias[c]) in place.                                                                                    */
int sgmcmc_bias_tanh_f32(float *a, const float *bias, size_t rows, size_t cols, sgmcmc_stream_t stream);
int sgmcmc_bias_tanh_f64(double *a, const double *bias, size_t rows, size_t cols, sgmcmc_stream_t stream);

/* A hidden layer of the BNN forward pass in ONE launch (the Dense(tanh) layers of
 * pysgmcmc/models/bayesian_neural_network.py:30-52): fp32 matrix-core product with the activation as its epilogue,
 *   out[m][n] = tanh( sum_k h[m][k] W[k][n] + bias[n] )
 * replacing library GEMM + sgmcmc_bias_tanh_f32 (two launches). h [M = batch][ldh >= K], W [K][ldw >= N] (the layer's
 * kernel as it lies in the arena), out [M][ldo >= N], all row-major.
 *   w_next, dot_parts: both NULL, or the single output unit's weights [N] and a [parts][M] buffer, parts =
 *     sgmcmc_bnn_dense_tanh_dot_parts(M, N): the launch also leaves dot_parts[t][m] = sum over the columns n of column tile t
 *     of out[m][n] * w_next[n] (fixed order) -- the Dense(1) layer of :53-56; sgmcmc_bnn_head_last_layer_backward_* adds the
 *     partials per row in order (n_mean_parts = parts).
 *   stats_ws, tsq_parts: both NULL, or as in sgmcmc_tanh_rowdot_*: workgroups 0 .. 15 add up one slice each of the
 *     sum(theta^2) records of the previous step kernel (needs >= 16 full tiles; SGMCMC_EINVAL otherwise).
 * M % 32 == 0, N % 64 == 0, K % 16 == 0, K >= 64; every row 16-byte aligned (ld* % 4 == 0); operands < 2 GiB each.
 * One workgroup (8 waves, one per CU: the operand ring fills LDS) per 32 x 64 output tile. More tiles than compute units run in
 * rounds; when the last round would be less than half full (256 x 4864 outputs: 608 tiles on 256 CUs) the columns it covers are
 * computed as 32 x 32 HALF tiles by a second launch on the same stream -- twice the workgroups at half the work each, 2.5 rounds
 * instead of 3. Which columns that is depends only on (M, N, compute units of the current device): results are deterministic.
 * fp32 MFMA is an exact k-ordered fmaf chain: results differ from a library GEMM in summation order only.                  */
int sgmcmc_bnn_dense_tanh_dot_parts(int M, int N);   /* rows of dot_parts for an M x N layer on the current device (0: invalid shape) */
int sgmcmc_bnn_dense_tanh_f32(const float *h, const float *W, const float *bias, float *out, int M, int N, int K, int ldh,
                              int ldw, int ldo, const float *w_next, float *dot_parts, const void *stats_ws, double *tsq_parts,
                              sgmcmc_stream_t stream);

/* Backward step through a hidden tanh layer of the same network in ONE launch (the reference's graph differentiates
 * bayesian_neural_network.py:30-52 through tf.gradients, samplers/base_classes.py:121-122): replaces library GEMM
 * delta W^T + sgmcmc_tanh_backward_colsum_f32.
 *   out[m][n] = ( sum_k delta[m][k] W[n][k] ) * (1 - act[m][n]^2)     delta [M][K] = d cost / d pre-activation of the layer above,
 *                                                                      W [N][K] that layer's weights, act [M][N] this layer's tanh outputs
 *   colsum_parts[t][n] = sum of out[m][n] over row tile t (rows 32 t .. 32 t + 31)      (optional; [M / 32][N])
 * i.e. out = d cost / d pre-activation of this layer; its column sums are this layer's bias gradient. Same pipeline and shape
 * limits and the same tiling (half tiles for a thin last round) as sgmcmc_bnn_dense_tanh_f32 (M % 32, N % 64, K % 16, K >= 64,
 * 16-byte aligned rows, operands < 2 GiB); W is read along its rows, no transpose is formed. A workgroup owns 32 rows, so the sums over ALL rows need a second pass over what other
 * workgroups wrote: inside one launch that costs more than the launch it saves (measured), so the row-tile
 * sums are left in colsum_parts and added up -- in row-tile order: bit-reproducible, no atomics --
 *   * by the NEXT launch of this function on the stream, on the side (fin_*: fin_colsum[c] = sum_r fin_parts[r][c]
 *     (+ fin_beta * fin_bias[c]), r < fin_rows, c < fin_n; fin_parts must not be the colsum_parts the same launch writes), or
 *   * by sgmcmc_colsum_finish_f32 (a small launch), or
 *   * not at all (colsum_parts NULL) when the caller gets the bias gradient elsewhere -- BNNCost takes the first layer's from
 *     the weight-gradient product itself, [x | 1]^T delta.                                                                   */
int sgmcmc_bnn_dense_tanh_backward_f32(const float *delta, const float *W, const float *act, float *out, float *colsum_parts,
                                       int M, int N, int K, int ldd, int ldw, int lda, int ldo, const float *fin_parts,
                                       int fin_rows, int fin_n, const float *fin_bias, float fin_beta, float *fin_colsum,
                                       sgmcmc_stream_t stream);
int sgmcmc_colsum_finish_f32(const float *parts, int rows, int n, const float *bias, float beta, float *colsum,
                             sgmcmc_stream_t stream);

/* sgmcmc_bnn_head_* and sgmcmc_bnn_last_layer_backward_* in ONE launch (every dependent launch of the step costs
 * ~5 us): `mean` [n_mean_parts][rows] is the single-output layer's pre-bias output -- n_mean_parts = 1: the vector
 * sgmcmc_tanh_rowdot_* writes; > 1: the per-column-tile partial dot products sgmcmc_bnn_dense_tanh_f32 writes, added here
 * in order (rows <= 1024) --, tsq_parts the slices of sum(theta^2) that launch left; d cost/d mean is formed on the fly by
 * every workgroup, one extra workgroup writes the head's scalar outputs. Arguments as in the two separate entry points
 * (fold_prior_grad: the same bit mask).                                                                            */
int sgmcmc_bnn_head_last_layer_backward_f32(
    const float *mean, size_t n_mean_parts, const float *y, const float *log_var, const double *tsq_parts, const float *last_bias, size_t rows,
    size_t cols, double batch_size, double n_examples, double n_params, double wdecay, double prior_mean, double prior_var,
    int fold_prior_grad, const float *w, const float *h, const float *bias_prev, float beta, float *cost_out,
    float *grad_log_var_out, float *grad_last_bias_out, float *mse_out, float *delta_prev, float *colsum, float *gw,
    sgmcmc_stream_t stream);
int sgmcmc_bnn_head_last_layer_backward_f64(
    const double *mean, size_t n_mean_parts, const double *y, const double *log_var, const double *tsq_parts, const double *last_bias, size_t rows,
    size_t cols, double batch_size, double n_examples, double n_params, double wdecay, double prior_mean, double prior_var,
    int fold_prior_grad, const double *w, const double *h, const double *bias_prev, double beta, double *cost_out,
    double *grad_log_var_out, double *grad_last_bias_out, double *mse_out, double *delta_prev, double *colsum, double *gw,
    sgmcmc_stream_t stream);

/* Minibatch window [start, start + batch) of the device-resident dataset copied into the (static) feed buffers with
 * ONE launch: x_out[batch][dim] = X[start ..][:], y_out[batch] = y[start ..] (pysgmcmc/data_batches.py:118-123).
 * x_out_ld >= dim is the row pitch of x_out in elements (dim: dense); columns beyond dim are left alone -- BNNCost keeps a
 * column of ones there, so that the first layer's weight-gradient product [x | 1]^T delta also yields its bias gradient. */
int sgmcmc_window_gather_f32(const float *X, const float *y, size_t n_data, size_t start, size_t batch, size_t dim,
                             float *x_out, size_t x_out_ld, float *y_out, sgmcmc_stream_t stream);
int sgmcmc_window_gather_f64(const double *X, const double *y, size_t n_data, size_t start, size_t batch, size_t dim,
                             double *x_out, size_t x_out_ld, double *y_out, sgmcmc_stream_t stream);

/* Weight gradients of wide dense layers, gW = h^T delta (tf.gradients of pysgmcmc/models/bayesian_neural_network.py:30-56), at fp32
 * accuracy on the bf16 matrix pipe (csrc/sgmcmc_bnn_gw.hip). Both operands are activations [M = batch][features]; each is first
 * written as three exact bf16 planes x = x0 + x1 + x2 (8 + 8 + 8 significant bits, truncation splits, no rounding) in the layout
 *   plane p of X [M][N]:  P[p][m / 8][n][m % 8]  (bf16),  planes of a set M * N * 2 bytes apart,
 * then `count` products C_z [nA][nB] (pitch ldc, overwritten) = sum_m A_z[m][:]^T B_z[m][:] run as ONE launch of six
 * v_mfma_f32_32x32x16_bf16 per 16 batch rows (the partial products of order <= 2^-16) with fp32 accumulation in a fixed order:
 * bit-reproducible, error against fp64 no larger than the library's fp32 product on the same operands (tests/test_bnn_dense_gpu.py).
 *   sgmcmc_bnn_planes_bytes(M, N): bytes of one plane set (0: invalid shape; M % 16 == 0 required).
 *   split_planes: X_z = X + z * x_stride (elements) [M][ldx >= N] -> planes + z * planes_stride (bytes), z < count.
 *   gw_planes: plane sets a_planes + z * a_stride, b_planes + z * b_stride (bytes); C + z * c_stride (elements).
 * Faster than the library's fp32 product where the gradient has many 128 x 128 tiles (two 4864 x 4864 products at batch 256:
 * 131 against 188 us); at 2048 x 2048 it is not once the planes have to be written (profiles/r06_gw_gate.txt).          */
size_t sgmcmc_bnn_planes_bytes(int M, int N);
int sgmcmc_bnn_split_planes_f32(const float *X, int count, size_t x_stride, int M, int N, int ldx, void *planes, size_t planes_stride,
                                sgmcmc_stream_t stream);
int sgmcmc_bnn_gw_planes_f32(const void *a_planes, size_t a_stride, const void *b_planes, size_t b_stride, float *C, size_t c_stride,
                             int count, int M, int nA, int nB, int ldc, sgmcmc_stream_t stream);

/* ---- Stein variational gradient descent: pysgmcmc/samplers/svgd.py:118-181 -----------------------
 * The n particles are the rows of a [n_particles x ld] device matrix (row pitch ld >= dim elements; with
 * ld a multiple of 4 and 16-byte aligned bases every access is 16 bytes wide -- the sampler pads ld to 64);
 * grad[i] = d cost / d particle i (svgd.py:118), hist_grad = `historical_grad` (svgd.py:107-110).
 * sgmcmc_svgd_step_* does one step in place:
 *   D = squareform(pdist(X)) ** 2 (tensor_utils.py:397-408, 466-565; svgd.py:165-166),
 *   h = sqrt(0.5 median(D) / log(n + 1)) (tensor_utils.py:197-209; svgd.py:169-171),
 *   K = exp(-D / h^2 / 2), kgrad = (-K X + X * rowsum(K)) / h^2 (svgd.py:173-181),
 *   grad_theta = (K grad + repulsion_sign * kgrad) / n (svgd.py:124-127),
 *   hist = alpha hist + (1 - alpha) grad_theta^2; X -= eps grad_theta / (fudge + sqrt(hist)) (:129-143).
 * repulsion_sign = +1 is the reference as written (its kernel-gradient term ATTRACTS the particles:
 * it is added to a COST gradient and the sum is subtracted); -1 is Liu & Wang's repulsive update.
 * workspace: sgmcmc_svgd_workspace_bytes(n_particles, sizeof element) bytes of device memory, 64-B
 * aligned, owned by the caller; it holds D, K, the bandwidth and the partial sums between launches.
 * n_particles <= sgmcmc_svgd_max_particles() (128), else SGMCMC_EINVAL.
 * sgmcmc_svgd_kernel_* is `SVGDSampler.svgd_kernel(particles)` (svgd.py:149-181): kernel_out
 * [n x n], kernel_grad_out [n x kernel_grad_ld] (already divided by h^2) and bandwidth_out
 * {median, h, h^2} are device pointers, each may be NULL.                                          */
size_t sgmcmc_svgd_workspace_bytes(size_t n_particles, size_t elem_bytes);
int sgmcmc_svgd_max_particles(void);
int sgmcmc_svgd_step_f32(float *particles, const float *grad, float *hist_grad, size_t n_particles, size_t dim, size_t ld,
                         float eps, double alpha, float fudge_factor, int repulsion_sign, void *workspace,
                         sgmcmc_stream_t stream);
int sgmcmc_svgd_step_f64(double *particles, const double *grad, double *hist_grad, size_t n_particles, size_t dim,
                         size_t ld, double eps, double alpha, double fudge_factor, int repulsion_sign, void *workspace,
                         sgmcmc_stream_t stream);
int sgmcmc_svgd_kernel_f32(const float *particles, size_t n_particles, size_t dim, size_t ld, void *workspace,
                           float *kernel_out, float *kernel_grad_out, size_t kernel_grad_ld, float *bandwidth_out,
                           sgmcmc_stream_t stream);
int sgmcmc_svgd_kernel_f64(const double *particles, size_t n_particles, size_t dim, size_t ld, void *workspace,
                           double *kernel_out, double *kernel_grad_out, size_t kernel_grad_ld, double *bandwidth_out,
                           sgmcmc_stream_t stream);

/* ---- Deviations from the signatures proposed in SURVEY.md section 8(b), and why -------------------
 *  1. eps_scaled is not an argument: K1 takes (eps, scale_grad) and derives eps/sqrt(scale_grad) and
 *     the noise-scale constants itself, in the dtype and in the reference's op order
 *     (sghmc.py:115,211-217) -- a caller computing them in double would break bit-parity.
 *  2. philox_offset became (step, step_dev): the Philox counter is (step, quad), so the caller passes
 *     n_iterations instead of step * ceil(n/4); step_dev (device counter) exists for hipGraph replay.
 *  3. grad_decay (all three steps): weight-prior gradient term formed in registers; 0 = reference.
 *  4. stats_ws: the fused "LDS-staged reduction" of north_star rides on the step call itself.
 *  5. launch: per-call geometry instead of a process-wide setter (section 8(b): no global state).
 *  6. sgmcmc_moments_update_*: as proposed (+ launch).
 *  7. sgmcmc_rhat_allreduce(ncclComm_t) is NOT exported: the library would have to own an RCCL
 *     communicator and link librccl; instead the exchange is split into sgmcmc_rhat_pack_* /
 *     sgmcmc_rhat_finish_* around ONE collective the host issues through torch.distributed
 *     (backend "nccl" = RCCL), which already owns the communicator: an all-reduce, or -- the sharded
 *     layout -- a reduce-scatter (+ an all-gather of R-hat only if the full vector is wanted), the
 *     "reduce-scatter + all-gather" of SURVEY 8(e). No torch/RCCL types in the ABI.
 *  8. sgmcmc_cpu_* is NOT exported by the product library: the CPU restatement is test
 *     infrastructure (oracle/libsgmcmc_oracle.so: oracle_sghmc_step_f32, ...), never shipped as a
 *     fallback -- the product path fails loudly without a GPU.
 *  9. (ABI v4) the hand-written weight-gradient GEMM entry points of v3 (with their tile-shape and timing-experiment
 *     arguments) did not beat the library products and are no longer exported (round 5 removed the separate
 *     experiments library too; the measurements are kept under profiles/, see profiles/HISTORY.md).
 * 10. f64 variants of everything (the reference's default dtype is float64, base_classes.py:25).
 * 11. (ABI v6) the contract map at the top of this file says which entry points ARE the section 8(b) boundary and which are the
 *     cost path's internals; sgmcmc_tanh_rowdot_* (superseded by sgmcmc_bias_tanh_rowdot_*, launched by no plan) is gone.
 */

#ifdef __cplusplus
}
#endif
#endif /* SGMCMC_HIP_H */
