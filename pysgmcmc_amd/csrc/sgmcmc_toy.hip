// sgmcmc_toy.hip -- many steps of many independent chains on the reference's built-in toy targets in ONE launch.
//
// pysgmcmc/diagnostics/objective_functions.py:49-98 holds the targets every sampler test and experiment of the reference
// uses (banana, 1-D Gaussian mixtures); its ESS experiment (docs/source/experiments/compute_ess.py:176-246) runs ONE chain
// of 2e6 steps per stepsize, one `session.run` per step. Here a lane owns a chain: it keeps the chain's 1-2 parameters and
// their sampler state in registers, evaluates the target's analytic cost gradient, applies the SAME per-element update
// operators as the streaming kernels K1-K3 (SghmcOp / SgldOp / RsghmcOp of sgmcmc_device.hpp, same Philox stream: seed
// of the chain, counter (step, quad 0) -- so a chain is the chain `next(sampler)` would produce given the same
// gradients), and writes every keep_every-th state to the trace. 2e6 steps take a fraction of a second; 64-1024 chains
// cost the same. This is the n-steps-per-launch path for LDS/register-sized problems (cf. sgmcmc_bnn_fused.hip for the
// small BNN); it is latency-bound by design (one dependent chain per lane).
//
// Gradients: cost = -log_likelihood (compute_ess.py:170-174, tests/samplers/sampler_testing.py:14-18), derivatives
// written out in the op order of the CPU restatement under oracle/ (gmm_cost_grad, banana), which tests/ compares it with.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "sgmcmc_hip.h"

#pragma clang fp contract(off)

#include "sgmcmc_device.hpp"
#include "sgmcmc_host.hpp"

using namespace sgmcmc_host;

namespace {

constexpr int TOY_MAX_K = 16;

template <typename T>
struct ToyArgs {
    int target, k, dim;
    // target 0 (1-D mixture): a[i] = log w_i, b[i] = 0.5 log(2 pi var_i), mu[i], var[i]   (a, b rounded to T on the host)
    // target 2 (2-D isotropic unit-variance equal-weight mixture): mu[i], var[i] hold the centre (x_i, y_i)
    T a[TOY_MAX_K], b[TOY_MAX_K], mu[TOY_MAX_K], var[TOY_MAX_K];
    T *theta, *mom, *tau, *g, *vh, *minv;     // [n_chains][dim]
    const uint64_t *seeds;                     // [n_chains] Philox keys
    size_t n_chains;
    T s[5];                                    // the sampler's derived scalars (same blocks as sgmcmc_*_scalars_*)
    uint64_t first_step, n_steps, burn_in_steps, keep_every;
    T *kept;                                   // nullable: [n_kept][n_chains][dim]
    int perpetual_adapt;                       // burn_in_steps <= 0: adaptation never stops (reference quirk Q4)
};

// d cost / d x of the 1-D mixture: cost = -logsumexp_i[ a_i - b_i - 0.5 (x - mu_i)^2 / var_i ]
template <typename T>
__device__ __forceinline__ T gmm1d_cost_grad(const ToyArgs<T> &A, T x)
{
    T t[TOY_MAX_K], mx = -(T)INFINITY, s = T(0), g = T(0);
    for (int i = 0; i < A.k; ++i) {
        T d = x - A.mu[i];
        t[i] = (A.a[i] - A.b[i]) - (T(0.5) * (d * d)) / A.var[i];
        if (t[i] > mx) mx = t[i];
    }
    for (int i = 0; i < A.k; ++i) { t[i] = (T)exp((double)(t[i] - mx)); s += t[i]; }
    for (int i = 0; i < A.k; ++i) g += (t[i] / s) * ((x - A.mu[i]) / A.var[i]);
    return g;
}
// 2-D mixture, unit variances, equal weights: cost = -logsumexp_i[ -0.5 |x - c_i|^2 ] + const
template <typename T>
__device__ __forceinline__ void gmm2d_cost_grad(const ToyArgs<T> &A, T x, T y, T &gx, T &gy)
{
    T t[TOY_MAX_K], mx = -(T)INFINITY, s = T(0);
    for (int i = 0; i < A.k; ++i) {
        T dx = x - A.mu[i], dy = y - A.var[i];
        t[i] = -(T(0.5) * (dx * dx + dy * dy));
        if (t[i] > mx) mx = t[i];
    }
    for (int i = 0; i < A.k; ++i) { t[i] = (T)exp((double)(t[i] - mx)); s += t[i]; }
    gx = T(0); gy = T(0);
    for (int i = 0; i < A.k; ++i) { T w = t[i] / s; gx += w * (x - A.mu[i]); gy += w * (y - A.var[i]); }
}

// SAMPLER 0: SGHMC (K1), 1: preconditioned SGLD (K2), 2: relativistic SGHMC (K3)
template <typename T, int SAMPLER>
__global__ void __launch_bounds__(64) toy_chains_kernel(const ToyArgs<T> A)
{
    const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= A.n_chains) return;
    const int dim = A.dim;
    T th[4], mo[4], tau[4], g[4], vh[4], mi[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { th[j] = T(1); mo[j] = T(1); tau[j] = T(1); g[j] = T(1); vh[j] = T(1); mi[j] = T(1); }
    for (int j = 0; j < dim; ++j) {
        th[j] = A.theta[c * dim + j];
        if (SAMPLER != 1) mo[j] = A.mom[c * dim + j];
        if (SAMPLER != 2) { tau[j] = A.tau[c * dim + j]; g[j] = A.g[c * dim + j]; vh[j] = A.vh[c * dim + j]; mi[j] = A.minv[c * dim + j]; }
    }
    NoiseKey nk;
    const uint64_t seed = A.seeds[c];
    nk.k0 = (uint32_t)seed; nk.k1 = (uint32_t)(seed >> 32);
    nk.step_dev = nullptr;
    uint64_t n_kept = 0;
    for (uint64_t s = 0; s < A.n_steps; ++s) {
        const uint64_t step = A.first_step + s;
        nk.s0 = (uint32_t)step; nk.s1 = (uint32_t)(step >> 32);
        T gr[4] = {T(0), T(0), T(0), T(0)};
        if (A.target == 0) {
            gr[0] = gmm1d_cost_grad<T>(A, th[0]);
        } else if (A.target == 1) {
            // banana: cost = 0.5 (0.01 x^2 + (y + 0.1 x^2 - 10)^2), objective_functions.py:49-59
            T x = th[0], y = th[1];
            T u = (y + T(0.1) * (x * x)) - T(10);
            gr[0] = T(0.01) * x + u * (T(0.2) * x);
            gr[1] = u;
        } else {
            gmm2d_cost_grad<T>(A, th[0], th[1], gr[0], gr[1]);
        }
        const bool adapt = A.perpetual_adapt || step < A.burn_in_steps;
        if constexpr (SAMPLER == 0) {
            if (adapt) {
                SghmcOp<T, true, false> op{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                           A.s[0], A.s[1], A.s[2], A.s[3], A.s[4], T(0), nk, nullptr};
                typename SghmcOp<T, true, false>::Regs R;
#pragma unroll
                for (int j = 0; j < 4; ++j) { R.th[j] = th[j]; R.v[j] = mo[j]; R.gr[j] = gr[j]; R.tau[j] = tau[j]; R.g[j] = g[j]; R.vh[j] = vh[j]; }
                op.compute(0, R);
#pragma unroll
                for (int j = 0; j < 4; ++j) { th[j] = R.th[j]; mo[j] = R.v[j]; tau[j] = R.tau[j]; g[j] = R.g[j]; vh[j] = R.vh[j]; mi[j] = R.mi[j]; }
            } else {
                SghmcOp<T, false, false> op{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                            A.s[0], A.s[1], A.s[2], A.s[3], A.s[4], T(0), nk, nullptr};
                typename SghmcOp<T, false, false>::Regs R;
#pragma unroll
                for (int j = 0; j < 4; ++j) { R.th[j] = th[j]; R.v[j] = mo[j]; R.gr[j] = gr[j]; R.mi[j] = mi[j]; }
                op.compute(0, R);
#pragma unroll
                for (int j = 0; j < 4; ++j) { th[j] = R.th[j]; mo[j] = R.v[j]; }
            }
        } else if constexpr (SAMPLER == 1) {
            if (adapt) {
                SgldOp<T, true, false> op{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                          A.s[0], A.s[1], A.s[2], A.s[3], A.s[4], T(0), nk, nullptr};
                typename SgldOp<T, true, false>::Regs R;
#pragma unroll
                for (int j = 0; j < 4; ++j) { R.th[j] = th[j]; R.gr[j] = gr[j]; R.tau[j] = tau[j]; R.g[j] = g[j]; R.vh[j] = vh[j]; }
                op.compute(0, R);
#pragma unroll
                for (int j = 0; j < 4; ++j) { th[j] = R.th[j]; tau[j] = R.tau[j]; g[j] = R.g[j]; vh[j] = R.vh[j]; mi[j] = R.mi[j]; }
            } else {
                SgldOp<T, false, false> op{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                           A.s[0], A.s[1], A.s[2], A.s[3], A.s[4], T(0), nk, nullptr};
                typename SgldOp<T, false, false>::Regs R;
#pragma unroll
                for (int j = 0; j < 4; ++j) { R.th[j] = th[j]; R.gr[j] = gr[j]; R.mi[j] = mi[j]; }
                op.compute(0, R);
#pragma unroll
                for (int j = 0; j < 4; ++j) th[j] = R.th[j];
            }
        } else {
            RsghmcOp<T, false, false> op{nullptr, nullptr, nullptr, nullptr, A.s[0], A.s[1], A.s[2], A.s[3], A.s[4], T(0), nk, nullptr};
            typename RsghmcOp<T, false, false>::Regs R;
#pragma unroll
            for (int j = 0; j < 4; ++j) { R.th[j] = th[j]; R.p[j] = mo[j]; R.gr[j] = gr[j]; }
            op.compute(0, R);
#pragma unroll
            for (int j = 0; j < 4; ++j) { th[j] = R.th[j]; mo[j] = R.p[j]; }
        }
        if (A.kept != nullptr && s % A.keep_every == 0) {
            // what itertools.islice(sampler, 0, n, keep_every) yields (compute_ess.py:176-182); [kept][chain][dim]:
            // the lanes of a wave write neighbouring chains
            for (int j = 0; j < dim; ++j) A.kept[(n_kept * A.n_chains + c) * dim + j] = th[j];
            ++n_kept;
        }
    }
    for (int j = 0; j < dim; ++j) {
        A.theta[c * dim + j] = th[j];
        if (SAMPLER != 1) A.mom[c * dim + j] = mo[j];
        if (SAMPLER != 2) { A.tau[c * dim + j] = tau[j]; A.g[c * dim + j] = g[j]; A.vh[c * dim + j] = vh[j]; A.minv[c * dim + j] = mi[j]; }
    }
}

template <typename T>
int toy_chains(int sampler, int target, const double *tp, int k, T *theta, T *mom, T *tau, T *g, T *vh, T *minv,
               size_t n_chains, int dim, const double *sc, const uint64_t *seeds, uint64_t first_step, uint64_t n_steps,
               int64_t burn_in_steps, uint64_t keep_every, T *kept, hipStream_t st)
{
    if (n_chains == 0 || n_steps == 0) return 0;
    if (sampler < 0 || sampler > 2) return fail(SGMCMC_EINVAL, "toy_chains: sampler must be 0 (SGHMC), 1 (SGLD) or 2 (relativistic SGHMC)");
    if (!theta || !seeds || !sc || (sampler != 1 && !mom) || (sampler != 2 && (!tau || !g || !vh || !minv)))
        return fail(SGMCMC_EINVAL, "toy_chains: NULL state array");
    if (keep_every == 0) return fail(SGMCMC_EINVAL, "toy_chains: keep_every must be >= 1");
    const int want_dim = target == 0 ? 1 : 2;
    if (target < 0 || target > 2 || dim != want_dim) return fail(SGMCMC_EINVAL, "toy_chains: target 0 (1-D mixture) has dim 1, 1 (banana) and 2 (2-D mixture) dim 2");
    if (target != 1 && (!tp || k < 1 || k > TOY_MAX_K)) return fail(SGMCMC_EINVAL, "toy_chains: mixtures take 1..16 components");
    ToyArgs<T> A;
    A.target = target; A.k = k; A.dim = dim;
    for (int i = 0; i < TOY_MAX_K; ++i) { A.a[i] = A.b[i] = A.mu[i] = T(0); A.var[i] = T(1); }
    if (target == 0) {
        // tp = {mu[k], var[k], w[k]}; the constant terms rounded to T exactly as the oracle forms them
        for (int i = 0; i < k; ++i) {
            A.mu[i] = (T)tp[i]; A.var[i] = (T)tp[k + i];
            A.a[i] = (T)std::log((double)(T)tp[2 * k + i]);
            A.b[i] = T(0.5) * (T)std::log(2.0 * 3.14159265358979323846 * (double)(T)tp[k + i]);
        }
    } else if (target == 2) {
        for (int i = 0; i < k; ++i) { A.mu[i] = (T)tp[2 * i]; A.var[i] = (T)tp[2 * i + 1]; }     // centres (x_i, y_i)
    }
    A.theta = theta; A.mom = mom; A.tau = tau; A.g = g; A.vh = vh; A.minv = minv;
    A.seeds = seeds; A.n_chains = n_chains;
    // derived scalars, in the dtype, same op order as the step calls (sgmcmc_sghmc.hip / sgmcmc_sgld.hip / sgmcmc_rsghmc.hip)
    if (sampler == 0) {
        T eps = (T)sc[0], scale_grad = (T)sc[1], mdecay = (T)sc[2];
        T eps_s = eps / std::sqrt(scale_grad);
        A.s[0] = std::pow(eps, T(2)); A.s[1] = (T(2) * std::pow(eps_s, T(2))) * mdecay; A.s[2] = T(2) * std::pow(eps_s, T(3));
        A.s[3] = std::pow(eps_s, T(4)); A.s[4] = mdecay;
    } else if (sampler == 1) {
        T eps = (T)sc[0], Aa = (T)sc[1], scale_grad = (T)sc[2];
        T sgn = (scale_grad > T(0)) ? T(1) : ((scale_grad < T(0)) ? T(-1) : T(0));
        A.s[0] = eps; A.s[1] = Aa; A.s[2] = Aa - T(0); A.s[3] = T(2) * eps; A.s[4] = scale_grad + ((T(2) * sgn) * T(1e-16) + T(1e-16));
    } else {
        T eps = (T)sc[0], mass = (T)sc[1], cc = (T)sc[2], D = (T)sc[3], b_hat = (T)sc[4];
        A.s[0] = eps; A.s[1] = mass; A.s[2] = D; A.s[3] = (mass * mass) * (cc * cc); A.s[4] = std::sqrt(eps * ((T(2) * D) - (eps * b_hat)));
    }
    A.first_step = first_step; A.n_steps = n_steps; A.keep_every = keep_every; A.kept = kept;
    A.perpetual_adapt = burn_in_steps <= 0 ? 1 : 0;
    A.burn_in_steps = burn_in_steps > 0 ? (uint64_t)burn_in_steps : 0;
    const unsigned grid = (unsigned)((n_chains + 63) / 64);
    if (sampler == 0) hipLaunchKernelGGL((toy_chains_kernel<T, 0>), dim3(grid), dim3(64), 0, st, A);
    else if (sampler == 1) hipLaunchKernelGGL((toy_chains_kernel<T, 1>), dim3(grid), dim3(64), 0, st, A);
    else hipLaunchKernelGGL((toy_chains_kernel<T, 2>), dim3(grid), dim3(64), 0, st, A);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "launch toy_chains");
}

}  // namespace

extern "C" {

int sgmcmc_toy_chains_f32(int sampler, int target, const double *target_params, int k, float *theta, float *mom, float *tau,
                          float *g, float *v_hat, float *minv, size_t n_chains, int dim, const double *scalars,
                          const uint64_t *seeds, uint64_t first_step, uint64_t n_steps, int64_t burn_in_steps,
                          uint64_t keep_every, float *kept, sgmcmc_stream_t stream)
{
    return toy_chains<float>(sampler, target, target_params, k, theta, mom, tau, g, v_hat, minv, n_chains, dim, scalars, seeds,
                             first_step, n_steps, burn_in_steps, keep_every, kept, static_cast<hipStream_t>(stream));
}
int sgmcmc_toy_chains_f64(int sampler, int target, const double *target_params, int k, double *theta, double *mom, double *tau,
                          double *g, double *v_hat, double *minv, size_t n_chains, int dim, const double *scalars,
                          const uint64_t *seeds, uint64_t first_step, uint64_t n_steps, int64_t burn_in_steps,
                          uint64_t keep_every, double *kept, sgmcmc_stream_t stream)
{
    return toy_chains<double>(sampler, target, target_params, k, theta, mom, tau, g, v_hat, minv, n_chains, dim, scalars, seeds,
                              first_step, n_steps, burn_in_steps, keep_every, kept, static_cast<hipStream_t>(stream));
}

}  // extern "C"
