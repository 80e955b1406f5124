// sgmcmc_sgld.hip -- K2, the fused preconditioned SGLD step (pysgmcmc/samplers/sgld.py:149-211): host side of
// sgmcmc_sgld_step_{f32,f64} and sgmcmc_sgld_scalars_*. Arithmetic: SgldOp (sgmcmc_device.hpp).
#include <cmath>

#include "sgmcmc_stream.hpp"

namespace {

// {eps, A, a_eff, two_eps, sg_den}, sgld.py:106-108,186-191,201-204
template <typename T>
void sgld_scalars(T eps, T A, T scale_grad, T (&s)[5])
{
    T sgn = (scale_grad > T(0)) ? T(1) : ((scale_grad < T(0)) ? T(-1) : T(0));
    s[0] = eps;
    s[1] = A;
    s[2] = A - T(0);
    s[3] = T(2) * eps;
    s[4] = scale_grad + ((T(2) * sgn) * T(1e-16) + T(1e-16));
}

template <typename T>
int sgld_step(T *theta, const T *grad, T *tau, T *g, T *v_hat, T *minv, T *r, size_t n,
              T eps, T A, T scale_grad, T grad_decay, int adapt, const T *xi, uint64_t seed, uint64_t step, const uint64_t *step_dev,
              void *stats_ws, const sgmcmc_step_opts_t *opts, const sgmcmc_launch_t *lc, hipStream_t st)
{
    if (n == 0) return 0;
    if (!theta || !grad || !minv) return fail(SGMCMC_EINVAL, "sgld_step: theta, grad and minv must be non-NULL");
    if (adapt && (!tau || !g || !v_hat)) return fail(SGMCMC_EINVAL, "sgld_step: adapt=1 needs tau, g and v_hat");
    StepExtras<T> se;
    uint64_t first = 0;
    if (int rc = resolve_step_opts<T>(opts, n, stats_ws, se, first, "sgld_step")) return rc;
    T s[5];
    sgld_scalars<T>(eps, A, scale_grad, s);
    const T *sdev = opts ? static_cast<const T *>(opts->scalars_dev) : nullptr;
    const bool skip_minv = adapt && opts && (opts->flags & SGMCMC_STEP_SKIP_MINV_STORE);
    NoiseKey nk = make_key(seed, step, step_dev, first);
    double *sp = static_cast<double *>(stats_ws);
    bool vec_ok = aligned16(theta) && aligned16(grad) && aligned16(minv) && aligned16(xi) &&
                  (!adapt || (aligned16(tau) && aligned16(g) && aligned16(v_hat) && aligned16(r))) &&
                  aligned16(se.ex.mom_mean) && aligned16(se.ex.mom_m2);
    bool mom_done = false, copy_done = false;
    se.copy_done = &copy_done;
    int rc;
#define SGLD_GO(AD, INJ)                                                                                         \
    {                                                                                                            \
        SgldOp<T, AD, INJ> op{theta, grad, tau, g, v_hat, minv, r, xi, s[0], s[1], s[2], s[3], s[4], grad_decay, nk, sp, \
                              skip_minv, sdev};                                                                  \
        rc = launch<SgldOp<T, AD, INJ>, !INJ>(op, n, vec_ok, sizeof(T) * ((AD ? 10 : 4) + (INJ ? 1 : 0)), lc, se, &mom_done, st); \
    }
    if (adapt) { if (xi) SGLD_GO(true, true) else SGLD_GO(true, false) }
    else { if (xi) SGLD_GO(false, true) else SGLD_GO(false, false) }
#undef SGLD_GO
    if (rc == 0 && se.want_moments && !mom_done) {
        MomentsOp<T> mop{theta, se.ex.mom_mean, se.ex.mom_m2, se.ex.mom_inv};
        sgmcmc_launch_t lc_mom = lc ? *lc : sgmcmc_launch_t{};      // same geometry, but NOT the caller's timestamp events: they
        lc_mom.start_event = lc_mom.stop_event = nullptr;           // belong to the step kernel above
        rc = launch(mop, n, aligned16(theta) && aligned16(se.ex.mom_mean) && aligned16(se.ex.mom_m2), 5 * sizeof(T), lc ? &lc_mom : nullptr, st);
    }
    if (rc == 0) rc = finish_side_copy<T>(se, copy_done, st);      // opts.gather_* on a path without a fused form
    return rc;
}

template <typename T>
__global__ void store_scalars5(T *dst, T a, T b, T c, T d, T e) { dst[0] = a; dst[1] = b; dst[2] = c; dst[3] = d; dst[4] = e; }

template <typename T>
int sgld_scalars_store(T eps, T A, T scale_grad, void *dst, hipStream_t st)
{
    if (!dst) return fail(SGMCMC_EINVAL, "sgld_scalars: scalars_dev is NULL");
    T s[5];
    sgld_scalars<T>(eps, A, scale_grad, s);
    hipLaunchKernelGGL((store_scalars5<T>), dim3(1), dim3(1), 0, st, static_cast<T *>(dst), s[0], s[1], s[2], s[3], s[4]);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "launch store_scalars");
}

}  // namespace

extern "C" {

int sgmcmc_sgld_step_f32(float *theta, const float *grad, float *tau, float *g, float *v_hat, float *minv, float *r,
                         size_t n, float eps, float A, float scale_grad, float grad_decay, int adapt, const float *xi,
                         uint64_t seed, uint64_t step, const uint64_t *step_dev, void *stats_ws,
                         const sgmcmc_step_opts_t *opts, const sgmcmc_launch_t *launch, sgmcmc_stream_t stream)
{
    return sgld_step<float>(theta, grad, tau, g, v_hat, minv, r, n, eps, A, scale_grad, grad_decay, adapt, xi, seed, step, step_dev,
                            stats_ws, opts, launch, static_cast<hipStream_t>(stream));
}
int sgmcmc_sgld_step_f64(double *theta, const double *grad, double *tau, double *g, double *v_hat, double *minv,
                         double *r, size_t n, double eps, double A, double scale_grad, double grad_decay, int adapt,
                         const double *xi, uint64_t seed, uint64_t step, const uint64_t *step_dev, void *stats_ws,
                         const sgmcmc_step_opts_t *opts, const sgmcmc_launch_t *launch, sgmcmc_stream_t stream)
{
    return sgld_step<double>(theta, grad, tau, g, v_hat, minv, r, n, eps, A, scale_grad, grad_decay, adapt, xi, seed, step, step_dev,
                             stats_ws, opts, launch, static_cast<hipStream_t>(stream));
}
int sgmcmc_sgld_scalars_f32(float eps, float A, float scale_grad, void *scalars_dev, sgmcmc_stream_t stream)
{
    return sgld_scalars_store<float>(eps, A, scale_grad, scalars_dev, static_cast<hipStream_t>(stream));
}
int sgmcmc_sgld_scalars_f64(double eps, double A, double scale_grad, void *scalars_dev, sgmcmc_stream_t stream)
{
    return sgld_scalars_store<double>(eps, A, scale_grad, scalars_dev, static_cast<hipStream_t>(stream));
}

}  // extern "C"
