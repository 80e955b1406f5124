// sgmcmc_sghmc.hip -- K1, the fused SGHMC step (pysgmcmc/samplers/sghmc.py:165-251 + the burn-in switch
// pysgmcmc/samplers/base_classes.py:432-456): host side of sgmcmc_sghmc_step_{f32,f64} and sgmcmc_sghmc_scalars_*.
// The arithmetic is SghmcOp (sgmcmc_device.hpp), the kernel shape stream_quads_vec (sgmcmc_stream.hpp).
#include <cmath>

#include "sgmcmc_stream.hpp"

namespace {

// scalars of the reference graph, in the dtype, same op order (sghmc.py:111-117,211-217,235): {e2, c1, c3, e4, mdecay}
template <typename T>
void sghmc_scalars(T eps, T scale_grad, T mdecay, T (&s)[5])
{
    T eps_s = eps / std::sqrt(scale_grad);
    s[0] = std::pow(eps, T(2));
    s[1] = (T(2) * std::pow(eps_s, T(2))) * mdecay;
    s[2] = T(2) * std::pow(eps_s, T(3));
    s[3] = std::pow(eps_s, T(4));
    s[4] = mdecay;
}

template <typename T>
int sghmc_step(T *theta, T *V, const T *grad, T *tau, T *g, T *v_hat, T *minv, T *r, size_t n,
               T eps, T scale_grad, T mdecay, T grad_decay, int adapt, const T *xi, uint64_t seed, uint64_t step,
               const uint64_t *step_dev, void *stats_ws, const sgmcmc_step_opts_t *opts, const sgmcmc_launch_t *lc, hipStream_t st)
{
    if (n == 0) return 0;
    if (!theta || !V || !grad || !minv) return fail(SGMCMC_EINVAL, "sghmc_step: theta, V, grad and minv must be non-NULL");
    if (adapt && (!tau || !g || !v_hat)) return fail(SGMCMC_EINVAL, "sghmc_step: adapt=1 needs tau, g and v_hat");
    StepExtras<T> se;
    uint64_t first = 0;
    if (int rc = resolve_step_opts<T>(opts, n, stats_ws, se, first, "sghmc_step")) return rc;
    T s[5];
    sghmc_scalars<T>(eps, scale_grad, mdecay, s);
    const T *sdev = opts ? static_cast<const T *>(opts->scalars_dev) : nullptr;
    const bool skip_minv = adapt && opts && (opts->flags & SGMCMC_STEP_SKIP_MINV_STORE);
    NoiseKey nk = make_key(seed, step, step_dev, first);
    double *sp = static_cast<double *>(stats_ws);
    bool vec_ok = aligned16(theta) && aligned16(V) && aligned16(grad) && aligned16(minv) && aligned16(xi) &&
                  (!adapt || (aligned16(tau) && aligned16(g) && aligned16(v_hat) && aligned16(r))) &&
                  aligned16(se.ex.mom_mean) && aligned16(se.ex.mom_m2);
    bool mom_done = false, copy_done = false;
    se.copy_done = &copy_done;
    int rc;
#define SGHMC_GO(AD, INJ)                                                                                        \
    {                                                                                                            \
        SghmcOp<T, AD, INJ> op{theta, V, grad, tau, g, v_hat, minv, r, xi, s[0], s[1], s[2], s[3], s[4], grad_decay, nk, sp, \
                               skip_minv, sdev};                                                                 \
        rc = launch<SghmcOp<T, AD, INJ>, !INJ>(op, n, vec_ok, sizeof(T) * ((AD ? 12 : 6) + (INJ ? 1 : 0)), lc, se, &mom_done, st); \
    }
    if (adapt) { if (xi) SGHMC_GO(true, true) else SGHMC_GO(true, false) }
    else { if (xi) SGHMC_GO(false, true) else SGHMC_GO(false, false) }
#undef SGHMC_GO
    if (rc == 0 && se.want_moments && !mom_done) {        // no fused form for this path: the separate K4 pass, same arithmetic
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
int sghmc_scalars_store(T eps, T scale_grad, T mdecay, void *dst, hipStream_t st)
{
    if (!dst) return fail(SGMCMC_EINVAL, "sghmc_scalars: scalars_dev is NULL");
    T s[5];
    sghmc_scalars<T>(eps, scale_grad, mdecay, s);
    hipLaunchKernelGGL((store_scalars5<T>), dim3(1), dim3(1), 0, st, static_cast<T *>(dst), s[0], s[1], s[2], s[3], s[4]);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "launch store_scalars");
}

}  // namespace

extern "C" {

int sgmcmc_sghmc_step_f32(float *theta, float *V, const float *grad, float *tau, float *g, float *v_hat,
                          float *minv, float *r, size_t n, float eps, float scale_grad, float mdecay, float grad_decay, int adapt,
                          const float *xi, uint64_t seed, uint64_t step, const uint64_t *step_dev,
                          void *stats_ws, const sgmcmc_step_opts_t *opts, const sgmcmc_launch_t *launch, sgmcmc_stream_t stream)
{
    return sghmc_step<float>(theta, V, grad, tau, g, v_hat, minv, r, n, eps, scale_grad, mdecay, grad_decay, adapt, xi, seed, step,
                             step_dev, stats_ws, opts, launch, static_cast<hipStream_t>(stream));
}
int sgmcmc_sghmc_step_f64(double *theta, double *V, const double *grad, double *tau, double *g, double *v_hat,
                          double *minv, double *r, size_t n, double eps, double scale_grad, double mdecay, double grad_decay,
                          int adapt, const double *xi, uint64_t seed, uint64_t step, const uint64_t *step_dev,
                          void *stats_ws, const sgmcmc_step_opts_t *opts, const sgmcmc_launch_t *launch, sgmcmc_stream_t stream)
{
    return sghmc_step<double>(theta, V, grad, tau, g, v_hat, minv, r, n, eps, scale_grad, mdecay, grad_decay, adapt, xi, seed, step,
                              step_dev, stats_ws, opts, launch, static_cast<hipStream_t>(stream));
}
int sgmcmc_sghmc_scalars_f32(float eps, float scale_grad, float mdecay, void *scalars_dev, sgmcmc_stream_t stream)
{
    return sghmc_scalars_store<float>(eps, scale_grad, mdecay, scalars_dev, static_cast<hipStream_t>(stream));
}
int sgmcmc_sghmc_scalars_f64(double eps, double scale_grad, double mdecay, void *scalars_dev, sgmcmc_stream_t stream)
{
    return sghmc_scalars_store<double>(eps, scale_grad, mdecay, scalars_dev, static_cast<hipStream_t>(stream));
}

}  // extern "C"
