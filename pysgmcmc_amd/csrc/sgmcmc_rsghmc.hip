// sgmcmc_rsghmc.hip -- K3, the fused relativistic SGHMC step (pysgmcmc/samplers/relativistic_sghmc.py:120-140): host
// side of sgmcmc_rsghmc_step_{f32,f64} and sgmcmc_rsghmc_scalars_*. Arithmetic: RsghmcOp (sgmcmc_device.hpp).
#include <cmath>

#include "sgmcmc_stream.hpp"

namespace {

// {eps, mass, D, m2c2, nscale}, relativistic_sghmc.py:105-106,117-125
template <typename T>
void rsghmc_scalars(T eps, T mass, T c, T D, T b_hat, T (&s)[5])
{
    s[0] = eps;
    s[1] = mass;
    s[2] = D;
    s[3] = (mass * mass) * (c * c);
    s[4] = std::sqrt(eps * ((T(2) * D) - (eps * b_hat)));
}

template <typename T>
int rsghmc_step(T *theta, T *p, const T *grad, size_t n, T eps, T mass, T c, T D, T b_hat, T grad_decay,
                const T *xi, uint64_t seed, uint64_t step, const uint64_t *step_dev,
                void *stats_ws, const sgmcmc_step_opts_t *opts, const sgmcmc_launch_t *lc, hipStream_t st)
{
    if (n == 0) return 0;
    if (!theta || !p || !grad) return fail(SGMCMC_EINVAL, "rsghmc_step: theta, p and grad_cost must be non-NULL");
    StepExtras<T> se;
    uint64_t first = 0;
    if (int rc = resolve_step_opts<T>(opts, n, stats_ws, se, first, "rsghmc_step")) return rc;
    T s[5];
    rsghmc_scalars<T>(eps, mass, c, D, b_hat, s);
    const T *sdev = opts ? static_cast<const T *>(opts->scalars_dev) : nullptr;
    NoiseKey nk = make_key(seed, step, step_dev, first);
    double *sp = static_cast<double *>(stats_ws);
    bool vec_ok = aligned16(theta) && aligned16(p) && aligned16(grad) && aligned16(xi) &&
                  aligned16(se.ex.mom_mean) && aligned16(se.ex.mom_m2);
    bool mom_done = false, copy_done = false;
    se.copy_done = &copy_done;
    int rc;
    // m^2 c^2 a power of two (the default m = c = 1): the divisions by it are exact multiplications (RsghmcOp POW2). Not with
    // device-resident scalars: the block may be refreshed with another mass / c after this launch was captured.
    int e2 = 0;
    const T inv = T(1) / s[3];
    const bool pow2 = sdev == nullptr && s[3] > T(0) && std::isfinite(s[3]) && std::frexp(s[3], &e2) == T(0.5) &&
                      std::isnormal(inv) && std::isnormal(s[3]);
#define RSGHMC_GO(P2, INJ)                                                                                        \
    {                                                                                                             \
        RsghmcOp<T, P2, INJ> op{theta, p, grad, xi, s[0], s[1], s[2], s[3], s[4], grad_decay, nk, sp, sdev, inv};  \
        rc = launch<RsghmcOp<T, P2, INJ>, !INJ>(op, n, vec_ok, sizeof(T) * (INJ ? 6 : 5), lc, se, &mom_done, st);  \
    }
    if (pow2) { if (xi) RSGHMC_GO(true, true) else RSGHMC_GO(true, false) }
    else { if (xi) RSGHMC_GO(false, true) else RSGHMC_GO(false, false) }
#undef RSGHMC_GO
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
int rsghmc_scalars_store(T eps, T mass, T c, T D, T b_hat, void *dst, hipStream_t st)
{
    if (!dst) return fail(SGMCMC_EINVAL, "rsghmc_scalars: scalars_dev is NULL");
    T s[5];
    rsghmc_scalars<T>(eps, mass, c, D, b_hat, s);
    hipLaunchKernelGGL((store_scalars5<T>), dim3(1), dim3(1), 0, st, static_cast<T *>(dst), s[0], s[1], s[2], s[3], s[4]);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "launch store_scalars");
}

}  // namespace

extern "C" {

int sgmcmc_rsghmc_step_f32(float *theta, float *p, const float *grad_cost, size_t n, float eps, float mass, float c,
                           float D, float b_hat, float grad_decay, const float *xi, uint64_t seed, uint64_t step,
                           const uint64_t *step_dev, void *stats_ws, const sgmcmc_step_opts_t *opts,
                           const sgmcmc_launch_t *launch, sgmcmc_stream_t stream)
{
    return rsghmc_step<float>(theta, p, grad_cost, n, eps, mass, c, D, b_hat, grad_decay, xi, seed, step, step_dev,
                              stats_ws, opts, launch, static_cast<hipStream_t>(stream));
}
int sgmcmc_rsghmc_step_f64(double *theta, double *p, const double *grad_cost, size_t n, double eps, double mass,
                           double c, double D, double b_hat, double grad_decay, const double *xi, uint64_t seed,
                           uint64_t step, const uint64_t *step_dev, void *stats_ws, const sgmcmc_step_opts_t *opts,
                           const sgmcmc_launch_t *launch, sgmcmc_stream_t stream)
{
    return rsghmc_step<double>(theta, p, grad_cost, n, eps, mass, c, D, b_hat, grad_decay, xi, seed, step, step_dev,
                               stats_ws, opts, launch, static_cast<hipStream_t>(stream));
}
int sgmcmc_rsghmc_scalars_f32(float eps, float mass, float c, float D, float b_hat, void *scalars_dev, sgmcmc_stream_t stream)
{
    return rsghmc_scalars_store<float>(eps, mass, c, D, b_hat, scalars_dev, static_cast<hipStream_t>(stream));
}
int sgmcmc_rsghmc_scalars_f64(double eps, double mass, double c, double D, double b_hat, void *scalars_dev, sgmcmc_stream_t stream)
{
    return rsghmc_scalars_store<double>(eps, mass, c, D, b_hat, scalars_dev, static_cast<hipStream_t>(stream));
}

}  // extern "C"
