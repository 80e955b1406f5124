// sgmcmc_device.hpp -- device-side building blocks shared by the kernel translation units of
// libsgmcmc_hip.so: quad memory access, Philox4x32-10 + Box-Muller, the reference scalar helpers
// (safe_divide / safe_sqrt), the per-sampler quad operators (the update arithmetic, one IEEE rounding
// per reference op) and the DPP wave reduction. Everything lives in an anonymous namespace: each
// translation unit gets its own copy. Build with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#pragma clang fp contract(off)

namespace {

// --------------------------------------------------------------------------
// vector types and quad memory access
// --------------------------------------------------------------------------

// experiment knobs (build-time): which side of an nt launch actually carries the nt hint
#ifndef SGMCMC_NT_LOADS
#define SGMCMC_NT_LOADS 1
#endif
#ifndef SGMCMC_NT_STORES
#define SGMCMC_NT_STORES 1
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

template <bool NT>
__device__ __forceinline__ void load_quad(const float *__restrict__ p, size_t q, float (&v)[4])
{
    const f32x4 *p4 = reinterpret_cast<const f32x4 *>(p) + q;
    f32x4 t = (NT && SGMCMC_NT_LOADS) ? __builtin_nontemporal_load(p4) : *p4;
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <bool NT>
__device__ __forceinline__ void store_quad(float *__restrict__ p, size_t q, const float (&v)[4])
{
    f32x4 t = {v[0], v[1], v[2], v[3]};
    f32x4 *p4 = reinterpret_cast<f32x4 *>(p) + q;
    if (NT && SGMCMC_NT_STORES) __builtin_nontemporal_store(t, p4); else *p4 = t;
}
template <bool NT>
__device__ __forceinline__ void load_quad(const double *__restrict__ p, size_t q, double (&v)[4])
{
    const f64x2 *p2 = reinterpret_cast<const f64x2 *>(p) + 2 * q;
    f64x2 a = (NT && SGMCMC_NT_LOADS) ? __builtin_nontemporal_load(p2) : p2[0];
    f64x2 b = (NT && SGMCMC_NT_LOADS) ? __builtin_nontemporal_load(p2 + 1) : p2[1];
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
}
template <bool NT>
__device__ __forceinline__ void store_quad(double *__restrict__ p, size_t q, const double (&v)[4])
{
    f64x2 a = {v[0], v[1]}, b = {v[2], v[3]};
    f64x2 *p2 = reinterpret_cast<f64x2 *>(p) + 2 * q;
    if (NT && SGMCMC_NT_STORES) { __builtin_nontemporal_store(a, p2); __builtin_nontemporal_store(b, p2 + 1); }
    else { p2[0] = a; p2[1] = b; }
}
// element-wise access for misaligned arrays and the ragged tail (cnt in 1..4)
template <typename T>
__device__ __forceinline__ void load_part(const T *__restrict__ p, size_t q, int cnt, T (&v)[4])
{
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (j < cnt) ? p[4 * q + j] : T(1);
}
template <typename T>
__device__ __forceinline__ void store_part(T *__restrict__ p, size_t q, int cnt, const T (&v)[4])
{
#pragma unroll
    for (int j = 0; j < 4; ++j) if (j < cnt) p[4 * q + j] = v[j];
}

// --------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11) + Box-Muller, all in registers
// --------------------------------------------------------------------------

// key = seed, (s0,s1) = step. step_dev (nullable) is a device-resident counter added to
// the by-value step when the kernel starts: a hipGraph replays identical kernel
// arguments, so graph-captured chains advance their noise stream through it.
struct NoiseKey {
    uint32_t k0, k1, s0, s1;
    const uint64_t *step_dev;
    // quad index of the launch's first element within the chain's parameter vector: a launch over a SLICE
    // [first, first + n) of the arena draws the quads the single launch over the whole arena would draw
    uint64_t q0 = 0;
    __device__ __forceinline__ void resolve()
    {
        if (step_dev) {
            uint64_t st = (((uint64_t)s1 << 32) | s0) + *step_dev;
            s0 = (uint32_t)st; s1 = (uint32_t)(st >> 32);
        }
    }
};

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&x)[4])
{
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    x[0] = c0; x[1] = c1; x[2] = c2; x[3] = c3;
}

__device__ __forceinline__ void philox_quad(const NoiseKey &nk, size_t q, uint32_t (&x)[4])
{
    const uint64_t gq = (uint64_t)q + nk.q0;
    philox4x32_10(nk.s0, nk.s1, (uint32_t)gq, (uint32_t)(gq >> 32), nk.k0, nk.k1, x);
}

// 4 standard normals for quad q. f32: hardware transcendentals
// (v_log_f32 = log2, v_sin/cos_f32 take revolutions).
__device__ __forceinline__ void normal_quad(const NoiseKey &nk, size_t q, float (&z)[4])
{
    uint32_t x[4];
    philox_quad(nk, q, x);
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        float u = __builtin_fmaf((float)x[2 * pr], 0x1p-32f, 0x1p-33f);        // (0,1]
        float rev = __builtin_fmaf((float)x[2 * pr + 1], 0x1p-32f, 0x1p-33f);  // (0,1]
        float s = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u));
        z[2 * pr] = s * __builtin_amdgcn_sinf(rev);
        z[2 * pr + 1] = s * __builtin_amdgcn_cosf(rev);
    }
}
__device__ __forceinline__ void normal_quad(const NoiseKey &nk, size_t q, double (&z)[4])
{
    uint32_t x[4];
    philox_quad(nk, q, x);
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        double u = ((double)x[2 * pr] + 0.5) * 0x1p-32;
        double rev = ((double)x[2 * pr + 1] + 0.5) * 0x1p-32;
        double s = sqrt(-2.0 * log(u));
        double sn, cs;
        sincospi(2.0 * rev, &sn, &cs);
        z[2 * pr] = s * sn;
        z[2 * pr + 1] = s * cs;
    }
}

// --------------------------------------------------------------------------
// reference scalar helpers (pysgmcmc/tensor_utils.py:269, :319-323)
// --------------------------------------------------------------------------

template <typename T> __device__ __forceinline__ T rsqrt_rn(T x);
template <> __device__ __forceinline__ float rsqrt_rn<float>(float x) { return sqrtf(x); }
template <> __device__ __forceinline__ double rsqrt_rn<double>(double x) { return sqrt(x); }

template <typename T>
__device__ __forceinline__ T sdiv(T x, T y)
{
    const T sc = T(1e-16);
    T sgn = (y > T(0)) ? T(1) : ((y < T(0)) ? T(-1) : T(0));
    T delta = (T(2) * sgn) * sc + sc;
    return x / (y + delta);
}
template <typename T>
__device__ __forceinline__ T ssqrt(T x)
{
    T c = (x > T(0)) ? x : T(0);
    c = (c < (T)INFINITY) ? c : (T)INFINITY;
    return rsqrt_rn<T>(c);
}

// burn-in statistics, sghmc.py:168-196 == sgld.py:154-180 (all reads are of OLD state)
template <typename T>
__device__ __forceinline__ T adapt_stats(T grad, T &tau, T &g, T &vh, T &r_out)
{
    T tau0 = tau, g0 = g, vh0 = vh;
    T r = T(1) / (tau0 + T(1));
    T tau1 = tau0 + (sdiv<T>(((-g0) * g0) * tau0, vh0) + T(1));
    T minv = sdiv<T>(T(1), ssqrt<T>(vh0));
    T g1 = g0 + ((-r) * g0 + r * grad);
    T vh1 = vh0 + ((-r) * vh0 + r * (grad * grad));
    tau = tau1; g = g1; vh = vh1; r_out = r;
    return minv;
}

// One parameter's SGHMC update given its (possibly decayed) gradient gr, preconditioner mi and normal draw z
// (sghmc.py:211-243 with fed / freshly adapted minv). Shared by the streaming operator below and by the epilogue of the
// weight-gradient GEMM (sgmcmc_gemm.hip), so both give the same bits.
template <typename T>
__device__ __forceinline__ void sghmc_elem_update(T &th, T &v, T gr, T mi, T z, T e2, T c1, T c3, T e4, T mdecay)
{
    T noise_scale = (c1 * mi - (c3 * (mi * mi)) * T(0)) - e4;                            // :211-217
    T sigma = rsqrt_rn<T>((noise_scale > T(1e-16)) ? noise_scale : T(1e-16));             // :220
    T sample = sigma * z;
    T v0 = v;
    T v1 = v0 + (((((-e2) * mi) * gr) - mdecay * v0) + sample);                           // :233-238
    v = v1;
    th = th + v1;                                                                         // :241-243
}

// --------------------------------------------------------------------------
// per-sampler quad operators
// --------------------------------------------------------------------------

template <typename T, bool ADAPT, bool INJECT>
struct SghmcOp {
    typedef T real;
    T *theta, *V; const T *grad; T *tau, *g, *vh, *minv, *r; const T *xi;
    T e2, c1, c3, e4, mdecay;      // host-derived scalars, sghmc.py:111-117,211-217,235
    T grad_decay;                  // gradient term grad_decay * theta added in registers (0 = off)
    NoiseKey nk;
    double *stats_part;            // nullable: per-block partials of {sum theta'^2, sum V'^2, sum minv, sum minv^2}
    bool skip_minv = false;        // ADAPT: do not write minv this step (44 instead of 48 B/param; see sgmcmc_step_opts_t)
    // nullable: DEVICE copy of {e2, c1, c3, e4, mdecay} that overrides the by-value scalars (a hipGraph-captured
    // launch replays its arguments; a scheduled stepsize reaches it through this block, sgmcmc_sghmc_scalars_*)
    const T *scalars_dev = nullptr;
    static constexpr unsigned stats_mask = 0xfu;     // which of the 4 statistics this operator produces
    __device__ __forceinline__ void prepare()
    {
        nk.resolve();
        if (scalars_dev) { e2 = scalars_dev[0]; c1 = scalars_dev[1]; c3 = scalars_dev[2]; e4 = scalars_dev[3]; mdecay = scalars_dev[4]; }
    }
    // TSQ_ONLY: only sum theta'^2 (the one statistic the BNN loss head consumes)
    template <bool TSQ_ONLY, typename RegsT, typename ACC>
    __device__ __forceinline__ void accumulate(const RegsT &R, int cnt, ACC (&acc)[4]) const
    {
        // the quad's 4 terms are summed in T (4 adds), the running totals in double
        T s0 = T(0), s1 = T(0), s2 = T(0), s3 = T(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) if (j < cnt) {
            T th = R.th[j], v = R.v[j], mi = R.mi[j];
            s0 += th * th;
            if constexpr (!TSQ_ONLY) { s1 += v * v; s2 += mi; s3 += mi * mi; }
        }
        acc[0] += (ACC)s0;
        if constexpr (!TSQ_ONLY) { acc[1] += (ACC)s1; acc[2] += (ACC)s2; acc[3] += (ACC)s3; }
    }
    struct Regs { T th[4], v[4], gr[4], mi[4], tau[4], g[4], vh[4], rr[4], z[4]; };

    template <bool NT> __device__ __forceinline__ void load_vec(size_t q, Regs &R) const
    {
        load_quad<NT>(theta, q, R.th); load_quad<NT>(V, q, R.v); load_quad<NT>(grad, q, R.gr);
        if (ADAPT) { load_quad<NT>(tau, q, R.tau); load_quad<NT>(g, q, R.g); load_quad<NT>(vh, q, R.vh); }
        else load_quad<NT>(minv, q, R.mi);
        if (INJECT) load_quad<NT>(xi, q, R.z);
    }
    __device__ __forceinline__ void load_part_(size_t q, int cnt, Regs &R) const
    {
        load_part(theta, q, cnt, R.th); load_part(V, q, cnt, R.v); load_part(grad, q, cnt, R.gr);
        if (ADAPT) { load_part(tau, q, cnt, R.tau); load_part(g, q, cnt, R.g); load_part(vh, q, cnt, R.vh); }
        else load_part(minv, q, cnt, R.mi);
        if (INJECT) load_part(xi, q, cnt, R.z);
    }
    __device__ __forceinline__ void compute(size_t q, Regs &R) const
    {
        if (!INJECT) normal_quad(nk, q, R.z);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            T gr = (grad_decay != T(0)) ? R.gr[j] + grad_decay * R.th[j] : R.gr[j];
            T mi;
            if constexpr (ADAPT) { mi = adapt_stats<T>(gr, R.tau[j], R.g[j], R.vh[j], R.rr[j]); R.mi[j] = mi; }
            else mi = R.mi[j];
            sghmc_elem_update<T>(R.th[j], R.v[j], gr, mi, R.z[j], e2, c1, c3, e4, mdecay);
        }
    }
    template <bool NT> __device__ __forceinline__ void store_vec(size_t q, const Regs &R) const
    {
        store_quad<NT>(theta, q, R.th); store_quad<NT>(V, q, R.v);
        if (ADAPT) {
            store_quad<NT>(tau, q, R.tau); store_quad<NT>(g, q, R.g); store_quad<NT>(vh, q, R.vh);
            if (!skip_minv) store_quad<NT>(minv, q, R.mi);
            if (r) store_quad<NT>(r, q, R.rr);
        }
    }
    __device__ __forceinline__ void store_part_(size_t q, int cnt, const Regs &R) const
    {
        store_part(theta, q, cnt, R.th); store_part(V, q, cnt, R.v);
        if (ADAPT) {
            store_part(tau, q, cnt, R.tau); store_part(g, q, cnt, R.g); store_part(vh, q, cnt, R.vh);
            if (!skip_minv) store_part(minv, q, cnt, R.mi);
            if (r) store_part(r, q, cnt, R.rr);
        }
    }
};

template <typename T, bool ADAPT, bool INJECT>
struct SgldOp {
    typedef T real;
    T *theta; const T *grad; T *tau, *g, *vh, *minv, *r; const T *xi;
    T eps, A, a_eff, two_eps, sg_den;     // sgld.py:106-108,186-191,201-204
    T grad_decay;
    NoiseKey nk;
    double *stats_part;
    bool skip_minv = false;
    const T *scalars_dev = nullptr;       // nullable device copy of {eps, A, a_eff, two_eps, sg_den} (see SghmcOp)
    static constexpr unsigned stats_mask = 0xdu;     // no momentum: {theta'^2, -, minv, minv^2}
    __device__ __forceinline__ void prepare()
    {
        nk.resolve();
        if (scalars_dev) { eps = scalars_dev[0]; A = scalars_dev[1]; a_eff = scalars_dev[2]; two_eps = scalars_dev[3]; sg_den = scalars_dev[4]; }
    }
    template <bool TSQ_ONLY, typename RegsT, typename ACC>
    __device__ __forceinline__ void accumulate(const RegsT &R, int cnt, ACC (&acc)[4]) const
    {
        T s0 = T(0), s2 = T(0), s3 = T(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) if (j < cnt) {
            T th = R.th[j], mi = R.mi[j];
            s0 += th * th;
            if constexpr (!TSQ_ONLY) { s2 += mi; s3 += mi * mi; }
        }
        acc[0] += (ACC)s0;
        if constexpr (!TSQ_ONLY) { acc[2] += (ACC)s2; acc[3] += (ACC)s3; }
    }
    struct Regs { T th[4], gr[4], mi[4], tau[4], g[4], vh[4], rr[4], z[4]; };

    template <bool NT> __device__ __forceinline__ void load_vec(size_t q, Regs &R) const
    {
        load_quad<NT>(theta, q, R.th); load_quad<NT>(grad, q, R.gr);
        if (ADAPT) { load_quad<NT>(tau, q, R.tau); load_quad<NT>(g, q, R.g); load_quad<NT>(vh, q, R.vh); }
        else load_quad<NT>(minv, q, R.mi);
        if (INJECT) load_quad<NT>(xi, q, R.z);
    }
    __device__ __forceinline__ void load_part_(size_t q, int cnt, Regs &R) const
    {
        load_part(theta, q, cnt, R.th); load_part(grad, q, cnt, R.gr);
        if (ADAPT) { load_part(tau, q, cnt, R.tau); load_part(g, q, cnt, R.g); load_part(vh, q, cnt, R.vh); }
        else load_part(minv, q, cnt, R.mi);
        if (INJECT) load_part(xi, q, cnt, R.z);
    }
    __device__ __forceinline__ void compute(size_t q, Regs &R) const
    {
        if (!INJECT) normal_quad(nk, q, R.z);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            T gr = (grad_decay != T(0)) ? R.gr[j] + grad_decay * R.th[j] : R.gr[j];
            T mi;
            if constexpr (ADAPT) { mi = adapt_stats<T>(gr, R.tau[j], R.g[j], R.vh[j], R.rr[j]); R.mi[j] = mi; }
            else mi = R.mi[j];
            T sigma = ssqrt<T>(two_eps * ((mi * a_eff) / sg_den));                        // :186-191
            T sample = sigma * R.z[j];
            R.th[j] = R.th[j] + (((((-eps) * mi) * A) * gr) + sample);                    // :201-204
        }
    }
    template <bool NT> __device__ __forceinline__ void store_vec(size_t q, const Regs &R) const
    {
        store_quad<NT>(theta, q, R.th);
        if (ADAPT) {
            store_quad<NT>(tau, q, R.tau); store_quad<NT>(g, q, R.g); store_quad<NT>(vh, q, R.vh);
            if (!skip_minv) store_quad<NT>(minv, q, R.mi);
            if (r) store_quad<NT>(r, q, R.rr);
        }
    }
    __device__ __forceinline__ void store_part_(size_t q, int cnt, const Regs &R) const
    {
        store_part(theta, q, cnt, R.th);
        if (ADAPT) {
            store_part(tau, q, cnt, R.tau); store_part(g, q, cnt, R.g); store_part(vh, q, cnt, R.vh);
            if (!skip_minv) store_part(minv, q, cnt, R.mi);
            if (r) store_part(r, q, cnt, R.rr);
        }
    }
};

// POW2: m^2 c^2 is a power of two (the reference's default m = c = 1, relativistic_sghmc.py:24-27): x / m2c2 and
// x * (1 / m2c2) are then the SAME correctly rounded value, so the two divisions by m2c2 of :123 / :131 become multiplications
// -- bit-identical results, 2 of the 4 divisions per element gone. The kernel is VALU-bound, not HBM-bound (SQ counters,
// profiles/r04_k3_counters.txt: its vector ALU is active 87-93 % of the launch), so instructions are time here.
template <typename T, bool POW2, bool INJECT>
struct RsghmcOp {
    typedef T real;
    T *theta, *p; const T *grad; const T *xi;
    T eps, mass, D, m2c2, nscale;         // relativistic_sghmc.py:105-106,117-125
    T grad_decay;
    NoiseKey nk;
    double *stats_part;
    const T *scalars_dev = nullptr;       // nullable device copy of {eps, mass, D, m2c2, nscale} (see SghmcOp)
    T inv_m2c2 = T(0);                    // POW2 only: 1 / m2c2 (exact)
    static constexpr unsigned stats_mask = 0x3u;     // {theta'^2, p'^2}
    __device__ __forceinline__ void prepare()
    {
        nk.resolve();
        if (scalars_dev) { eps = scalars_dev[0]; mass = scalars_dev[1]; D = scalars_dev[2]; m2c2 = scalars_dev[3]; nscale = scalars_dev[4]; }
    }
    template <bool TSQ_ONLY, typename RegsT, typename ACC>
    __device__ __forceinline__ void accumulate(const RegsT &R, int cnt, ACC (&acc)[4]) const
    {
        T s0 = T(0), s1 = T(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) if (j < cnt) {
            T th = R.th[j], pp = R.p[j];
            s0 += th * th;
            if constexpr (!TSQ_ONLY) s1 += pp * pp;
        }
        acc[0] += (ACC)s0;
        if constexpr (!TSQ_ONLY) acc[1] += (ACC)s1;
    }
    struct Regs { T th[4], p[4], gr[4], z[4]; };

    template <bool NT> __device__ __forceinline__ void load_vec(size_t q, Regs &R) const
    {
        load_quad<NT>(theta, q, R.th); load_quad<NT>(p, q, R.p); load_quad<NT>(grad, q, R.gr);
        if (INJECT) load_quad<NT>(xi, q, R.z);
    }
    __device__ __forceinline__ void load_part_(size_t q, int cnt, Regs &R) const
    {
        load_part(theta, q, cnt, R.th); load_part(p, q, cnt, R.p); load_part(grad, q, cnt, R.gr);
        if (INJECT) load_part(xi, q, cnt, R.z);
    }
    __device__ __forceinline__ T over_m2c2(T x) const { return POW2 ? x * inv_m2c2 : x / m2c2; }
    __device__ __forceinline__ void compute(size_t q, Regs &R) const
    {
        if (!INJECT) normal_quad(nk, q, R.z);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            T p0 = R.p[j];
            T gl = -((grad_decay != T(0)) ? R.gr[j] + grad_decay * R.th[j] : R.gr[j]);           // :100-103
            T pg = (eps * p0) / (mass * rsqrt_rn<T>(over_m2c2(p0 * p0) + T(1)));         // :123
            T nz = nscale * R.z[j];                                                      // :125
            T p1 = p0 + (((eps * gl) + nz) - (D * pg));                                  // :126-129
            T pg1 = (eps * p1) / (mass * rsqrt_rn<T>(over_m2c2(p1 * p1) + T(1)));        // :131
            R.p[j] = p1;
            R.th[j] = R.th[j] + pg1;                                                     // :132-135
        }
    }
    template <bool NT> __device__ __forceinline__ void store_vec(size_t q, const Regs &R) const
    {
        store_quad<NT>(theta, q, R.th); store_quad<NT>(p, q, R.p);
    }
    __device__ __forceinline__ void store_part_(size_t q, int cnt, const Regs &R) const
    {
        store_part(theta, q, cnt, R.th); store_part(p, q, cnt, R.p);
    }
};

template <typename T>
struct NormalFillOp {
    typedef T real;
    T *out; NoiseKey nk;
    static constexpr double *stats_part = nullptr;
    static constexpr unsigned stats_mask = 0u;
    __device__ __forceinline__ void prepare() { nk.resolve(); }
    struct Regs { T z[4]; };
    template <bool TSQ_ONLY, typename ACC> __device__ __forceinline__ void accumulate(const Regs &, int, ACC (&)[4]) const {}
    template <bool NT> __device__ __forceinline__ void load_vec(size_t, Regs &) const {}
    __device__ __forceinline__ void load_part_(size_t, int, Regs &) const {}
    __device__ __forceinline__ void compute(size_t q, Regs &R) const { normal_quad(nk, q, R.z); }
    template <bool NT> __device__ __forceinline__ void store_vec(size_t q, const Regs &R) const { store_quad<NT>(out, q, R.z); }
    __device__ __forceinline__ void store_part_(size_t q, int cnt, const Regs &R) const { store_part(out, q, cnt, R.z); }
};

template <typename T>
struct MomentsOp {
    typedef T real;
    const T *theta; T *mean, *m2; T inv;
    static constexpr double *stats_part = nullptr;
    static constexpr unsigned stats_mask = 0u;
    __device__ __forceinline__ void prepare() {}
    struct Regs { T x[4], mu[4], m2[4]; };
    template <bool TSQ_ONLY, typename ACC> __device__ __forceinline__ void accumulate(const Regs &, int, ACC (&)[4]) const {}
    template <bool NT> __device__ __forceinline__ void load_vec(size_t q, Regs &R) const
    { load_quad<NT>(theta, q, R.x); load_quad<NT>(mean, q, R.mu); load_quad<NT>(m2, q, R.m2); }
    __device__ __forceinline__ void load_part_(size_t q, int cnt, Regs &R) const
    { load_part(theta, q, cnt, R.x); load_part(mean, q, cnt, R.mu); load_part(m2, q, cnt, R.m2); }
    __device__ __forceinline__ void compute(size_t, Regs &R) const
    {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            T x = R.x[j];
            T d = x - R.mu[j];
            T mu = R.mu[j] + d * inv;
            R.mu[j] = mu;
            R.m2[j] = R.m2[j] + d * (x - mu);
        }
    }
    template <bool NT> __device__ __forceinline__ void store_vec(size_t q, const Regs &R) const
    { store_quad<NT>(mean, q, R.mu); store_quad<NT>(m2, q, R.m2); }
    __device__ __forceinline__ void store_part_(size_t q, int cnt, const Regs &R) const
    { store_part(mean, q, cnt, R.mu); store_part(m2, q, cnt, R.m2); }
};

// One DPP data-movement step on a double (two 32-bit halves). Lanes the control word / row mask
// leaves without a source receive 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}

// Wave64 sum with DPP cross-lane moves (VALU, no LDS crossbar traffic like __shfl/ds_bpermute):
// quad_perm [1,0,3,2], [2,3,0,1], row_shr:4, row_shr:8, row_bcast:15 (rows 1,3), row_bcast:31
// (rows 2,3). The total ends up in lane 63; fixed association => deterministic.
__device__ __forceinline__ double wave_sum_dpp_lane63(double v)
{
    v += dpp_mov_f64<0xB1, 0xf>(v);
    v += dpp_mov_f64<0x4E, 0xf>(v);
    v += dpp_mov_f64<0x114, 0xf>(v);
    v += dpp_mov_f64<0x118, 0xf>(v);
    v += dpp_mov_f64<0x142, 0xa>(v);
    v += dpp_mov_f64<0x143, 0xc>(v);
    return v;
}

// The same in single precision: ONE v_add_f32 with a DPP operand per step (6 instructions per statistic instead of
// 18 for a double). Used by the f32 step kernels: every lane contributes the f32 sum of one quad's 4 terms, the 64-lane
// tree adds at most 6 roundings (relative error < 4e-7, typically 1e-7); blocks and launches are combined in double.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov_f32(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum_dpp_lane63(float v)
{
    v += dpp_mov_f32<0xB1, 0xf>(v);
    v += dpp_mov_f32<0x4E, 0xf>(v);
    v += dpp_mov_f32<0x114, 0xf>(v);
    v += dpp_mov_f32<0x118, 0xf>(v);
    v += dpp_mov_f32<0x142, 0xa>(v);
    v += dpp_mov_f32<0x143, 0xc>(v);
    return v;
}

}  // namespace
