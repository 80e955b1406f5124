// sgmcmc_svgd.hip -- Stein variational gradient descent step (pysgmcmc/samplers/svgd.py:118-181).
//
// n particles of `dim` parameters each live as rows of one [n x ld] matrix X (the theta row of the
// sampler's arena); G holds d cost / d X, H the running mean of squared updates ("historical_grad").
// One step is four launches:
//   S1 svgd_sqdist_kernel     partial sums of sum_c (X[i][c] - X[j][c])^2 over column ranges: a tile of
//                             columns is staged TRANSPOSED in LDS ([c][i]) and every lane owns a 4x4
//                             block of (i, j) pairs in registers (2 ds_read_b128 per 32 VALU ops); only
//                             pair blocks on or above the diagonal are computed. Reads X once.
//   S2 svgd_reduce_kernel     adds the partials in fixed order (bit-reproducible, no atomics), applies
//                             tf.norm's sqrt and the `** 2` of svgd.py:166, writes the symmetric D.
//   S3 svgd_bandwidth_kernel  one workgroup: median of all n*n entries of D by radix select
//                             (tensor_utils.py:197-209), h = sqrt(0.5 median / log(n + 1)),
//                             K = exp(-D / h^2 / 2), row sums (svgd.py:169-174).
//   S4 svgd_update_kernel     streams X, G, H once: one wave per 64-column tile, the tile staged in LDS,
//                             A = K G and B = K X accumulated with v_fma (K read through scalar loads,
//                             16 output rows per pass), then the element-wise tail of svgd.py:124-143
//                             with one rounding per reference op. R{X,G,H} W{X,H} = 20 B per element.
// The contractions are n x n x dim with n <= 128: fp32 MFMA and fp32 VALU have the SAME peak on gfx950
// (157 TFLOP/s) and S4 is HBM-bound up to n ~ 64 (2n FMA per 20 bytes), so the VALU form with scalar
// K operands is used; there is no precision mode to trade (the reference computes in fp32/fp64).
//
// Sums over columns / particles have no reference rounding order (tf.reduce_sum / tf.matmul), so
// parity with the oracle is to accumulated-rounding tolerance; everything after them is op-for-op.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "sgmcmc_hip.h"

#pragma clang fp contract(off)

#include "sgmcmc_host.hpp"

using namespace sgmcmc_host;

namespace {

constexpr int SVGD_MAX_PARTICLES = 128;
constexpr int SVGD_THREADS = 256;
constexpr int SVGD_MAX_PARTS = 512;          // column-range workgroups of S1
constexpr int SVGD_TILE_BYTES = 32 * 1024;   // LDS tile of S1
constexpr int SVGD_HDR = 16;                 // workspace header elements: median, h, h^2
constexpr int SVGD_ITILE = 16;               // output rows per pass of S4
constexpr int SVGD_UCOLS = 64;               // columns per tile of S4 (one wave)

__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float sqrt_t(float x) { return __builtin_sqrtf(x); }
__device__ __forceinline__ double sqrt_t(double x) { return __builtin_sqrt(x); }
__device__ __forceinline__ float exp_t(float x) { return expf(x); }
__device__ __forceinline__ double exp_t(double x) { return exp(x); }
__device__ __forceinline__ float log_t(float x) { return logf(x); }
__device__ __forceinline__ double log_t(double x) { return log(x); }

template <typename T> struct alignas(4 * sizeof(T)) Vec4 { T v[4]; };

struct SvgdGeom {
    int n;       // particles
    int np;      // n rounded up to a multiple of 4
    int nb;      // np / 4 blocks of 4 particles
    int npb;     // nb (nb + 1) / 2 pair blocks on or above the diagonal
    int np16;    // n rounded up to a multiple of 16: row pitch of K
};

__host__ __device__ inline SvgdGeom svgd_geom(int n) {
    SvgdGeom g;
    g.n = n;
    g.np = (n + 3) & ~3;
    g.nb = g.np / 4;
    g.npb = g.nb * (g.nb + 1) / 2;
    g.np16 = (n + 15) & ~15;
    return g;
}

// workspace layout, in elements of T
struct SvgdWs {
    size_t hdr, D, K, ksum, parts, total;
};

inline SvgdWs svgd_ws(int n) {
    SvgdGeom g = svgd_geom(n);
    SvgdWs w;
    size_t off = 0;
    w.hdr = off; off += SVGD_HDR;
    w.D = off; off += (size_t)n * n; off = (off + 15) & ~(size_t)15;
    w.K = off; off += (size_t)g.np16 * g.np16;
    w.ksum = off; off += g.np16;
    w.parts = off; off += (size_t)SVGD_MAX_PARTS * g.npb * 16;
    w.total = off;
    return w;
}

__device__ __forceinline__ void decode_pair_block(int pb, int nb, int &bi, int &bj) {
    bi = 0;
    int row = nb;
    while (pb >= row) { pb -= row; ++bi; --row; }
    bj = bi + pb;
}

// ---------------------------------------------------------------------------------------------
// S1: partial squared distances over this workgroup's column tiles
// ---------------------------------------------------------------------------------------------
template <typename T, int MAXPB>
__global__ __launch_bounds__(SVGD_THREADS) void svgd_sqdist_kernel(const T *__restrict__ X, size_t dim, size_t ld,
                                                                    int n, int tc_log2, T *__restrict__ parts) {
    extern __shared__ __align__(32) unsigned char svgd_lds_raw[];
    T *xs = reinterpret_cast<T *>(svgd_lds_raw);
    const SvgdGeom g = svgd_geom(n);
    const int NP = g.np + 4;                       // padded LDS pitch, keeps Vec4 alignment
    const int tc = 1 << tc_log2;
    const int t = threadIdx.x;
    const int S = (g.npb >= SVGD_THREADS) ? 1 : (SVGD_THREADS / g.npb);   // column slices per pair block

    int bi[MAXPB], bj[MAXPB];
    bool have[MAXPB];
    int slice = 0;
#pragma unroll
    for (int k = 0; k < MAXPB; ++k) {
        int pb;
        if (S == 1) {
            pb = t + k * SVGD_THREADS;
            have[k] = pb < g.npb;
        } else {
            pb = t % g.npb;
            slice = t / g.npb;
            have[k] = (k == 0) && (t < S * g.npb);
        }
        bi[k] = bj[k] = 0;
        if (have[k]) decode_pair_block(pb, g.nb, bi[k], bj[k]);
    }
    T acc[MAXPB][16];
#pragma unroll
    for (int k = 0; k < MAXPB; ++k)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[k][e] = (T)0;

    for (int idx = t; idx < tc * NP; idx += SVGD_THREADS) xs[idx] = (T)0;   // padding rows stay zero
    __syncthreads();

    const size_t n_tiles = (dim + (size_t)tc - 1) >> tc_log2;
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t c0 = tile << tc_log2;
        for (int idx = t; idx < (n << tc_log2); idx += SVGD_THREADS) {
            const int i = idx >> tc_log2, c = idx & (tc - 1);
            const size_t col = c0 + (size_t)c;
            xs[c * NP + i] = (col < dim) ? X[(size_t)i * ld + col] : (T)0;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < MAXPB; ++k) {
            if (!have[k]) continue;
            const int oi = 4 * bi[k], oj = 4 * bj[k];
            for (int c = slice; c < tc; c += S) {
                const Vec4<T> a = *reinterpret_cast<const Vec4<T> *>(xs + c * NP + oi);
                const Vec4<T> b = *reinterpret_cast<const Vec4<T> *>(xs + c * NP + oj);
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const T d = a.v[p] - b.v[q];
                        acc[k][p * 4 + q] = fma_t(d, d, acc[k][p * 4 + q]);
                    }
            }
        }
        __syncthreads();
    }

    T *out = parts + (size_t)blockIdx.x * g.npb * 16;
    if (S == 1) {
#pragma unroll
        for (int k = 0; k < MAXPB; ++k) {
            const int pb = t + k * SVGD_THREADS;
            if (pb < g.npb)
#pragma unroll
                for (int e = 0; e < 16; ++e) out[(size_t)pb * 16 + e] = acc[k][e];
        }
    } else {
        // add the column slices in slice order through LDS (S * npb * 16 <= 4096 elements)
        T *red = xs;
        if (t < S * g.npb)
#pragma unroll
            for (int e = 0; e < 16; ++e) red[((size_t)slice * g.npb + (t % g.npb)) * 16 + e] = acc[0][e];
        __syncthreads();
        for (int idx = t; idx < g.npb * 16; idx += SVGD_THREADS) {
            T s = (T)0;
            for (int sl = 0; sl < S; ++sl) s += red[(size_t)sl * g.npb * 16 + idx];
            out[idx] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// S2: D[i][j] = (sqrt(sum_parts))^2, symmetric, zero diagonal
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(SVGD_THREADS) void svgd_reduce_kernel(const T *__restrict__ parts, int n_parts, int n,
                                                                    T *__restrict__ D) {
    const SvgdGeom g = svgd_geom(n);
    const int idx = blockIdx.x * SVGD_THREADS + threadIdx.x;
    if (idx >= g.npb * 16) return;
    T s = (T)0;
    for (int p = 0; p < n_parts; ++p) s += parts[(size_t)p * g.npb * 16 + idx];
    int bi, bj;
    decode_pair_block(idx >> 4, g.nb, bi, bj);
    const int i = 4 * bi + ((idx & 15) >> 2), j = 4 * bj + (idx & 3);
    if (i >= n || j >= n) return;
    const T dist = sqrt_t(s);                      // tf.norm, tensor_utils.py:399
    const T sq = dist * dist;                      // `** 2`, svgd.py:166
    D[(size_t)i * n + j] = sq;
    D[(size_t)j * n + i] = sq;
}

// ---------------------------------------------------------------------------------------------
// S3: median bandwidth, kernel matrix, row sums -- one workgroup
// ---------------------------------------------------------------------------------------------
template <typename T> struct KeyOf;
template <> struct KeyOf<float> {
    using type = uint32_t;
    static __device__ __forceinline__ uint32_t bits(float x) { return __float_as_uint(x); }
    static __device__ __forceinline__ float value(uint32_t k) { return __uint_as_float(k); }
};
template <> struct KeyOf<double> {
    using type = uint64_t;
    static __device__ __forceinline__ uint64_t bits(double x) { return (uint64_t)__double_as_longlong(x); }
    static __device__ __forceinline__ double value(uint64_t k) { return __longlong_as_double((long long)k); }
};

// value of rank `rank` (0-based, ascending) among the N non-negative values v[]: most significant
// byte first, 256-bin LDS histogram per pass (non-negative IEEE values order like their bit patterns)
template <typename T>
__device__ T radix_select(const T *__restrict__ v, int N, int rank, unsigned int *hist, unsigned long long *bcast) {
    using Key = typename KeyOf<T>::type;
    Key prefix = 0, mask = 0;
    int remaining = rank;
    for (int shift = (int)sizeof(Key) * 8 - 8; shift >= 0; shift -= 8) {
        for (int b = threadIdx.x; b < 256; b += blockDim.x) hist[b] = 0u;
        __syncthreads();
        for (int idx = threadIdx.x; idx < N; idx += blockDim.x) {
            const Key k = KeyOf<T>::bits(v[idx]);
            if ((k & mask) == prefix) atomicAdd(&hist[(unsigned)((k >> shift) & (Key)255)], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int cum = 0, digit = 255;
            for (int b = 0; b < 256; ++b) {
                const int h = (int)hist[b];
                if (cum + h > remaining) { digit = b; break; }
                cum += h;
            }
            bcast[0] = (unsigned long long)digit;
            bcast[1] = (unsigned long long)cum;
        }
        __syncthreads();
        const Key digit = (Key)bcast[0];
        remaining -= (int)bcast[1];
        prefix |= digit << shift;
        mask |= (Key)255 << shift;
        __syncthreads();
    }
    return KeyOf<T>::value(prefix);
}

template <typename T>
__global__ __launch_bounds__(1024) void svgd_bandwidth_kernel(const T *__restrict__ D, int n, T *__restrict__ hdr,
                                                               T *__restrict__ K, T *__restrict__ ksum) {
    __shared__ unsigned int hist[256];
    __shared__ unsigned long long bcast[2];
    const SvgdGeom g = svgd_geom(n);
    const int N = n * n;
    const int mid = N / 2;
    T med;
    if (N & 1) {
        med = radix_select(D, N, mid, hist, bcast);                     // tensor_utils.py:205-206
    } else {
        const T lo = radix_select(D, N, mid - 1, hist, bcast);
        const T hi = radix_select(D, N, mid, hist, bcast);
        med = (lo + hi) / (T)2;                                         // tensor_utils.py:208
    }
    const T h = sqrt_t((T)0.5 * med / log_t((T)n + (T)1));              // svgd.py:169-171
    const T h2 = h * h;
    if (threadIdx.x == 0) { hdr[0] = med; hdr[1] = h; hdr[2] = h2; }
    for (int idx = threadIdx.x; idx < g.np16 * g.np16; idx += blockDim.x) {
        const int i = idx / g.np16, j = idx % g.np16;
        T k = (T)0;
        if (i < n && j < n) k = exp_t(-D[(size_t)i * n + j] / h2 / (T)2);   // svgd.py:173
        K[idx] = k;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < g.np16; i += blockDim.x) {
        T s = (T)0;
        if (i < n)
            for (int j = 0; j < n; ++j) s += K[(size_t)i * g.np16 + j];    // svgd.py:174
        ksum[i] = s;
    }
}

// ---------------------------------------------------------------------------------------------
// S4: A = K G, B = K X per 64-column tile, then the element-wise update (or the kernel gradients)
// ---------------------------------------------------------------------------------------------
template <typename T, bool UPDATE>
__global__ __launch_bounds__(SVGD_UCOLS) void svgd_update_kernel(T *__restrict__ X, const T *__restrict__ G,
                                                                  T *__restrict__ H, T *__restrict__ kgrad_out,
                                                                  size_t dim, size_t ld, size_t out_ld, int n,
                                                                  const T *__restrict__ hdr, const T *__restrict__ K,
                                                                  const T *__restrict__ ksum, T eps, T alpha,
                                                                  T one_minus_alpha, T fudge, T sign) {
    extern __shared__ __align__(32) unsigned char svgd_lds_raw[];
    T *xs = reinterpret_cast<T *>(svgd_lds_raw);          // [n][64]
    T *gs = xs + (size_t)n * SVGD_UCOLS;                  // [n][64]
    const SvgdGeom g = svgd_geom(n);
    const int lane = threadIdx.x;
    const T h2 = hdr[2];
    const T n_t = (T)n;
    const size_t n_tiles = (dim + SVGD_UCOLS - 1) / SVGD_UCOLS;
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t c = tile * SVGD_UCOLS + (size_t)lane;
        const bool valid = c < dim;
        for (int j = 0; j < n; ++j) {
            xs[j * SVGD_UCOLS + lane] = valid ? X[(size_t)j * ld + c] : (T)0;
            if (UPDATE) gs[j * SVGD_UCOLS + lane] = valid ? G[(size_t)j * ld + c] : (T)0;
        }
        __syncthreads();
        for (int i0 = 0; i0 < n; i0 += SVGD_ITILE) {
            T accg[SVGD_ITILE], accx[SVGD_ITILE];
#pragma unroll
            for (int a = 0; a < SVGD_ITILE; ++a) { accg[a] = (T)0; accx[a] = (T)0; }
            for (int j = 0; j < n; ++j) {
                const T xj = xs[j * SVGD_UCOLS + lane];
                const T gj = UPDATE ? gs[j * SVGD_UCOLS + lane] : (T)0;
                const T *krow = K + (size_t)j * g.np16 + i0;       // K is symmetric: row j, columns i0..i0+15
#pragma unroll
                for (int a = 0; a < SVGD_ITILE; ++a) {
                    const T k = krow[a];
                    accx[a] = fma_t(k, xj, accx[a]);
                    if (UPDATE) accg[a] = fma_t(k, gj, accg[a]);
                }
            }
#pragma unroll
            for (int a = 0; a < SVGD_ITILE; ++a) {
                const int i = i0 + a;
                if (i < n && valid) {
                    const T x = xs[i * SVGD_UCOLS + lane];
                    const T kg = (-accx[a] + x * ksum[i]) / h2;              // svgd.py:176-181
                    if (UPDATE) {
                        const size_t at = (size_t)i * ld + c;
                        const T gt = (accg[a] + sign * kg) / n_t;            // svgd.py:124-127
                        const T hnew = alpha * H[at] + one_minus_alpha * (gt * gt);   // svgd.py:129-132
                        const T adj = gt / (fudge + sqrt_t(hnew));           // svgd.py:134-137
                        H[at] = hnew;
                        X[at] = x - eps * adj;                               // svgd.py:139-143
                    } else {
                        kgrad_out[(size_t)i * out_ld + c] = kg;
                    }
                }
            }
        }
        __syncthreads();
    }
}

template <typename T>
__global__ void svgd_copy_kernel(const T *__restrict__ K, int n, int np16, T *__restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n * n) out[idx] = K[(size_t)(idx / n) * np16 + (idx % n)];
}

int svgd_check(const void *X, size_t n, size_t dim, size_t ld, const void *ws, const char *who) {
    if (!X || !ws) return fail(SGMCMC_EINVAL, "%s: null pointer", who);
    if (n < 1 || n > (size_t)SVGD_MAX_PARTICLES)
        return fail(SGMCMC_EINVAL, "%s: n_particles = %zu outside [1, %d]", who, n, SVGD_MAX_PARTICLES);
    if (dim < 1 || ld < dim) return fail(SGMCMC_EINVAL, "%s: need 1 <= dim <= ld", who);
    return 0;
}

template <typename T>
int svgd_kernel_matrix_impl(const T *X, size_t n, size_t dim, size_t ld, T *ws, hipStream_t st) {
    const SvgdGeom g = svgd_geom((int)n);
    const SvgdWs w = svgd_ws((int)n);
    const int NP = g.np + 4;
    int tc_log2 = 4;
    while (tc_log2 < 10 && (size_t)(2 << tc_log2) * NP * sizeof(T) <= (size_t)SVGD_TILE_BYTES) ++tc_log2;
    const size_t tc = (size_t)1 << tc_log2;
    size_t lds_elems = tc * NP;
    if (lds_elems < 4096) lds_elems = 4096;                 // slice-reduction scratch
    const size_t n_tiles = (dim + tc - 1) / tc;
    const int n_parts = (int)(n_tiles < (size_t)SVGD_MAX_PARTS ? n_tiles : (size_t)SVGD_MAX_PARTS);
    T *parts = ws + w.parts;
    const size_t lds_bytes = lds_elems * sizeof(T);
    if (g.npb <= SVGD_THREADS)
        hipLaunchKernelGGL((svgd_sqdist_kernel<T, 1>), dim3(n_parts), dim3(SVGD_THREADS), lds_bytes, st, X, dim, ld,
                           (int)n, tc_log2, parts);
    else if (g.npb <= 2 * SVGD_THREADS)
        hipLaunchKernelGGL((svgd_sqdist_kernel<T, 2>), dim3(n_parts), dim3(SVGD_THREADS), lds_bytes, st, X, dim, ld,
                           (int)n, tc_log2, parts);
    else
        hipLaunchKernelGGL((svgd_sqdist_kernel<T, 3>), dim3(n_parts), dim3(SVGD_THREADS), lds_bytes, st, X, dim, ld,
                           (int)n, tc_log2, parts);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "launch svgd_sqdist_kernel");
    const int red_blocks = (g.npb * 16 + SVGD_THREADS - 1) / SVGD_THREADS;
    hipLaunchKernelGGL((svgd_reduce_kernel<T>), dim3(red_blocks), dim3(SVGD_THREADS), 0, st, parts, n_parts, (int)n,
                       ws + w.D);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "launch svgd_reduce_kernel");
    hipLaunchKernelGGL((svgd_bandwidth_kernel<T>), dim3(1), dim3(1024), 0, st, ws + w.D, (int)n, ws + w.hdr, ws + w.K,
                       ws + w.ksum);
    e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "launch svgd_bandwidth_kernel");
}

template <typename T, bool UPDATE>
int svgd_apply_impl(T *X, const T *G, T *H, T *kgrad_out, size_t out_ld, size_t n, size_t dim, size_t ld, T eps,
                    double alpha, T fudge, T sign, const T *ws, hipStream_t st) {
    const SvgdWs w = svgd_ws((int)n);
    const size_t lds_bytes = (size_t)2 * n * SVGD_UCOLS * sizeof(T);
    if (lds_bytes > 64 * 1024) {
        hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void *>(&svgd_update_kernel<T, UPDATE>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e0 != hipSuccess) return hip_fail(e0, "hipFuncSetAttribute(svgd_update_kernel)");
    }
    const size_t n_tiles = (dim + SVGD_UCOLS - 1) / SVGD_UCOLS;
    const size_t cap = (size_t)1 << 20;
    const unsigned grid = (unsigned)(n_tiles < cap ? n_tiles : cap);
    hipLaunchKernelGGL((svgd_update_kernel<T, UPDATE>), dim3(grid), dim3(SVGD_UCOLS), lds_bytes, st, X, G, H, kgrad_out,
                       dim, ld, out_ld, (int)n, ws + w.hdr, ws + w.K, ws + w.ksum, eps, (T)alpha, (T)(1.0 - alpha), fudge,
                       sign);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "launch svgd_update_kernel");
}

template <typename T>
int svgd_step_impl(T *X, const T *G, T *H, size_t n, size_t dim, size_t ld, T eps, double alpha, T fudge,
                   int repulsion_sign, void *ws, sgmcmc_stream_t stream) {
    int rc = svgd_check(X, n, dim, ld, ws, "sgmcmc_svgd_step");
    if (rc) return rc;
    if (!G || !H) return fail(SGMCMC_EINVAL, "sgmcmc_svgd_step: null pointer");
    if (repulsion_sign != 1 && repulsion_sign != -1)
        return fail(SGMCMC_EINVAL, "sgmcmc_svgd_step: repulsion_sign must be +1 (reference) or -1");
    hipStream_t st = static_cast<hipStream_t>(stream);
    rc = svgd_kernel_matrix_impl<T>(X, n, dim, ld, static_cast<T *>(ws), st);
    if (rc) return rc;
    return svgd_apply_impl<T, true>(X, G, H, nullptr, 0, n, dim, ld, eps, alpha, fudge, (T)repulsion_sign,
                                    static_cast<const T *>(ws), st);
}

template <typename T>
int svgd_kernel_impl(const T *X, size_t n, size_t dim, size_t ld, void *ws, T *kernel_out, T *kgrad_out,
                     size_t kgrad_ld, T *bandwidth_out, sgmcmc_stream_t stream) {
    int rc = svgd_check(X, n, dim, ld, ws, "sgmcmc_svgd_kernel");
    if (rc) return rc;
    if (kgrad_out && kgrad_ld < dim) return fail(SGMCMC_EINVAL, "sgmcmc_svgd_kernel: kgrad_ld < dim");
    hipStream_t st = static_cast<hipStream_t>(stream);
    T *w = static_cast<T *>(ws);
    rc = svgd_kernel_matrix_impl<T>(X, n, dim, ld, w, st);
    if (rc) return rc;
    const SvgdGeom g = svgd_geom((int)n);
    const SvgdWs lay = svgd_ws((int)n);
    if (kernel_out) {
        const int total = (int)(n * n);
        hipLaunchKernelGGL((svgd_copy_kernel<T>), dim3((total + 255) / 256), dim3(256), 0, st, w + lay.K, (int)n, g.np16,
                           kernel_out);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "launch svgd_copy_kernel");
    }
    if (bandwidth_out) {
        hipError_t e = hipMemcpyAsync(bandwidth_out, w + lay.hdr, 3 * sizeof(T), hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(bandwidth)");
    }
    if (kgrad_out)
        return svgd_apply_impl<T, false>(const_cast<T *>(X), nullptr, nullptr, kgrad_out, kgrad_ld, n, dim, ld, (T)0, 0.0,
                                         (T)0, (T)1, w, st);
    return 0;
}

}  // namespace

extern "C" {

size_t sgmcmc_svgd_workspace_bytes(size_t n_particles, size_t elem_bytes) {
    if (n_particles < 1 || n_particles > (size_t)SVGD_MAX_PARTICLES) return 0;
    return svgd_ws((int)n_particles).total * elem_bytes;
}

int sgmcmc_svgd_max_particles(void) { return SVGD_MAX_PARTICLES; }

int sgmcmc_svgd_step_f32(float *particles, const float *grad, float *hist_grad, size_t n_particles, size_t dim, size_t ld,
                         float eps, double alpha, float fudge_factor, int repulsion_sign, void *workspace,
                         sgmcmc_stream_t stream) {
    return svgd_step_impl<float>(particles, grad, hist_grad, n_particles, dim, ld, eps, alpha, fudge_factor,
                                 repulsion_sign, workspace, stream);
}

int sgmcmc_svgd_step_f64(double *particles, const double *grad, double *hist_grad, size_t n_particles, size_t dim,
                         size_t ld, double eps, double alpha, double fudge_factor, int repulsion_sign, void *workspace,
                         sgmcmc_stream_t stream) {
    return svgd_step_impl<double>(particles, grad, hist_grad, n_particles, dim, ld, eps, alpha, fudge_factor,
                                  repulsion_sign, workspace, stream);
}

int sgmcmc_svgd_kernel_f32(const float *particles, size_t n_particles, size_t dim, size_t ld, void *workspace,
                           float *kernel_out, float *kernel_grad_out, size_t kernel_grad_ld, float *bandwidth_out,
                           sgmcmc_stream_t stream) {
    return svgd_kernel_impl<float>(particles, n_particles, dim, ld, workspace, kernel_out, kernel_grad_out,
                                   kernel_grad_ld, bandwidth_out, stream);
}

int sgmcmc_svgd_kernel_f64(const double *particles, size_t n_particles, size_t dim, size_t ld, void *workspace,
                           double *kernel_out, double *kernel_grad_out, size_t kernel_grad_ld, double *bandwidth_out,
                           sgmcmc_stream_t stream) {
    return svgd_kernel_impl<double>(particles, n_particles, dim, ld, workspace, kernel_out, kernel_grad_out,
                                    kernel_grad_ld, bandwidth_out, stream);
}

}  // extern "C"
