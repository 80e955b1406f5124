// sgmcmc_svgd.hip -- Stein variational gradient descent step (pysgmcmc/samplers/svgd.py:118-181).
//
// n <= 128 particles of `dim` parameters each live as rows of one [n x ld] matrix X (the theta row of the
// sampler's arena, row pitch padded to 64 elements); G holds d cost / d X, H the running mean of squared
// updates ("historical_grad"). One step is four launches:
//   S1  pairwise squared distances, partial sums per column range. Reads X once (4 B / element).
//         n <= 12 (any dtype): svgd_sqdist_small_kernel -- registers only, 8/16-byte loads;
//         13..32 f32, 13..64 f64: svgd_gram_mfma16_kernel -- Gram matrix of column-centred 64-column tiles on the
//                                                          matrix cores (v_mfma_{f32,f64}_16x16x4);
//         33..128, f32:        svgd_gram_mfma_kernel    -- the same with v_mfma_f32_32x32x2_f32, 128-column tiles;
//         65..128, f64:        svgd_sqdist_kernel       -- difference form, transposed LDS tile, 4x4 pair
//                                                          blocks per lane.
//   S2  svgd_reduce_*_kernel   adds the partials in a fixed order (bit-reproducible, no atomics).
//   S3  svgd_bandwidth_kernel  one workgroup: (D from the Gram matrix,) median of all n*n entries of D by
//                              radix select (tensor_utils.py:197-209), h = sqrt(0.5 median / log(n + 1)),
//                              K = exp(-D / h^2 / 2), row sums (svgd.py:169-174).
//   S4  A = K G, B = K X, then the element-wise tail of svgd.py:124-143 with one rounding per reference
//       op. R{X,G,H} W{X,H} = 20 B / element.
//         n <= 8:                    svgd_update_small_kernel    -- registers, K rows as scalar operands;
//         9..64, f32 and f64:        svgd_update_mfma16_kernel   -- matrix cores (v_mfma_*_16x16x4), cooperative
//                                                                   64-column tiles, 16-byte row-major global accesses;
//         65..128, f32:              svgd_update_mfma_kernel     -- the same with v_mfma_f32_32x32x2_f32;
//         65..128, f64:              svgd_update_mfma_f64_big_kernel -- two passes of four 16-particle blocks
//       (svgd_update_reg_kernel / svgd_update_kernel also serve sgmcmc_svgd_kernel_*'s kernel-gradient output).
// fp32 MFMA and packed fp32 VALU have the same peak on gfx950 (157 TFLOP/s; 155 measured for
// v_mfma_f32_32x32x2_f32, measured in round 1) and the f32 MFMA is exact f32, so there is no precision
// to trade; the matrix-core forms win because K lives in LDS/registers instead of stalling on scalar loads and
// because the VALU stays free for the tail (3 IEEE divisions + sqrt per element).
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
constexpr int SVGD_MAX_PARTS = 1024;         // column-range workgroups of S1
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
    const size_t per_part = (size_t)g.npb * 16 > 120 ? (size_t)g.npb * 16 : 120;   // S1 or S1s layout
    size_t parts_elems = (size_t)SVGD_MAX_PARTS * per_part;
    const size_t gram_elems = n < 13 ? 0 : n <= 32 ? (size_t)2048 * 1024 : n <= 64 ? (size_t)1024 * 3072 : (size_t)512 * 10240;
    if (parts_elems < gram_elems) parts_elems = gram_elems;           // partial Gram blocks of the matrix-core S1
    w.parts = off; off += parts_elems;
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
    const bool vec4 = (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & (4 * sizeof(T) - 1)) == 0);

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
        if (vec4 && c0 + (size_t)tc <= dim) {
            // interior tile, rows 4-element aligned: one 4-wide load per lane, 4 in flight
            const int q_log2 = tc_log2 - 2, total = n << q_log2;
            for (int base = 0; base < total; base += SVGD_THREADS * 4) {
                Vec4<T> v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = base + u * SVGD_THREADS + t;
                    if (idx < total) {
                        const int i = idx >> q_log2, cq = idx & ((1 << q_log2) - 1);
                        v[u] = *reinterpret_cast<const Vec4<T> *>(X + (size_t)i * ld + c0 + 4 * (size_t)cq);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = base + u * SVGD_THREADS + t;
                    if (idx < total) {
                        const int i = idx >> q_log2, cq = idx & ((1 << q_log2) - 1);
#pragma unroll
                        for (int e = 0; e < 4; ++e) xs[(4 * cq + e) * NP + i] = v[u].v[e];
                    }
                }
            }
        } else {
            const int total = n << tc_log2;
            for (int base = 0; base < total; base += SVGD_THREADS * 8) {
                T v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = base + u * SVGD_THREADS + t;
                    v[u] = (T)0;
                    if (idx < total) {
                        const size_t col = c0 + (size_t)(idx & (tc - 1));
                        if (col < dim) v[u] = X[(size_t)(idx >> tc_log2) * ld + col];
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = base + u * SVGD_THREADS + t;
                    if (idx < total) xs[(idx & (tc - 1)) * NP + (idx >> tc_log2)] = v[u];
                }
            }
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
// 64 outputs x 16 part-slices per workgroup: every lane adds its slice of the partials (independent loads),
// the slices are then added in slice order -- a fixed summation tree, bit-reproducible.
constexpr int SVGD_RED_SLICES = 16;

template <typename T>
__device__ __forceinline__ bool reduce_parts(const T *__restrict__ parts, int n_parts, int n_out, T &total, int &idx) {
    __shared__ T red[SVGD_RED_SLICES][64];
    const int l = threadIdx.x & 63, slice = threadIdx.x >> 6;
    idx = blockIdx.x * 64 + l;
    T s = (T)0;
    if (idx < n_out) {
#pragma unroll 4
        for (int q = slice; q < n_parts; q += SVGD_RED_SLICES) s += parts[(size_t)q * n_out + idx];
    }
    red[slice][l] = s;
    __syncthreads();
    if (slice != 0 || idx >= n_out) return false;
    total = red[0][l];
#pragma unroll
    for (int k = 1; k < SVGD_RED_SLICES; ++k) total += red[k][l];
    return true;
}

template <typename T>
__global__ __launch_bounds__(64 * SVGD_RED_SLICES) void svgd_reduce_kernel(const T *__restrict__ parts, int n_parts,
                                                                            int n, T *__restrict__ D) {
    const SvgdGeom g = svgd_geom(n);
    T s;
    int idx;
    if (!reduce_parts(parts, n_parts, g.npb * 16, s, idx)) return;
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
// byte first, 256-bin LDS histogram per pass (non-negative IEEE values order like their bit patterns);
// the digit is located by wave 0 with a shuffle scan over 4 bins per lane
template <typename T>
__device__ typename KeyOf<T>::type radix_select(const T *__restrict__ v, int N, int rank, unsigned int *hist,
                                                unsigned long long *bcast) {
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
        if (threadIdx.x < 64) {
            const int l = threadIdx.x;
            const int h0 = (int)hist[4 * l], h1 = (int)hist[4 * l + 1], h2 = (int)hist[4 * l + 2], h3 = (int)hist[4 * l + 3];
            const int mine = h0 + h1 + h2 + h3;
            int incl = mine;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int up = __shfl_up(incl, off, 64);
                if (l >= off) incl += up;
            }
            int cum = incl - mine;                       // entries in bins below 4l
            if (cum <= remaining && remaining < incl) {  // exactly one lane
                int digit = 4 * l;
                if (remaining >= cum + h0) { cum += h0; ++digit;
                    if (remaining >= cum + h1) { cum += h1; ++digit;
                        if (remaining >= cum + h2) { cum += h2; ++digit; } } }
                bcast[0] = (unsigned long long)digit;
                bcast[1] = (unsigned long long)cum;
            }
        }
        __syncthreads();
        const Key digit = (Key)bcast[0];
        remaining -= (int)bcast[1];
        prefix |= digit << shift;
        mask |= (Key)255 << shift;
    }
    return prefix;
}

template <typename T>
__global__ __launch_bounds__(512) void svgd_bandwidth_kernel(T *__restrict__ D, int n, T *__restrict__ hdr,
                                                              T *__restrict__ K, T *__restrict__ ksum, int from_gram) {
    using Key = typename KeyOf<T>::type;
    __shared__ unsigned int hist[256];
    __shared__ unsigned long long bcast[2];
    __shared__ unsigned long long next_key;
    __shared__ unsigned int count_le;
    const SvgdGeom g = svgd_geom(n);
    const int N = n * n;
    const int mid = N / 2;
    if (from_gram) {
        // the matrix-core path left the Gram matrix of the (column-centred) particles in K's storage:
        // |x_i - x_j|^2 = G_ii + G_jj - 2 G_ij, then tf.norm's sqrt and the `** 2` as on the other paths
        for (int idx = threadIdx.x; idx < N; idx += blockDim.x) {
            const int i = idx / n, j = idx % n;
            T sq = (T)0;
            if (i != j) {
                const int lo_i = i < j ? i : j, hi_j = i < j ? j : i;     // one rounding sequence for (i,j) and (j,i)
                T s = (K[(size_t)lo_i * g.np16 + lo_i] + K[(size_t)hi_j * g.np16 + hi_j]) -
                      (T)2 * K[(size_t)lo_i * g.np16 + hi_j];
                s = s > (T)0 ? s : (T)0;
                const T dist = sqrt_t(s);
                sq = dist * dist;
            }
            D[idx] = sq;
        }
        __syncthreads();
    }
    T med;
    if (N & 1) {
        med = KeyOf<T>::value(radix_select(D, N, mid, hist, bcast));    // tensor_utils.py:205-206
    } else {
        // the two middle values: rank mid-1 by radix select, rank mid = the same value if it repeats,
        // else the smallest larger entry
        const Key lo = radix_select(D, N, mid - 1, hist, bcast);
        if (threadIdx.x == 0) { next_key = ~0ull; count_le = 0u; }
        __syncthreads();
        unsigned int c_le = 0;
        unsigned long long mn = ~0ull;
        for (int idx = threadIdx.x; idx < N; idx += blockDim.x) {
            const Key k = KeyOf<T>::bits(D[idx]);
            if (k <= lo) ++c_le;
            else if ((unsigned long long)k < mn) mn = (unsigned long long)k;
        }
        atomicAdd(&count_le, c_le);
        atomicMin(&next_key, mn);
        __syncthreads();
        const Key hi = ((int)count_le > mid) ? lo : (Key)next_key;
        med = (KeyOf<T>::value(lo) + KeyOf<T>::value(hi)) / (T)2;       // tensor_utils.py:208
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

// ---------------------------------------------------------------------------------------------
// n <= 16: the memory-bound regime (a few particles of a large model). No LDS tiles: a lane owns CPL
// adjacent columns (one 8/16-byte access per row), keeps every particle's values in registers, and
//   S1s accumulates all n (n - 1) / 2 squared differences per lane, reduced once per workgroup;
//   S4s forms A = K G, B = K X for its columns with K rows as scalar operands, then the element-wise tail.
// ---------------------------------------------------------------------------------------------
template <typename T, int NPAD, int CPL>
__device__ __forceinline__ void load_rows(const T *__restrict__ A, size_t ld, size_t c, size_t dim, int n, bool vec,
                                          T (&out)[NPAD][CPL]) {
#pragma unroll
    for (int j = 0; j < NPAD; ++j) {
#pragma unroll
        for (int k = 0; k < CPL; ++k) out[j][k] = (T)0;
        if (j < n) {
            if (vec) {
                // the address is made opaque so the compiler cannot merge this access with the element-wise
                // path below and split it into a 1-wide plus a 3-wide access
                const T *pa = A + (size_t)j * ld + c;
                asm volatile("" : "+v"(pa));
                typedef T VecT __attribute__((ext_vector_type(CPL)));
                typedef const __attribute__((address_space(1))) VecT *GlobalVecPtr;   // global_load, not flat
                const VecT v = *(GlobalVecPtr)(reinterpret_cast<const VecT *>(pa));
#pragma unroll
                for (int k = 0; k < CPL; ++k) out[j][k] = v[k];
            } else {
#pragma unroll
                for (int k = 0; k < CPL; ++k)
                    if (c + k < dim) out[j][k] = A[(size_t)j * ld + c + k];
            }
        }
    }
}

template <typename T, int NPAD, int CPL>
__device__ __forceinline__ void store_rows(T *__restrict__ A, size_t ld, size_t c, size_t dim, int n, bool vec,
                                           const T (&in)[NPAD][CPL]) {
#pragma unroll
    for (int j = 0; j < NPAD; ++j) {
        if (j < n) {
            if (vec) {
                typedef T VecT __attribute__((ext_vector_type(CPL)));
                typedef __attribute__((address_space(1))) VecT *GlobalVecPtr;
                VecT v;
#pragma unroll
                for (int k = 0; k < CPL; ++k) v[k] = in[j][k];
                T *pa = A + (size_t)j * ld + c;
                asm volatile("" : "+v"(pa));
                *(GlobalVecPtr)(reinterpret_cast<VecT *>(pa)) = v;
            } else {
#pragma unroll
                for (int k = 0; k < CPL; ++k)
                    if (c + k < dim) A[(size_t)j * ld + c + k] = in[j][k];
            }
        }
    }
}

template <typename T, int NPAD, int CPL>
__global__ __launch_bounds__(SVGD_THREADS) void svgd_sqdist_small_kernel(const T *__restrict__ X, size_t dim, size_t ld,
                                                                          int n, T *__restrict__ parts) {
    constexpr int NPAIR = NPAD * (NPAD - 1) / 2;
    __shared__ T red[SVGD_THREADS / 64][NPAIR];
    const bool aligned = (ld % CPL == 0) && ((reinterpret_cast<uintptr_t>(X) & (CPL * sizeof(T) - 1)) == 0);
    T acc[NPAIR];
#pragma unroll
    for (int p = 0; p < NPAIR; ++p) acc[p] = (T)0;
    const size_t chunks = (dim + CPL - 1) / CPL;
    for (size_t q = (size_t)blockIdx.x * SVGD_THREADS + threadIdx.x; q < chunks; q += (size_t)gridDim.x * SVGD_THREADS) {
        const size_t c = q * CPL;
        T xv[NPAD][CPL];
        load_rows<T, NPAD, CPL>(X, ld, c, dim, n, aligned && c + CPL <= dim, xv);
        int p = 0;
#pragma unroll
        for (int i = 0; i < NPAD; ++i)
#pragma unroll
            for (int j = i + 1; j < NPAD; ++j, ++p) {
                if (j < n) {
#pragma unroll
                    for (int k = 0; k < CPL; ++k) {
                        const T d = xv[i][k] - xv[j][k];
                        acc[p] = fma_t(d, d, acc[p]);
                    }
                }
            }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int p = 0; p < NPAIR; ++p) {
        T v = acc[p];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) red[wave][p] = v;
    }
    __syncthreads();
    for (int p = threadIdx.x; p < NPAIR; p += SVGD_THREADS) {
        T v = red[0][p];
#pragma unroll
        for (int w = 1; w < SVGD_THREADS / 64; ++w) v += red[w][p];
        parts[(size_t)blockIdx.x * NPAIR + p] = v;
    }
}

// S2 for the dense pair layout of S1s
template <typename T, int NPAD>
__global__ __launch_bounds__(64 * SVGD_RED_SLICES) void svgd_reduce_small_kernel(const T *__restrict__ parts,
                                                                                  int n_parts, int n,
                                                                                  T *__restrict__ D) {
    constexpr int NPAIR = NPAD * (NPAD - 1) / 2;
    T s;
    int p;
    if (blockIdx.x == 0 && threadIdx.x < n) D[(size_t)threadIdx.x * n + threadIdx.x] = (T)0;
    if (!reduce_parts(parts, n_parts, NPAIR, s, p)) return;
    int i = 0, rem = p;                            // p = i * NPAD - i (i + 1) / 2 + (j - i - 1)
    while (rem >= NPAD - 1 - i) { rem -= NPAD - 1 - i; ++i; }
    const int j = i + 1 + rem;
    if (j >= n) return;
    const T dist = sqrt_t(s);                      // tf.norm, tensor_utils.py:399
    const T sq = dist * dist;                      // `** 2`, svgd.py:166
    D[(size_t)i * n + j] = sq;
    D[(size_t)j * n + i] = sq;
}

template <typename T, int NPAD, int CPL, bool UPDATE>
__global__ __launch_bounds__(SVGD_THREADS) void svgd_update_small_kernel(T *__restrict__ X, const T *__restrict__ G,
                                                                          T *__restrict__ H, T *__restrict__ kgrad_out,
                                                                          size_t dim, size_t ld, size_t out_ld, int n,
                                                                          const T *__restrict__ hdr,
                                                                          const T *__restrict__ K,
                                                                          const T *__restrict__ ksum, T eps, T alpha,
                                                                          T one_minus_alpha, T fudge, T sign) {
    const SvgdGeom g = svgd_geom(n);
    const T h2 = hdr[2];
    const T n_t = (T)n;
    const size_t c = ((size_t)blockIdx.x * SVGD_THREADS + threadIdx.x) * CPL;
    if (c >= dim) return;
    const bool full = c + CPL <= dim;
    const bool vec = full && (ld % CPL == 0) && ((reinterpret_cast<uintptr_t>(X) & (CPL * sizeof(T) - 1)) == 0) &&
                     (!UPDATE || (((reinterpret_cast<uintptr_t>(G) | reinterpret_cast<uintptr_t>(H)) &
                                   (CPL * sizeof(T) - 1)) == 0));
    const bool vec_out = full && (out_ld % CPL == 0) &&
                         ((reinterpret_cast<uintptr_t>(kgrad_out) & (CPL * sizeof(T) - 1)) == 0);
    T xv[NPAD][CPL], gv[NPAD][CPL], hv[NPAD][CPL];
    load_rows<T, NPAD, CPL>(X, ld, c, dim, n, vec, xv);
    if (UPDATE) {
        load_rows<T, NPAD, CPL>(G, ld, c, dim, n, vec, gv);
        load_rows<T, NPAD, CPL>(H, ld, c, dim, n, vec, hv);
    }
    T ks[NPAD];
#pragma unroll
    for (int i = 0; i < NPAD; ++i) ks[i] = ksum[i];
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        T accg[NPAD], accx[NPAD];
#pragma unroll
        for (int i = 0; i < NPAD; ++i) { accg[i] = (T)0; accx[i] = (T)0; }
        // rows n..15 of K are zero: no guard, the scalar loads stream ahead of the FMAs
#pragma unroll
        for (int j = 0; j < NPAD; ++j) {
            const T *krow = K + (size_t)j * g.np16;                  // K symmetric: row j, columns 0..NPAD-1
#pragma unroll
            for (int i = 0; i < NPAD; ++i) {
                const T kij = krow[i];
                accx[i] = fma_t(kij, xv[j][k], accx[i]);
                if (UPDATE) accg[i] = fma_t(kij, gv[j][k], accg[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < NPAD; ++i) {
            const T x = xv[i][k];
            const T kg = (-accx[i] + x * ks[i]) / h2;                        // svgd.py:176-181
            if (UPDATE) {
                const T gt = (accg[i] + sign * kg) / n_t;                    // svgd.py:124-127
                const T hnew = alpha * hv[i][k] + one_minus_alpha * (gt * gt);   // svgd.py:129-132
                const T adj = gt / (fudge + sqrt_t(hnew));                   // svgd.py:134-137
                hv[i][k] = hnew;
                gv[i][k] = x - eps * adj;                                    // svgd.py:139-143 (gv is free now)
            } else {
                gv[i][k] = kg;
            }
        }
    }
    if (UPDATE) {
        store_rows<T, NPAD, CPL>(X, ld, c, dim, n, vec, gv);
        store_rows<T, NPAD, CPL>(H, ld, c, dim, n, vec, hv);
    } else {
        store_rows<T, NPAD, CPL>(kgrad_out, out_ld, c, dim, n, vec_out, gv);
    }
}

// S4 for n <= NPAD <= 64: no LDS. A lane owns ONE column and keeps {G[j][c], X[j][c]} of all particles in
// registers (2 NPAD VGPRs, all loads in flight at once); every output row costs n packed FMAs
// {A_i, B_i} += K[i][j] * {g_j, x_j} (v_pk_fma_f32: fp32 VALU at the MFMA rate) with K[i][j] a scalar operand.
template <typename T> using Pair = T __attribute__((ext_vector_type(2)));

template <typename T, int NPAD, bool UPDATE>
__global__ __launch_bounds__(SVGD_THREADS) void svgd_update_reg_kernel(T *__restrict__ X, const T *__restrict__ G,
                                                                        T *__restrict__ H, T *__restrict__ kgrad_out,
                                                                        size_t dim, size_t ld, size_t out_ld, int n,
                                                                        const T *__restrict__ hdr,
                                                                        const T *__restrict__ K,
                                                                        const T *__restrict__ ksum, T eps, T alpha,
                                                                        T one_minus_alpha, T fudge, T sign) {
    const SvgdGeom g = svgd_geom(n);
    const T h2 = hdr[2];
    const T n_t = (T)n;
    const size_t c = (size_t)blockIdx.x * SVGD_THREADS + threadIdx.x;
    if (c >= dim) return;
    Pair<T> m[NPAD];
#pragma unroll
    for (int j = 0; j < NPAD; ++j) {
        m[j] = Pair<T>{(T)0, (T)0};
        if (j < n) {
            m[j].y = X[(size_t)j * ld + c];
            if (UPDATE) m[j].x = G[(size_t)j * ld + c];
        }
    }
#pragma unroll
    for (int it = 0; it < NPAD / SVGD_ITILE; ++it) {
        const int i0 = it * SVGD_ITILE;
        if (i0 >= n) break;
        T hold[SVGD_ITILE];
        if (UPDATE) {
#pragma unroll
            for (int a = 0; a < SVGD_ITILE; ++a) hold[a] = (i0 + a < n) ? H[(size_t)(i0 + a) * ld + c] : (T)0;
        }
        Pair<T> acc[SVGD_ITILE];
#pragma unroll
        for (int a = 0; a < SVGD_ITILE; ++a) acc[a] = Pair<T>{(T)0, (T)0};
        // rows n..np16-1 of K are zero, so a group of 16 j needs no per-row guard (the scalar loads of the
        // next rows are issued under the FMAs of the current one); at NPAD = 64 that schedule spills, so
        // rows are guarded one by one there
#pragma unroll
        for (int jg = 0; jg < NPAD / 16; ++jg) {
            if (16 * jg >= n) break;
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                const int j = 16 * jg + jj;
                if (NPAD <= 32 || j < n) {
                    const T *krow = K + (size_t)j * g.np16 + i0;   // K symmetric: row j, columns i0..i0+15
#pragma unroll
                    for (int a = 0; a < SVGD_ITILE; ++a) {
                        const T k = krow[a];
                        acc[a] = __builtin_elementwise_fma(Pair<T>{k, k}, m[j], acc[a]);
                    }
                }
            }
        }
#pragma unroll
        for (int a = 0; a < SVGD_ITILE; ++a) {
            const int i = i0 + a;
            if (i < n) {
                const T x = m[i0 + a].y;
                const T kg = (-acc[a].y + x * ksum[i]) / h2;                 // svgd.py:176-181
                if (UPDATE) {
                    const size_t at = (size_t)i * ld + c;
                    const T gt = (acc[a].x + sign * kg) / n_t;               // svgd.py:124-127
                    const T hnew = alpha * hold[a] + one_minus_alpha * (gt * gt);    // svgd.py:129-132
                    const T adj = gt / (fudge + sqrt_t(hnew));               // svgd.py:134-137
                    H[at] = hnew;
                    X[at] = x - eps * adj;                                   // svgd.py:139-143
                } else {
                    kgrad_out[(size_t)i * out_ld + c] = kg;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// S4 on the matrix cores, 32x32x2 form (used for f32 with 65 <= n <= 128; the 16x16x4 kernel below serves n <= 64):
// A = K G and B = K X with v_mfma_f32_32x32x2_f32 (exact f32,
// an fmaf chain per output). A workgroup owns a tile of 128 columns:
//   phase 1  all 256 lanes fetch the tile's G, X (into LDS, row-major) and H (registers) with 16-byte
//            accesses -- 512 contiguous bytes per particle row per wave instruction. With up to 192 row
//            streams 40 MB apart, 128-byte row segments (the natural MFMA operand fetch) ran at 2.4 TB/s;
//   phase 2  each wave multiplies its 32-column strip: A operand = K fragments (staged once per workgroup in
//            LDS in operand order, lane l: K[32 ib + (l & 31)][2 ks + (l >> 5)]), B operand = LDS tile element
//            [2 ks + (l >> 5)][column l & 31]; the outputs (column l & 31, row (r & 3) + 8 (r >> 2) + 4 (l >> 5)
//            for accumulator register r) are turned into grad_theta and written over the strip's G values;
//   phase 3  the element-wise tail runs row-major again: 4 columns per lane, 16-byte stores of X and H.
// ---------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int SVGD_MT = 128;                 // tile columns of the MFMA update kernel

template <int IB, int MT>
__global__ __launch_bounds__(SVGD_THREADS, (IB == 2 && MT == 64 ? 3 : IB <= 2 ? 2 : 1)) void svgd_update_mfma_kernel(float *__restrict__ X, const float *__restrict__ G,
                                                                         float *__restrict__ H, size_t dim, size_t ld,
                                                                         int n, const float *__restrict__ hdr,
                                                                         const float *__restrict__ K,
                                                                         const float *__restrict__ ksum, float eps,
                                                                         float alpha, float one_minus_alpha, float fudge,
                                                                         float sign) {
    constexpr int KSMAX = 16 * IB, NR = 32 * IB;
    constexpr int QPR = MT / 4, RSTEP = SVGD_THREADS / QPR, RPT = NR / RSTEP;   // row-major phases: quads per row, rows per lane
    constexpr int STRIPS = MT / 32, WPS = (SVGD_THREADS / 64) / STRIPS, IBW = IB / WPS;   // waves per strip, blocks per wave
    static_assert(IBW >= 1 && IBW * WPS == IB, "particle blocks must divide among the waves of a strip");
    extern __shared__ __align__(16) unsigned char svgd_lds_raw[];
    float *kfs = reinterpret_cast<float *>(svgd_lds_raw);           // [IB][KSMAX][64]
    float *gs = kfs + IB * KSMAX * 64;                              // [NR][MT]
    float *xs = gs + NR * MT;                                  // [NR][MT]
    const SvgdGeom g = svgd_geom(n);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int KS = (n + 1) / 2;
    const float h2 = hdr[2];
    const float n_t = (float)n;
    const bool vec = (ld % 4 == 0) && (((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(G) |
                                          reinterpret_cast<uintptr_t>(H)) & 15) == 0);

    for (int idx = t; idx < IB * KSMAX * 64; idx += SVGD_THREADS) {
        const int l = idx & 63, ks = (idx >> 6) % KSMAX, ib = (idx >> 6) / KSMAX;
        const int i = 32 * ib + (l & 31), j = 2 * ks + (l >> 5);
        kfs[idx] = (i < g.np16 && j < g.np16) ? K[(size_t)i * g.np16 + j] : 0.0f;         // zero beyond n
    }

    const int q = t % QPR, r0 = t / QPR;                            // row-major phases: 4 columns 4q.., rows r0 + RSTEP k
    const size_t n_tiles = (dim + MT - 1) / MT;
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t c0 = tile * MT;
        const size_t cq = c0 + 4 * (size_t)q;
        const bool fullq = vec && cq + 4 <= dim;
        // ---- phase 1
        f32x4 gv[RPT], xv[RPT], hv[RPT];
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int r = r0 + RSTEP * k;
            gv[k] = f32x4{0, 0, 0, 0};
            xv[k] = gv[k];
            hv[k] = gv[k];
            if (r < n) {
                const size_t at = (size_t)r * ld + cq;
                if (fullq) {
                    gv[k] = *reinterpret_cast<const f32x4 *>(G + at);
                    xv[k] = *reinterpret_cast<const f32x4 *>(X + at);
                    hv[k] = *reinterpret_cast<const f32x4 *>(H + at);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (cq + e < dim) { gv[k][e] = G[at + e]; xv[k][e] = X[at + e]; hv[k][e] = H[at + e]; }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int r = r0 + RSTEP * k;
            *reinterpret_cast<f32x4 *>(gs + r * MT + 4 * q) = gv[k];
            *reinterpret_cast<f32x4 *>(xs + r * MT + 4 * q) = xv[k];
        }
        __syncthreads();
        // ---- phase 2: this wave's 32-column strip (and, when two waves share a strip, its half of the blocks)
        {
            const int sc = (WPS == 1 ? wave : wave % STRIPS) * 32 + col;
            const int ib0 = WPS == 1 ? 0 : (wave / STRIPS) * IBW;
            f32x16 ag[IBW], ax[IBW];
#pragma unroll
            for (int ib = 0; ib < IBW; ++ib)
#pragma unroll
                for (int r = 0; r < 16; ++r) { ag[ib][r] = 0.0f; ax[ib][r] = 0.0f; }
            for (int ks = 0; ks < KS; ++ks) {
                const float bg = gs[(2 * ks + half) * MT + sc];
                const float bx = xs[(2 * ks + half) * MT + sc];
#pragma unroll
                for (int ib = 0; ib < IBW; ++ib) {
                    const float kf = kfs[((ib0 + ib) * KSMAX + ks) * 64 + lane];
                    ag[ib] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf, bg, ag[ib], 0, 0, 0);
                    ax[ib] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf, bx, ax[ib], 0, 0, 0);
                }
            }
            if (WPS > 1) __syncthreads();                           // the strip's other wave still reads G rows
#pragma unroll
            for (int ib = 0; ib < IBW; ++ib)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = 32 * (ib0 + ib) + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (i < n) {
                        const float x = xs[i * MT + sc];
                        const float kg = (-ax[ib][r] + x * ksum[i]) / h2;          // svgd.py:176-181
                        gs[i * MT + sc] = (ag[ib][r] + sign * kg) / n_t;           // svgd.py:124-127
                    }
                }
        }
        __syncthreads();
        // ---- phase 3
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int r = r0 + RSTEP * k;
            if (r < n && cq < dim) {
                const f32x4 gt = *reinterpret_cast<const f32x4 *>(gs + r * MT + 4 * q);
                const f32x4 xo = *reinterpret_cast<const f32x4 *>(xs + r * MT + 4 * q);   // not kept in registers
                f32x4 xn, hn;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float hnew = alpha * hv[k][e] + one_minus_alpha * (gt[e] * gt[e]);   // svgd.py:129-132
                    const float adj = gt[e] / (fudge + sqrt_t(hnew));                          // svgd.py:134-137
                    hn[e] = hnew;
                    xn[e] = xo[e] - eps * adj;                                                 // svgd.py:139-143
                }
                const size_t at = (size_t)r * ld + cq;
                if (fullq) {
                    *reinterpret_cast<f32x4 *>(H + at) = hn;
                    *reinterpret_cast<f32x4 *>(X + at) = xn;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (cq + e < dim) { H[at + e] = hn[e]; X[at + e] = xn[e]; }
                }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// S4 on the matrix cores, 16x16x4 form (f32 and f64, 9 <= n <= 64): the same three phases with
// v_mfma_{f32,f64}_16x16x4 -- 16-particle
// blocks, 16-column strips (4 waves = one 64-column tile), k-steps of 4 particles. Operand maps: A lane l =
// K[16 ib + (l & 15)][4 ks + (l >> 4)], B lane l = tile[4 ks + (l >> 4)][column l & 15]; output register r of
// lane l = row (l >> 4) + 4 r, column l & 15 (f64) / row 4 (l >> 4) + r (f32: the standard 16x16 map).
// ---------------------------------------------------------------------------------------------
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

template <typename T> struct Mfma16;                 // 16x16x4 matrix-core step for f32 / f64
template <> struct Mfma16<float> {
    typedef f32x4 Acc;
    static __device__ __forceinline__ Acc mma(float a, float b, Acc c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int lane_hi, int reg) { return 4 * lane_hi + reg; }   // standard 16x16 map
};
template <> struct Mfma16<double> {
    typedef f64x4 Acc;
    static __device__ __forceinline__ Acc mma(double a, double b, Acc c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int lane_hi, int reg) { return lane_hi + 4 * reg; }   // the f64 map
};


template <typename T, int IB>                  // 16-particle blocks: 1 (n <= 16), 2 (n <= 32) or 4 (n <= 64)
__global__ __launch_bounds__(SVGD_THREADS) void svgd_update_mfma16_kernel(T *__restrict__ X,
                                                                             const T *__restrict__ G,
                                                                             T *__restrict__ H, size_t dim, size_t ld,
                                                                             int n, const T *__restrict__ hdr,
                                                                             const T *__restrict__ K,
                                                                             const T *__restrict__ ksum, T eps,
                                                                             T alpha, T one_minus_alpha,
                                                                             T fudge, T sign) {
    constexpr int MT = 64, KSMAX = 4 * IB, NR = 16 * IB;
    constexpr int VW = 16 / (int)sizeof(T);                                     // elements per 16-byte access
    typedef T VecT __attribute__((ext_vector_type(VW)));
    constexpr int QPR = MT / VW, RSTEP = SVGD_THREADS / QPR, RPT = (NR + RSTEP - 1) / RSTEP;   // row-major phases
    extern __shared__ __align__(16) unsigned char svgd_lds_raw[];
    T *kfs = reinterpret_cast<T *>(svgd_lds_raw);                    // [IB][KSMAX][64]
    T *gs = kfs + IB * KSMAX * 64;                                   // [NR][MT]
    T *xs = gs + NR * MT;                                            // [NR][MT]
    const SvgdGeom g = svgd_geom(n);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int col = lane & 15, kq = lane >> 4;
    const int KS = (n + 3) / 4;
    const T h2 = hdr[2];
    const T n_t = (T)n;
    const bool vec = (ld % VW == 0) && (((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(G) |
                                          reinterpret_cast<uintptr_t>(H)) & 15) == 0);
    for (int idx = t; idx < IB * KSMAX * 64; idx += SVGD_THREADS) {
        const int l = idx & 63, ks = (idx >> 6) % KSMAX, ib = (idx >> 6) / KSMAX;
        const int i = 16 * ib + (l & 15), j = 4 * ks + (l >> 4);
        kfs[idx] = (i < g.np16 && j < g.np16) ? K[(size_t)i * g.np16 + j] : (T)0;         // zero beyond n
    }
    const int q = t % QPR, r0 = t / QPR;
    const size_t n_tiles = (dim + MT - 1) / MT;
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t c0 = tile * MT;
        const size_t cq = c0 + VW * (size_t)q;
        const bool fullq = vec && cq + VW <= dim;
        // ---- phase 1
        VecT gv[RPT], xv[RPT], hv[RPT];
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int r = r0 + RSTEP * k;
            gv[k] = VecT{};
            xv[k] = gv[k];
            hv[k] = gv[k];
            if (r < n) {
                const size_t at = (size_t)r * ld + cq;
                if (fullq) {
                    gv[k] = *reinterpret_cast<const VecT *>(G + at);
                    xv[k] = *reinterpret_cast<const VecT *>(X + at);
                    hv[k] = *reinterpret_cast<const VecT *>(H + at);
                } else {
#pragma unroll
                    for (int e = 0; e < VW; ++e)
                        if (cq + e < dim) { gv[k][e] = G[at + e]; xv[k][e] = X[at + e]; hv[k][e] = H[at + e]; }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int r = r0 + RSTEP * k;
            if (r < NR) {
                *reinterpret_cast<VecT *>(gs + r * MT + VW * q) = gv[k];
                *reinterpret_cast<VecT *>(xs + r * MT + VW * q) = xv[k];
            }
        }
        __syncthreads();
        // ---- phase 2: this wave's 16-column strip, all particle blocks
        {
            const int sc = wave * 16 + col;
            typename Mfma16<T>::Acc ag[IB], ax[IB];
#pragma unroll
            for (int ib = 0; ib < IB; ++ib)
#pragma unroll
                for (int r = 0; r < 4; ++r) { ag[ib][r] = (T)0; ax[ib][r] = (T)0; }
            for (int ks = 0; ks < KS; ++ks) {
                const T bg = gs[(4 * ks + kq) * MT + sc];
                const T bx = xs[(4 * ks + kq) * MT + sc];
#pragma unroll
                for (int ib = 0; ib < IB; ++ib) {
                    const T kf = kfs[(ib * KSMAX + ks) * 64 + lane];
                    ag[ib] = Mfma16<T>::mma(kf, bg, ag[ib]);
                    ax[ib] = Mfma16<T>::mma(kf, bx, ax[ib]);
                }
            }
#pragma unroll
            for (int ib = 0; ib < IB; ++ib)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * ib + Mfma16<T>::row(kq, r);
                    if (i < n) {
                        const T x = xs[i * MT + sc];
                        const T kg = (-ax[ib][r] + x * ksum[i]) / h2;              // svgd.py:176-181
                        gs[i * MT + sc] = (ag[ib][r] + sign * kg) / n_t;           // svgd.py:124-127
                    }
                }
        }
        __syncthreads();
        // ---- phase 3
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int r = r0 + RSTEP * k;
            if (r < n && cq < dim) {
                const VecT gt = *reinterpret_cast<const VecT *>(gs + r * MT + VW * q);
                const VecT xo = *reinterpret_cast<const VecT *>(xs + r * MT + VW * q);
                VecT xn, hn;
#pragma unroll
                for (int e = 0; e < VW; ++e) {
                    const T hnew = alpha * hv[k][e] + one_minus_alpha * (gt[e] * gt[e]);        // svgd.py:129-132
                    const T adj = gt[e] / (fudge + sqrt_t(hnew));                               // svgd.py:134-137
                    hn[e] = hnew;
                    xn[e] = xo[e] - eps * adj;                                                  // svgd.py:139-143
                }
                const size_t at = (size_t)r * ld + cq;
                if (fullq) {
                    *reinterpret_cast<VecT *>(H + at) = hn;
                    *reinterpret_cast<VecT *>(X + at) = xn;
                } else {
#pragma unroll
                    for (int e = 0; e < VW; ++e)
                        if (cq + e < dim) { H[at + e] = hn[e]; X[at + e] = xn[e]; }
                }
            }
        }
        __syncthreads();
    }
}

// S4 on the matrix cores, f64, 65 <= n <= 128: eight 16-particle blocks do not fit the LDS as K fragments next to the
// tile, so the products run in TWO passes of four blocks each (fragments of the pass staged from L2, 64 KB), on 32-column
// tiles; two waves share a 16-column strip and take two blocks of the pass each. grad_theta of pass 0 waits in registers
// until pass 1 has read the G rows.
__global__ __launch_bounds__(SVGD_THREADS) void svgd_update_mfma_f64_big_kernel(double *__restrict__ X,
                                                                                 const double *__restrict__ G,
                                                                                 double *__restrict__ H, size_t dim,
                                                                                 size_t ld, int n,
                                                                                 const double *__restrict__ hdr,
                                                                                 const double *__restrict__ K,
                                                                                 const double *__restrict__ ksum,
                                                                                 double eps, double alpha,
                                                                                 double one_minus_alpha, double fudge,
                                                                                 double sign) {
    constexpr int MT = 32, IBP = 4, KSMAX = 32, NR = 128;            // blocks per pass, k-steps, tile rows
    constexpr int QPR = MT / 2, RSTEP = SVGD_THREADS / QPR, RPT = NR / RSTEP;
    extern __shared__ __align__(16) unsigned char svgd_lds_raw[];
    double *kfs = reinterpret_cast<double *>(svgd_lds_raw);          // [IBP][KSMAX][64] of the current pass
    double *gs = kfs + IBP * KSMAX * 64;                             // [NR][MT]
    double *xs = gs + NR * MT;                                       // [NR][MT]
    const SvgdGeom g = svgd_geom(n);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int col = lane & 15, kq = lane >> 4;
    const int KS = (n + 3) / 4;
    const double h2 = hdr[2];
    const double n_t = (double)n;
    const bool vec = (ld % 2 == 0) && (((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(G) |
                                          reinterpret_cast<uintptr_t>(H)) & 15) == 0);
    const int q = t % QPR, r0 = t / QPR;
    const int sc = (wave & 1) * 16 + col;                            // strip of this wave
    const int ibw = (wave >> 1) * 2;                                 // its two blocks within a pass
    const size_t n_tiles = (dim + MT - 1) / MT;
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t cq = tile * MT + 2 * (size_t)q;
        const bool fullq = vec && cq + 2 <= dim;
        f64x2 gv[RPT], xv[RPT], hv[RPT];
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int r = r0 + RSTEP * k;
            gv[k] = f64x2{0, 0};
            xv[k] = gv[k];
            hv[k] = gv[k];
            if (r < n) {
                const size_t at = (size_t)r * ld + cq;
                if (fullq) {
                    gv[k] = *reinterpret_cast<const f64x2 *>(G + at);
                    xv[k] = *reinterpret_cast<const f64x2 *>(X + at);
                    hv[k] = *reinterpret_cast<const f64x2 *>(H + at);
                } else {
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        if (cq + e < dim) { gv[k][e] = G[at + e]; xv[k][e] = X[at + e]; hv[k][e] = H[at + e]; }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int r = r0 + RSTEP * k;
            *reinterpret_cast<f64x2 *>(gs + r * MT + 2 * q) = gv[k];
            *reinterpret_cast<f64x2 *>(xs + r * MT + 2 * q) = xv[k];
        }
        double gt_keep[2][2][4];                                     // [pass][block of the wave][register]
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            __syncthreads();                                         // tile written / previous pass done with kfs
            for (int idx = t; idx < IBP * KSMAX * 64; idx += SVGD_THREADS) {
                const int l = idx & 63, ks = (idx >> 6) % KSMAX, ib = (idx >> 6) / KSMAX;
                const int i = 16 * (IBP * pass + ib) + (l & 15), j = 4 * ks + (l >> 4);
                kfs[idx] = (i < g.np16 && j < g.np16) ? K[(size_t)i * g.np16 + j] : 0.0;
            }
            __syncthreads();
            f64x4 ag[2], ax[2];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) { ag[b][r] = 0.0; ax[b][r] = 0.0; }
            for (int ks = 0; ks < KS; ++ks) {
                const double bg = gs[(4 * ks + kq) * MT + sc];
                const double bx = xs[(4 * ks + kq) * MT + sc];
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const double kf = kfs[((ibw + b) * KSMAX + ks) * 64 + lane];
                    ag[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(kf, bg, ag[b], 0, 0, 0);
                    ax[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(kf, bx, ax[b], 0, 0, 0);
                }
            }
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * (IBP * pass + ibw + b) + kq + 4 * r;
                    double gt = 0.0;
                    if (i < n) {
                        const double x = xs[i * MT + sc];
                        const double kg = (-ax[b][r] + x * ksum[i]) / h2;          // svgd.py:176-181
                        gt = (ag[b][r] + sign * kg) / n_t;                         // svgd.py:124-127
                    }
                    gt_keep[pass][b][r] = gt;
                }
        }
        __syncthreads();                                             // every product has read the G rows
#pragma unroll
        for (int pass = 0; pass < 2; ++pass)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * (IBP * pass + ibw + b) + kq + 4 * r;
                    if (i < n) gs[i * MT + sc] = gt_keep[pass][b][r];
                }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int r = r0 + RSTEP * k;
            if (r < n && cq < dim) {
                const f64x2 gt = *reinterpret_cast<const f64x2 *>(gs + r * MT + 2 * q);
                const f64x2 xo = *reinterpret_cast<const f64x2 *>(xs + r * MT + 2 * q);
                f64x2 xn, hn;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const double hnew = alpha * hv[k][e] + one_minus_alpha * (gt[e] * gt[e]);   // svgd.py:129-132
                    const double adj = gt[e] / (fudge + sqrt_t(hnew));                          // svgd.py:134-137
                    hn[e] = hnew;
                    xn[e] = xo[e] - eps * adj;                                                  // svgd.py:139-143
                }
                const size_t at = (size_t)r * ld + cq;
                if (fullq) {
                    *reinterpret_cast<f64x2 *>(H + at) = hn;
                    *reinterpret_cast<f64x2 *>(X + at) = xn;
                } else {
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        if (cq + e < dim) { H[at + e] = hn[e]; X[at + e] = xn[e]; }
                }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// S1 on the matrix cores (f32, 17 <= n <= 128): Gram matrix of the particles, G = X~ X~^T, accumulated over
// 128-column tiles with v_mfma_f32_32x32x2_f32; |x_i - x_j|^2 = G_ii + G_jj - 2 G_ij afterwards (S3).
// Every tile is first CENTRED per column (x~_ic = x_ic - mean_i x_ic; distances do not change): the Gram
// entries are then of the size of the cloud's spread, not of |x|^2, and the subtraction loses nothing that
// matters for K = exp(-D / 2h^2) (relative error ~1e-6 in f32). Tile in LDS row-major (16-byte global
// loads); operand for particle block ib and k-step ks: lane l reads x~[32 ib + (l & 31)][2 ks + (l >> 5)];
// both MFMA operands come from the same registers (A block ib, B block jb), block pairs ib <= jb only.
// The 4 waves split the k-steps of a tile; their accumulators are added through LDS at the end.
// ---------------------------------------------------------------------------------------------
constexpr int SVGD_GP = SVGD_MT + 4;          // LDS row pitch of the Gram tile

template <int IB>
__global__ __launch_bounds__(SVGD_THREADS) void svgd_gram_mfma_kernel(const float *__restrict__ X, size_t dim, size_t ld,
                                                                       int n, float *__restrict__ parts) {
    constexpr int NR = 32 * IB, RPT = NR / 8, NPAIRB = IB * (IB + 1) / 2;
    extern __shared__ __align__(16) unsigned char svgd_lds_raw[];
    float *xs = reinterpret_cast<float *>(svgd_lds_raw);            // [NR][SVGD_GP]; later the reduction buffer
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int q = t & 31, r0 = t >> 5;
    const bool vec = (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
    const float inv_n = 1.0f / (float)n;

    f32x16 acc[NPAIRB];
#pragma unroll
    for (int p = 0; p < NPAIRB; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.0f;

    const size_t n_tiles = (dim + SVGD_MT - 1) / SVGD_MT;
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t cq = tile * SVGD_MT + 4 * (size_t)q;
        const bool fullq = vec && cq + 4 <= dim;
        f32x4 xv[RPT];
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int r = r0 + 8 * k;
            xv[k] = f32x4{0, 0, 0, 0};
            if (r < n) {
                const size_t at = (size_t)r * ld + cq;
                if (fullq) {
                    xv[k] = *reinterpret_cast<const f32x4 *>(X + at);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (cq + e < dim) xv[k][e] = X[at + e];
                }
            }
        }
        __syncthreads();                                            // the previous tile's operands are consumed
#pragma unroll
        for (int k = 0; k < RPT; ++k) *reinterpret_cast<f32x4 *>(xs + (r0 + 8 * k) * SVGD_GP + 4 * q) = xv[k];
        __syncthreads();
        if (t < SVGD_MT) {                                          // centre column t of the tile
            float s = 0.0f;
            for (int r = 0; r < n; ++r) s += xs[r * SVGD_GP + t];
            const float m = s * inv_n;
            for (int r = 0; r < n; ++r) xs[r * SVGD_GP + t] -= m;
        }
        __syncthreads();
        for (int ks = wave; ks < SVGD_MT / 2; ks += SVGD_THREADS / 64) {
            float a[IB];
#pragma unroll
            for (int ib = 0; ib < IB; ++ib) a[ib] = xs[(32 * ib + col) * SVGD_GP + 2 * ks + half];
            int p = 0;
#pragma unroll
            for (int ib = 0; ib < IB; ++ib)
#pragma unroll
                for (int jb = ib; jb < IB; ++jb, ++p)
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ib], a[jb], acc[p], 0, 0, 0);
        }
    }
    // add the 4 waves' accumulators in wave order, then one coalesced store of the workgroup's partial
    __syncthreads();
    for (int w = 0; w < SVGD_THREADS / 64; ++w) {
        if (wave == w) {
#pragma unroll
            for (int p = 0; p < NPAIRB; ++p)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float *slot = xs + (p * 16 + r) * 64 + lane;
                    *slot = (w == 0) ? acc[p][r] : (*slot + acc[p][r]);
                }
        }
        __syncthreads();
    }
    float *out = parts + (size_t)blockIdx.x * (NPAIRB * 1024);
    for (int idx = t; idx < NPAIRB * 1024; idx += SVGD_THREADS) out[idx] = xs[idx];
}

// S2 for the Gram partials: fixed-order sum, scatter into the [n x n] Gram matrix (row pitch np16)
template <int IB>
__global__ __launch_bounds__(64 * SVGD_RED_SLICES) void svgd_reduce_gram_kernel(const float *__restrict__ parts,
                                                                                 int n_parts, int n,
                                                                                 float *__restrict__ gram) {
    constexpr int NPAIRB = IB * (IB + 1) / 2;
    const SvgdGeom g = svgd_geom(n);
    float s;
    int idx;
    if (!reduce_parts(parts, n_parts, NPAIRB * 1024, s, idx)) return;
    int p = idx >> 10, ib = 0;
    while (p >= IB - ib) { p -= IB - ib; ++ib; }
    const int jb = ib + p;
    const int r = (idx >> 6) & 15, l = idx & 63;
    const int i = 32 * ib + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), j = 32 * jb + (l & 31);
    if (i >= n || j >= n) return;
    gram[(size_t)i * g.np16 + j] = s;
    if (ib != jb) gram[(size_t)j * g.np16 + i] = s;
}

// S1 on the matrix cores, f64 (17 <= n <= 64): as above with v_mfma_f64_16x16x4_f64 -- 16-particle blocks, 64-column
// tiles, 4 columns per instruction; operand lane l = x~[16 ib + (l & 15)][4 ks + (l >> 4)], output register r of lane l =
// G[16 ib + (l >> 4) + 4 r][16 jb + (l & 15)].
constexpr int SVGD_GT64 = 64;                  // tile columns; LDS pitch = 64 + one 16-byte vector

template <typename T, int IB>
__global__ __launch_bounds__(SVGD_THREADS) void svgd_gram_mfma16_kernel(const T *__restrict__ X, size_t dim, size_t ld,
                                                                         int n, T *__restrict__ parts) {
    constexpr int VW = 16 / (int)sizeof(T), SVGD_GP64 = SVGD_GT64 + VW;
    typedef T VecT __attribute__((ext_vector_type(VW)));
    constexpr int NR = 16 * IB, QPR = SVGD_GT64 / VW, RSTEP = SVGD_THREADS / QPR, RPT = (NR + RSTEP - 1) / RSTEP;
    constexpr int NPAIRB = IB * (IB + 1) / 2;
    extern __shared__ __align__(16) unsigned char svgd_lds_raw[];
    T *xs = reinterpret_cast<T *>(svgd_lds_raw);                     // [NR][SVGD_GP64]; later the reduction buffer
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int col = lane & 15, kq = lane >> 4;
    const int q = t % QPR, r0 = t / QPR;
    const bool vec = (ld % VW == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
    const T inv_n = (T)1 / (T)n;
    typename Mfma16<T>::Acc acc[NPAIRB];
#pragma unroll
    for (int p = 0; p < NPAIRB; ++p)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[p][r] = (T)0;
    const size_t n_tiles = (dim + SVGD_GT64 - 1) / SVGD_GT64;
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t cq = tile * SVGD_GT64 + VW * (size_t)q;
        const bool fullq = vec && cq + VW <= dim;
        VecT xv[RPT];
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int r = r0 + RSTEP * k;
            xv[k] = VecT{};
            if (r < n) {
                const size_t at = (size_t)r * ld + cq;
                if (fullq) {
                    xv[k] = *reinterpret_cast<const VecT *>(X + at);
                } else {
#pragma unroll
                    for (int e = 0; e < VW; ++e)
                        if (cq + e < dim) xv[k][e] = X[at + e];
                }
            }
        }
        __syncthreads();                                            // the previous tile's operands are consumed
#pragma unroll
        for (int k = 0; k < RPT; ++k)
            if (r0 + RSTEP * k < NR) *reinterpret_cast<VecT *>(xs + (r0 + RSTEP * k) * SVGD_GP64 + VW * q) = xv[k];
        __syncthreads();
        if (t < SVGD_GT64) {                                        // centre column t of the tile
            T s = (T)0;
            for (int r = 0; r < n; ++r) s += xs[r * SVGD_GP64 + t];
            const T m = s * inv_n;
            for (int r = 0; r < n; ++r) xs[r * SVGD_GP64 + t] -= m;
        }
        __syncthreads();
        for (int ks = wave; ks < SVGD_GT64 / 4; ks += SVGD_THREADS / 64) {
            T a[IB];
#pragma unroll
            for (int ib = 0; ib < IB; ++ib) a[ib] = xs[(16 * ib + col) * SVGD_GP64 + 4 * ks + kq];
            int p = 0;
#pragma unroll
            for (int ib = 0; ib < IB; ++ib)
#pragma unroll
                for (int jb = ib; jb < IB; ++jb, ++p)
                    acc[p] = Mfma16<T>::mma(a[ib], a[jb], acc[p]);
        }
    }
    __syncthreads();
    for (int w = 0; w < SVGD_THREADS / 64; ++w) {
        if (wave == w) {
#pragma unroll
            for (int p = 0; p < NPAIRB; ++p)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    T *slot = xs + (p * 4 + r) * 64 + lane;
                    *slot = (w == 0) ? acc[p][r] : (*slot + acc[p][r]);
                }
        }
        __syncthreads();
    }
    T *out = parts + (size_t)blockIdx.x * (NPAIRB * 256);
    for (int idx = t; idx < NPAIRB * 256; idx += SVGD_THREADS) out[idx] = xs[idx];
}

template <typename T, int IB>
__global__ __launch_bounds__(64 * SVGD_RED_SLICES) void svgd_reduce_gram16_kernel(const T *__restrict__ parts,
                                                                                   int n_parts, int n,
                                                                                   T *__restrict__ gram) {
    constexpr int NPAIRB = IB * (IB + 1) / 2;
    const SvgdGeom g = svgd_geom(n);
    T s;
    int idx;
    if (!reduce_parts(parts, n_parts, NPAIRB * 256, s, idx)) return;
    int p = idx >> 8, ib = 0;
    while (p >= IB - ib) { p -= IB - ib; ++ib; }
    const int jb = ib + p;
    const int r = (idx >> 6) & 3, l = idx & 63;
    const int i = 16 * ib + Mfma16<T>::row(l >> 4, r), j = 16 * jb + (l & 15);
    if (i >= n || j >= n) return;
    gram[(size_t)i * g.np16 + j] = s;
    if (ib != jb) gram[(size_t)j * g.np16 + i] = s;
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

template <typename T> struct SmallCfg {                    // columns per lane of the n <= 16 kernels
    static constexpr int CPL8 = 16 / sizeof(T);            // 16-byte accesses
    static constexpr int CPL16 = 8 / sizeof(T);            // 8-byte accesses (register budget)
};

template <typename T, int NPAD, int CPL>
int svgd_sqdist_small(const T *X, size_t n, size_t dim, size_t ld, T *parts, T *D, hipStream_t st) {
    const size_t chunks = (dim + CPL - 1) / CPL;
    const size_t blocks = (chunks + SVGD_THREADS - 1) / SVGD_THREADS;
    const int n_parts = (int)(blocks < (size_t)SVGD_MAX_PARTS ? blocks : (size_t)SVGD_MAX_PARTS);
    hipLaunchKernelGGL((svgd_sqdist_small_kernel<T, NPAD, CPL>), dim3(n_parts), dim3(SVGD_THREADS), 0, st, X, dim, ld,
                       (int)n, parts);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "launch svgd_sqdist_small_kernel");
    constexpr int NPAIR = NPAD * (NPAD - 1) / 2;
    hipLaunchKernelGGL((svgd_reduce_small_kernel<T, NPAD>), dim3((NPAIR + 63) / 64), dim3(64 * SVGD_RED_SLICES), 0, st,
                       parts, n_parts, (int)n, D);
    e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "launch svgd_reduce_small_kernel");
}

template <int IB>
int svgd_gram_launch_ib(const float *X, size_t n, size_t dim, size_t ld, float *parts, float *gram, hipStream_t st) {
    constexpr int NPAIRB = IB * (IB + 1) / 2;
    size_t lds_bytes = (size_t)32 * IB * SVGD_GP * sizeof(float);
    const size_t red_bytes = (size_t)NPAIRB * 1024 * sizeof(float);
    if (lds_bytes < red_bytes) lds_bytes = red_bytes;
    const size_t n_tiles = (dim + SVGD_MT - 1) / SVGD_MT;
    // partial buffers of NPAIRB * 1024 floats each must fit the workspace's partial area
    const size_t cap = IB == 1 ? 2048 : IB == 2 ? 1024 : 512;      // resident workgroups (LDS-limited: 8 / 4 / 2 per CU)
    const int n_parts = (int)(n_tiles < cap ? n_tiles : cap);
    if (lds_bytes > 64 * 1024) {
        hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void *>(&svgd_gram_mfma_kernel<IB>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e0 != hipSuccess) return hip_fail(e0, "hipFuncSetAttribute(svgd_gram_mfma_kernel)");
    }
    hipLaunchKernelGGL((svgd_gram_mfma_kernel<IB>), dim3(n_parts), dim3(SVGD_THREADS), lds_bytes, st, X, dim, ld, (int)n,
                       parts);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "launch svgd_gram_mfma_kernel");
    hipLaunchKernelGGL((svgd_reduce_gram_kernel<IB>), dim3(NPAIRB * 1024 / 64), dim3(64 * SVGD_RED_SLICES), 0, st, parts,
                       n_parts, (int)n, gram);
    e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "launch svgd_reduce_gram_kernel");
}

template <typename T, int IB>
int svgd_gram16_launch_ib(const T *X, size_t n, size_t dim, size_t ld, T *parts, T *gram, hipStream_t st) {
    constexpr int NPAIRB = IB * (IB + 1) / 2;
    constexpr int GP = SVGD_GT64 + 16 / (int)sizeof(T);
    size_t lds_bytes = (size_t)16 * IB * GP * sizeof(T);
    const size_t red_bytes = (size_t)NPAIRB * 256 * sizeof(T);
    if (lds_bytes < red_bytes) lds_bytes = red_bytes;
    const size_t n_tiles = (dim + SVGD_GT64 - 1) / SVGD_GT64;
    const size_t cap = IB <= 2 ? 2048 : 1024;
    const int n_parts = (int)(n_tiles < cap ? n_tiles : cap);
    hipLaunchKernelGGL((svgd_gram_mfma16_kernel<T, IB>), dim3(n_parts), dim3(SVGD_THREADS), lds_bytes, st, X, dim, ld,
                       (int)n, parts);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "launch svgd_gram_mfma16_kernel");
    hipLaunchKernelGGL((svgd_reduce_gram16_kernel<T, IB>), dim3(NPAIRB * 256 / 64), dim3(64 * SVGD_RED_SLICES), 0, st, parts,
                       n_parts, (int)n, gram);
    e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "launch svgd_reduce_gram16_kernel");
}

inline int svgd_gram_launch(const float *X, size_t n, size_t dim, size_t ld, float *parts, float *gram, hipStream_t st) {
    // up to 32 particles the 16x16x4 form on 64-column tiles wins (16 x 10 M: 172 vs 247 us for the register kernel,
    // 32: 296 vs ~330 us for the 32x32x2 form); at 64 the 32x32x2 form does (755 vs 820 us)
    if (n <= 16) return svgd_gram16_launch_ib<float, 1>(X, n, dim, ld, parts, gram, st);
    if (n <= 32) return svgd_gram16_launch_ib<float, 2>(X, n, dim, ld, parts, gram, st);
    if (n <= 64) return svgd_gram_launch_ib<2>(X, n, dim, ld, parts, gram, st);
    return svgd_gram_launch_ib<4>(X, n, dim, ld, parts, gram, st);
}
constexpr int SVGD_NOT_HANDLED = -1000;

// f64: matrix-core Gram form up to 64 particles; beyond, the difference-form kernel
inline int svgd_gram_launch(const double *X, size_t n, size_t dim, size_t ld, double *parts, double *gram, hipStream_t st) {
    if (n <= 16) return svgd_gram16_launch_ib<double, 1>(X, n, dim, ld, parts, gram, st);
    if (n <= 32) return svgd_gram16_launch_ib<double, 2>(X, n, dim, ld, parts, gram, st);
    if (n <= 64) return svgd_gram16_launch_ib<double, 4>(X, n, dim, ld, parts, gram, st);
    return SVGD_NOT_HANDLED;
}

template <typename T>
int svgd_kernel_matrix_impl(const T *X, size_t n, size_t dim, size_t ld, T *ws, hipStream_t st) {
    const SvgdGeom g = svgd_geom((int)n);
    const SvgdWs w = svgd_ws((int)n);
    T *parts = ws + w.parts;
    hipError_t e;
    int from_gram = 0;
    if (n >= 13) {                                               // n <= 12: the register kernel S1s is as fast or faster
        const int rc = svgd_gram_launch(X, n, dim, ld, parts, ws + w.K, st);
        if (rc != 0 && rc != SVGD_NOT_HANDLED) return rc;
        from_gram = rc == 0;                                     // not handled: f64 with more than 64 particles
    }
    if (from_gram) {
    } else if (n <= 16) {
        const int rc = (n <= 8) ? svgd_sqdist_small<T, 8, SmallCfg<T>::CPL8>(X, n, dim, ld, parts, ws + w.D, st)
                                : svgd_sqdist_small<T, 16, SmallCfg<T>::CPL16>(X, n, dim, ld, parts, ws + w.D, st);
        if (rc) return rc;
    } else {
        const int NP = g.np + 4;
        int tc_log2 = 4;
        while (tc_log2 < 10 && (size_t)(2 << tc_log2) * NP * sizeof(T) <= (size_t)SVGD_TILE_BYTES) ++tc_log2;
        const size_t tc = (size_t)1 << tc_log2;
        size_t lds_elems = tc * NP;
        if (lds_elems < 4096) lds_elems = 4096;                 // slice-reduction scratch
        const size_t n_tiles = (dim + tc - 1) / tc;
        const int n_parts = (int)(n_tiles < (size_t)SVGD_MAX_PARTS ? n_tiles : (size_t)SVGD_MAX_PARTS);
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
        e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "launch svgd_sqdist_kernel");
        const int red_blocks = (g.npb * 16 + 63) / 64;
        hipLaunchKernelGGL((svgd_reduce_kernel<T>), dim3(red_blocks), dim3(64 * SVGD_RED_SLICES), 0, st, parts, n_parts,
                           (int)n, ws + w.D);
        e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "launch svgd_reduce_kernel");
    }
    hipLaunchKernelGGL((svgd_bandwidth_kernel<T>), dim3(1), dim3(512), 0, st, ws + w.D, (int)n, ws + w.hdr, ws + w.K,
                       ws + w.ksum, from_gram);
    e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "launch svgd_bandwidth_kernel");
}

template <typename T, int NPAD, int CPL, bool UPDATE>
int svgd_apply_small(T *X, const T *G, T *H, T *kgrad_out, size_t out_ld, size_t n, size_t dim, size_t ld, T eps,
                     double alpha, T fudge, T sign, const T *ws, hipStream_t st) {
    const SvgdWs w = svgd_ws((int)n);
    const size_t chunks = (dim + CPL - 1) / CPL;
    const size_t blocks = (chunks + SVGD_THREADS - 1) / SVGD_THREADS;
    if (blocks > 0x7fffffffull) return fail(SGMCMC_EINVAL, "sgmcmc_svgd: dim too large");
    hipLaunchKernelGGL((svgd_update_small_kernel<T, NPAD, CPL, UPDATE>), dim3((unsigned)blocks), dim3(SVGD_THREADS), 0, st,
                       X, G, H, kgrad_out, dim, ld, out_ld, (int)n, ws + w.hdr, ws + w.K, ws + w.ksum, eps, (T)alpha,
                       (T)(1.0 - alpha), fudge, sign);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "launch svgd_update_small_kernel");
}

template <typename T, int NPAD, bool UPDATE>
int svgd_apply_reg(T *X, const T *G, T *H, T *kgrad_out, size_t out_ld, size_t n, size_t dim, size_t ld, T eps,
                   double alpha, T fudge, T sign, const T *ws, hipStream_t st) {
    const SvgdWs w = svgd_ws((int)n);
    const size_t blocks = (dim + SVGD_THREADS - 1) / SVGD_THREADS;
    if (blocks > 0x7fffffffull) return fail(SGMCMC_EINVAL, "sgmcmc_svgd: dim too large");
    hipLaunchKernelGGL((svgd_update_reg_kernel<T, NPAD, UPDATE>), dim3((unsigned)blocks), dim3(SVGD_THREADS), 0, st, X, G,
                       H, kgrad_out, dim, ld, out_ld, (int)n, ws + w.hdr, ws + w.K, ws + w.ksum, eps, (T)alpha,
                       (T)(1.0 - alpha), fudge, sign);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "launch svgd_update_reg_kernel");
}

template <typename T, bool UPDATE>
int svgd_apply_impl(T *X, const T *G, T *H, T *kgrad_out, size_t out_ld, size_t n, size_t dim, size_t ld, T eps,
                    double alpha, T fudge, T sign, const T *ws, hipStream_t st) {
    if (n <= 8)
        return svgd_apply_small<T, 8, SmallCfg<T>::CPL8, UPDATE>(X, G, H, kgrad_out, out_ld, n, dim, ld, eps, alpha, fudge,
                                                                 sign, ws, st);
    // f32 steps with 9..128 particles run on the matrix cores (measured: 763 vs 992 us at 16 x 10 M)
    if (n <= 16 && !UPDATE)
        return svgd_apply_small<T, 16, SmallCfg<T>::CPL16, UPDATE>(X, G, H, kgrad_out, out_ld, n, dim, ld, eps, alpha,
                                                                   fudge, sign, ws, st);
    if constexpr (UPDATE && sizeof(T) == 4) {
        const SvgdWs w = svgd_ws((int)n);
        float a32 = (float)alpha, oma32 = (float)(1.0 - alpha), eps32 = eps, fudge32 = fudge, sign32 = sign;
        int n_i = (int)n;
        const float *hdr = ws + w.hdr, *Kp = ws + w.K, *ksum = ws + w.ksum;
        void *args[] = {&X, &G, &H, &dim, &ld, &n_i, &hdr, &Kp, &ksum, &eps32, &a32, &oma32, &fudge32, &sign32};
        if (n <= 64) {
            // 16x16x4 instruction, 64-column tiles: 20 KB (n <= 32) / 48 KB of LDS per workgroup. Measured against the
            // 32x32x2 form with 128/64-column tiles: 889 vs 969 us at 16 x 10 M, 1.62 vs 1.71 ms at 32, 3.98 vs 4.14 at 64.
            // Many more workgroups than fit the chip balance better than a persistent grid (64 x 10 M: 4.14 ms at 8 k
            // workgroups vs 4.40 ms at 768, 5.2 ms uncapped).
            const int ib = n <= 16 ? 1 : n <= 32 ? 2 : 4;
            const size_t n_tiles = (dim + 63) / 64;
            const size_t lds_bytes = ((size_t)ib * 4 * ib * 64 + (size_t)2 * 16 * ib * 64) * sizeof(float);
            const unsigned grid = (unsigned)(n_tiles < 8192 ? n_tiles : 8192);
            const void *fn = ib == 1   ? reinterpret_cast<const void *>(&svgd_update_mfma16_kernel<float, 1>)
                             : ib == 2 ? reinterpret_cast<const void *>(&svgd_update_mfma16_kernel<float, 2>)
                                       : reinterpret_cast<const void *>(&svgd_update_mfma16_kernel<float, 4>);
            hipError_t e = hipLaunchKernel(fn, dim3(grid), dim3(SVGD_THREADS), args, lds_bytes, st);
            return e == hipSuccess ? 0 : hip_fail(e, "launch svgd_update_mfma16_kernel");
        }
        {
            // 65..128 particles: 32x32x2 instruction, 64-column tiles, two waves per 32-column strip with two of the four
            // particle blocks each; staging the 64 KB of K fragments per workgroup dominates, so the grid is resident-sized
            const size_t n_tiles = (dim + 63) / 64;
            const size_t lds_bytes = ((size_t)4 * 16 * 4 * 64 + (size_t)2 * 32 * 4 * 64) * sizeof(float);
            const unsigned grid = (unsigned)(n_tiles < 256 ? n_tiles : 256);
            const void *fn = reinterpret_cast<const void *>(&svgd_update_mfma_kernel<4, 64>);
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(svgd_update_mfma_kernel)");
            e = hipLaunchKernel(fn, dim3(grid), dim3(SVGD_THREADS), args, lds_bytes, st);
            return e == hipSuccess ? 0 : hip_fail(e, "launch svgd_update_mfma_kernel");
        }
    }
    if constexpr (UPDATE && sizeof(T) == 8) {
        if (n > 64) {
            const SvgdWs w = svgd_ws((int)n);
            const size_t n_tiles = (dim + 31) / 32;
            const size_t lds_bytes = ((size_t)4 * 32 * 64 + (size_t)2 * 128 * 32) * sizeof(double);     // 128 KB
            const unsigned grid = (unsigned)(n_tiles < 256 ? n_tiles : 256);
            const void *fn = reinterpret_cast<const void *>(&svgd_update_mfma_f64_big_kernel);
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(svgd_update_mfma_f64_big_kernel)");
            double a64 = alpha, oma64 = 1.0 - alpha, eps64 = eps, fudge64 = fudge, sign64 = sign;
            int n_i = (int)n;
            const double *hdr = ws + w.hdr, *Kp = ws + w.K, *ksum = ws + w.ksum;
            void *args[] = {&X, &G, &H, &dim, &ld, &n_i, &hdr, &Kp, &ksum, &eps64, &a64, &oma64, &fudge64, &sign64};
            e = hipLaunchKernel(fn, dim3(grid), dim3(SVGD_THREADS), args, lds_bytes, st);
            return e == hipSuccess ? 0 : hip_fail(e, "launch svgd_update_mfma_f64_big_kernel");
        }
        if (n <= 64) {
            const SvgdWs w = svgd_ws((int)n);
            const int ib = n <= 16 ? 1 : n <= 32 ? 2 : 4;
            const size_t n_tiles = (dim + 63) / 64;
            const size_t lds_bytes = ((size_t)ib * 4 * ib * 64 + (size_t)2 * 16 * ib * 64) * sizeof(double);
            const size_t cap = ib <= 2 ? 8192 : 256;                // as for f32: large grids balance better up to 32 particles
            const unsigned grid = (unsigned)(n_tiles < cap ? n_tiles : cap);
            const void *fn = ib == 1   ? reinterpret_cast<const void *>(&svgd_update_mfma16_kernel<double, 1>)
                             : ib == 2 ? reinterpret_cast<const void *>(&svgd_update_mfma16_kernel<double, 2>)
                                       : reinterpret_cast<const void *>(&svgd_update_mfma16_kernel<double, 4>);
            hipError_t e = hipSuccess;
            if (lds_bytes > 64 * 1024) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(svgd_update_mfma_f64_kernel)");
            double a64 = alpha, oma64 = 1.0 - alpha, eps64 = eps, fudge64 = fudge, sign64 = sign;
            int n_i = (int)n;
            const double *hdr = ws + w.hdr, *Kp = ws + w.K, *ksum = ws + w.ksum;
            void *args[] = {&X, &G, &H, &dim, &ld, &n_i, &hdr, &Kp, &ksum, &eps64, &a64, &oma64, &fudge64, &sign64};
            e = hipLaunchKernel(fn, dim3(grid), dim3(SVGD_THREADS), args, lds_bytes, st);
            return e == hipSuccess ? 0 : hip_fail(e, "launch svgd_update_mfma_f64_kernel");
        }
    }
    if (n <= 32)
        return svgd_apply_reg<T, 32, UPDATE>(X, G, H, kgrad_out, out_ld, n, dim, ld, eps, alpha, fudge, sign, ws, st);
    if (n <= 64 && sizeof(T) == 4)                              // 64 f64 pairs would not fit the register file
        return svgd_apply_reg<T, 64, UPDATE>(X, G, H, kgrad_out, out_ld, n, dim, ld, eps, alpha, fudge, sign, ws, st);
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
