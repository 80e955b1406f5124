// sgmcmc_kernels.hip -- fused SG-MCMC update kernels for MI355X (gfx950, CDNA4)
// and the C ABI declared in include/sgmcmc_hip.h.
//
// One launch per sampler step over a flat array of ALL parameters replaces the
// ~25 TensorFlow elementwise ops per parameter tensor that the reference runs in
// `session.run` (pysgmcmc/samplers/sghmc.py:165-251, sgld.py:149-211,
// relativistic_sghmc.py:120-140; driver pysgmcmc/samplers/base_classes.py:298-300).
//
// Design:
//   * HBM-bound elementwise pass, no contraction => no MFMA, no LDS staging of
//     the streamed arrays. Work unit = one "quad" of 4 consecutive elements per
//     lane: 16 B per lane per array (global_load/store_dwordx4, 1 KiB per wave
//     instruction), and exactly one Philox4x32-10 call per quad.
//   * Launch geometry is a knob (quads in flight per lane, grid cap, nt hints);
//     measured best on MI355X at 10 M params (gpurun tune, round 1): ONE quad per
//     lane, uncapped grid (~9.8 k blocks of 256), plain (not nt) accesses --
//     occupancy, not per-lane ILP, is what keeps HBM busy here. Those are the
//     defaults.
//   * Noise lives in registers only: Philox counter = (step, quad), key = seed,
//     Box-Muller on the hardware transcendental units (v_log_f32, v_sqrt_f32,
//     v_sin_f32, v_cos_f32). 0 bytes of HBM traffic for xi.
//   * All old state is read into registers before anything is written: that is
//     the tf.control_dependencies contract of sghmc.py:170-200 made structural.
//   * One IEEE rounding per reference op, reference op order
//     (-ffp-contract=off, correctly rounded '/' and sqrt), so injected-noise
//     results equal the CPU oracle bit for bit.
//   * K6 (summary) is the only kernel with a reduction: wave shuffles ->
//     LDS -> per-block partials -> fixed-order final pass.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "sgmcmc_stream.hpp"

namespace {


// Final pass: ONE block of 1024 lanes = 256 records x 4 statistics per trip. Lane t handles statistic t & 3 of records
// t >> 2, (t >> 2) + 256, ... (the 32-byte records are read as fully coalesced 8-byte elements, 8 loads in flight),
// then a fixed shuffle/LDS tree combines the 256 lanes of each statistic. Fixed association => deterministic.
__global__ void __launch_bounds__(1024) stats_final_kernel(const double *__restrict__ part, double *__restrict__ out4)
{
    __shared__ double lds[16][4];
    const unsigned nparts = (unsigned)reinterpret_cast<const unsigned long long *>(part)[0];
    const int k = threadIdx.x & 3, t = threadIdx.x >> 2;
    const int wave = threadIdx.x >> 6;
    const double *__restrict__ p = part + 4 + k;
    double v = 0.0;
    unsigned i = t;
    for (; i + 7u * 256u < nparts; i += 8u * 256u) {
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = p[4 * (size_t)(i + (unsigned)u * 256u)];
#pragma unroll
        for (int u = 0; u < 8; ++u) v += x[u];
    }
    for (; i < nparts; i += 256u) v += p[4 * (size_t)i];
    // lanes with equal (lane & 3) hold the same statistic: xor-shuffles over the other 4 lane bits
#pragma unroll
    for (int off = 32; off >= 4; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) < 4) lds[wave][k] = v;
    __syncthreads();
    if (threadIdx.x < 4) {
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) tot += lds[w][threadIdx.x];
        out4[threadIdx.x] = tot;
    }
}

// --------------------------------------------------------------------------
// K6 summary: wave shuffles -> LDS -> per-block partials -> fixed-order final
// --------------------------------------------------------------------------

constexpr int SUMMARY_BLOCKS = 1024;
constexpr int SUMMARY_THREADS = 256;

struct Summary { double s, ss, mn, mx; };

__device__ __forceinline__ Summary summary_combine(Summary a, Summary b)
{
    Summary o;
    o.s = a.s + b.s; o.ss = a.ss + b.ss;
    o.mn = a.mn < b.mn ? a.mn : b.mn;
    o.mx = a.mx > b.mx ? a.mx : b.mx;
    return o;
}
__device__ __forceinline__ Summary summary_wave_reduce(Summary v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        Summary o;
        o.s = __shfl_down(v.s, off, 64);
        o.ss = __shfl_down(v.ss, off, 64);
        o.mn = __shfl_down(v.mn, off, 64);
        o.mx = __shfl_down(v.mx, off, 64);
        v = summary_combine(v, o);
    }
    return v;
}
__device__ __forceinline__ Summary summary_block_reduce(Summary v)
{
    __shared__ Summary lds[SUMMARY_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = summary_wave_reduce(v);
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    if (wave == 0) {
        const int nw = blockDim.x >> 6;
        Summary w = lds[lane < nw ? lane : 0];
        if (lane >= nw) { w.s = 0; w.ss = 0; w.mn = INFINITY; w.mx = -INFINITY; }
        v = summary_wave_reduce(w);
    }
    return v;
}

template <typename T>
__global__ void __launch_bounds__(SUMMARY_THREADS) summary_partial(const T *__restrict__ x, size_t n, Summary *__restrict__ part)
{
    Summary acc = {0.0, 0.0, INFINITY, -INFINITY};
    const size_t G = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += G) {
        double v = (double)x[i];
        acc.s += v; acc.ss += v * v;
        acc.mn = v < acc.mn ? v : acc.mn;
        acc.mx = v > acc.mx ? v : acc.mx;
    }
    acc = summary_block_reduce(acc);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}
__global__ void __launch_bounds__(SUMMARY_THREADS) summary_final(const Summary *__restrict__ part, int nparts, double *__restrict__ out4)
{
    Summary acc = {0.0, 0.0, INFINITY, -INFINITY};
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) acc = summary_combine(acc, part[i]);
    acc = summary_block_reduce(acc);
    if (threadIdx.x == 0) { out4[0] = acc.s; out4[1] = acc.ss; out4[2] = acc.mn; out4[3] = acc.mx; }
}

// --------------------------------------------------------------------------
// R-hat pack / finish and raw Philox words (small elementwise kernels)
// --------------------------------------------------------------------------

// One IEEE rounding per operation in T (fp contract off): the C oracle's rhat_pack / rhat_finish give the same bits.
// Sharded layout: the padded index range [0, n_shards * shard_len) is cut into n_shards chunks, chunk s holds
// [mean | mean^2 | var] of parameters [s * shard_len, (s + 1) * shard_len) -- what reduce_scatter hands to rank s.
// n_shards = 1, shard_len = n is the plain [mean | mean^2 | var] layout of an all-reduce. Padding is written as 0.
template <typename T>
__global__ void __launch_bounds__(256) rhat_pack_kernel(const T *__restrict__ mean, const T *__restrict__ m2,
                                                        size_t n, T inv_cm1, size_t shard_len, size_t total,
                                                        T *__restrict__ out3)
{
    const size_t G = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += G) {
        const size_t s = i / shard_len, j = i - s * shard_len;
        T mu = T(0), var = T(0);
        if (i < n) { mu = mean[i]; var = m2[i] * inv_cm1; }
        T *o = out3 + s * 3 * shard_len + j;
        o[0] = mu;
        o[shard_len] = mu * mu;
        o[2 * shard_len] = var;
    }
}
// sum3 = [S_mean | S_sq | S_var] with row pitch ld; n = valid elements of this (shard of the) buffer
template <typename T>
__global__ void __launch_bounds__(256) rhat_finish_kernel(const T *__restrict__ sum3, size_t n, size_t ld, T m, T cnt,
                                                          T *__restrict__ rhat)
{
    const size_t G = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += G) {
        T s_mean = sum3[i], s_sq = sum3[ld + i], s_var = sum3[2 * ld + i];
        T W = s_var / m;
        T B = cnt * ((s_sq - (s_mean * s_mean) / m) / (m - T(1)));
        T Vhat = W * ((cnt - T(1)) / cnt) + B / cnt;
        rhat[i] = sqrt(Vhat / W);
    }
}
__global__ void __launch_bounds__(256) philox_bits_kernel(uint32_t *__restrict__ out, size_t n, NoiseKey nk)
{
    nk.resolve();
    const size_t G = (size_t)gridDim.x * blockDim.x;
    const size_t nq = (n + 3) / 4;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += G) {
        uint32_t x[4];
        philox_quad(nk, q, x);
#pragma unroll
        for (int j = 0; j < 4; ++j) if (4 * q + j < n) out[4 * q + j] = x[j];
    }
}


// --------------------------------------------------------------------------
// BNN cost path helpers (pysgmcmc/models/bayesian_neural_network.py:365-388): the
// loss head and the tanh backward, so that a whole BNN step is ~26 launches instead
// of ~90 tiny framework ops. Single-block / plain elementwise: launch-bound by design.
// --------------------------------------------------------------------------

struct BnnHeadConsts {
    double batch_size, n_examples, wp_den, lvp_den, ln_prior_mean, ln_prior_var, wdecay;
    int fold_prior_grad;     // 1: the update kernel adds the weight-prior gradient (grad_decay), omit it here
    int add_last_bias;       // 1: mean[] lacks the last layer's bias; add *last_bias
};

// mean[B], y[B]: network mean output and targets; s_ptr: the scalar log-variance parameter
// (output_bias); theta_sumsq: sum over ALL parameters of theta^2 (double, from the step kernel's
// fused statistics). Writes delta[B] = d cost/d mean, cost_out, grad_s_out (into the gradient
// arena slot of output_bias) and mse_out.
template <typename T>
__global__ void __launch_bounds__(1024) bnn_head_kernel(const T *__restrict__ mean, const T *__restrict__ y,
                                                       const T *__restrict__ s_ptr, const double *__restrict__ theta_sumsq,
                                                       const double *__restrict__ stats_ws, const T *__restrict__ last_bias,
                                                       size_t B, BnnHeadConsts k, T *__restrict__ delta,
                                                       T *__restrict__ cost_out, T *__restrict__ grad_s_out,
                                                       T *__restrict__ grad_bias_out, T *__restrict__ mse_out)
{
    __shared__ double lds[2][16];
    const double s = (double)*s_ptr;
    const double es = exp(s);
    const double inv = 1.0 / (es + 1e-16);                       // :369
    const double dscale = -(inv / k.batch_size);
    double sse = 0.0, sumr = 0.0;
    // add_bias: `mean` holds h W (no bias yet); the single-output layer's bias is added here
    const double bias_add = (k.add_last_bias && last_bias != nullptr) ? (double)*last_bias : 0.0;
    for (size_t i = threadIdx.x; i < B; i += blockDim.x) {
        double r = (double)y[i] - ((double)mean[i] + bias_add);
        sse += r * r;                                            // :370
        sumr += r;
        delta[i] = (T)(r * dscale);                              // d cost / d mean_i
    }
    // sum(theta^2): given directly, or as the per-block partials the previous step kernel left in its
    // statistics workspace (statistic 0 of the block-major records [nparts][4] after the 32-byte header), summed here in a
    // fixed order -- saves the separate K7 launch on the step's critical path
    double tsq = 0.0;
    if (stats_ws != nullptr) {
        const unsigned nparts = (unsigned)reinterpret_cast<const unsigned long long *>(stats_ws)[0];
        const double *__restrict__ p = stats_ws + 4;
        unsigned i = threadIdx.x;
        const unsigned bd = blockDim.x;
        // block-major 32-byte records: statistic 0 of record i is p[4 i]
        for (; i + 3u * bd < nparts; i += 4u * bd) {           // 4 loads in flight, fixed add order
            double x0 = p[4 * (size_t)i], x1 = p[4 * (size_t)(i + bd)], x2 = p[4 * (size_t)(i + 2u * bd)], x3 = p[4 * (size_t)(i + 3u * bd)];
            tsq += x0; tsq += x1; tsq += x2; tsq += x3;
        }
        for (; i < nparts; i += bd) tsq += p[4 * (size_t)i];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        sse += __shfl_down(sse, off, 64);
        sumr += __shfl_down(sumr, off, 64);
        tsq += __shfl_down(tsq, off, 64);
    }
    __shared__ double lds_t[16];
    if ((threadIdx.x & 63) == 0) { lds[0][threadIdx.x >> 6] = sse; lds[1][threadIdx.x >> 6] = sumr; lds_t[threadIdx.x >> 6] = tsq; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0, rs = 0.0, tq = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { tot += lds[0][w]; rs += lds[1][w]; tq += lds_t[w]; }
        if (stats_ws == nullptr) tq = *theta_sumsq;
        const double Bd = (double)B;
        double log_like = (-(tot * (0.5 * inv)) - 0.5 * s * Bd) / k.batch_size;            // :371-377
        double d = s - k.ln_prior_mean;
        double lvp = -(d * d) / k.lvp_den - 0.5 * k.ln_prior_var;                           // :102-107
        double wp = (-0.5 * k.wdecay) * tq / k.wp_den;                                      // :131-141
        double cost = -(log_like + lvp / k.n_examples + wp / k.n_examples);                 // :380-388
        double prior_coef = k.fold_prior_grad ? 0.0 : k.wdecay / (k.wp_den * k.n_examples);
        double ds = -((tot * (0.5 * es * inv * inv) - 0.5 * Bd) / k.batch_size
                      + (-2.0 * d / k.lvp_den) / k.n_examples) + prior_coef * s;
        *cost_out = (T)cost;
        *grad_s_out = (T)ds;
        *mse_out = (T)(tot / Bd);
        // bias gradient of the single-output last layer: sum_i delta_i (+ prior term unless folded)
        if (grad_bias_out != nullptr) *grad_bias_out = (T)(rs * dscale + prior_coef * (double)*last_bias);
    }
}

// delta *= (1 - h^2), the tanh backward (h = tanh(a) kept from the forward pass)
template <typename T>
__global__ void __launch_bounds__(256) tanh_backward_kernel(T *__restrict__ delta, const T *__restrict__ h, size_t n)
{
    const size_t G = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += G) {
        T hv = h[i];
        delta[i] = delta[i] * (T(1) - hv * hv);
    }
}

// Fused tanh backward + bias gradient of the layer below: delta[r][c] *= 1 - h[r][c]^2 and
// colsum[c] = sum_r delta[r][c] (+ beta * bias[c]). Row-major [rows][cols].
// One block of 1024 lanes owns COLS_PER_BLOCK = 16 columns (64-byte row segments): lane & 15 = column, the other 64
// "row lanes" (4 per wave x 16 waves) stride over the rows, so a 256 x 2048 matrix is 128 blocks (one per two CUs)
// instead of the 32 a 64-column block gives, and every lane has its 4 rows' loads in flight at once. The row lanes
// of a wave are combined by two shuffles, the 16 waves through LDS in a fixed order (deterministic, no atomics).
constexpr int CS_COLS = 16;
template <typename T>
__global__ void __launch_bounds__(1024) tanh_backward_colsum_kernel(T *__restrict__ delta, const T *__restrict__ h,
                                                                     size_t rows, size_t cols, const T *__restrict__ bias,
                                                                     T beta, T *__restrict__ colsum)
{
    __shared__ T lds[16][CS_COLS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cl = lane & (CS_COLS - 1);
    const size_t c = (size_t)blockIdx.x * CS_COLS + cl;
    const size_t rl = (size_t)wave * 4 + (lane >> 4);          // row lane 0..63
    T acc = T(0);
    if (c < cols) {
        size_t r = rl;
        for (; r + 192 < rows; r += 256) {                    // 4 rows per trip: 8 loads in flight per lane
            T hv[4], dv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const size_t i = (r + 64 * u) * cols + c; hv[u] = h[i]; dv[u] = delta[i]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                T d = dv[u] * (T(1) - hv[u] * hv[u]);
                delta[(r + 64 * u) * cols + c] = d;
                acc += d;
            }
        }
        for (; r < rows; r += 64) {
            const size_t i = r * cols + c;
            T hv = h[i];
            T d = delta[i] * (T(1) - hv * hv);
            delta[i] = d;
            acc += d;
        }
    }
    acc += __shfl_xor(acc, 16, 64);
    acc += __shfl_xor(acc, 32, 64);
    if (lane < CS_COLS) lds[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && lane < CS_COLS && c < cols) {
        T tot = T(0);
#pragma unroll
        for (int w = 0; w < 16; ++w) tot += lds[w][lane];
        colsum[c] = (beta != T(0)) ? tot + beta * bias[c] : tot;
    }
}

// Forward of the last hidden layer fused with the single-output layer above it:
//   h[r][c] = tanh(a[r][c]) in place,  out[r] = sum_c h[r][c] * w[c]   (the output unit's pre-bias mean)
// one workgroup per row, fixed summation tree (deterministic). Replaces a tanh launch and a GEMV launch.
__device__ __forceinline__ float tanh_dev(float x) { return tanhf(x); }
__device__ __forceinline__ double tanh_dev(double x) { return tanh(x); }

// 4 consecutive elements per lane per trip (16-byte accesses when the row pitch allows; all loads of a lane issued
// before the first tanh), one 256-lane workgroup per row.
// Optional side job (stats_ws != NULL): workgroups 0 .. min(TSQ_SLICES, rows) - 1 also add up one contiguous slice each of
// the sum(theta^2) partials the previous step kernel left in its statistics workspace and write it to tsq_parts[slice];
// the fused head (head_last_layer_backward_kernel) adds the slices in order. Saves the loss head's own pass over the
// ~10 k partials, and with it the separate head launch (each dependent launch of the step costs ~5 us).
// a[r][c] = tanh(a[r][c] + bias[c]) in place: the hidden layers' activation with the bias add that the forward GEMM then
// does not need as an epilogue (the library's plain product is 1.4-2.1 us faster than its bias-epilogue one at batch 256,
// round 3). One quad per lane per trip, 16-byte accesses when the pitch allows.
template <typename T>
__global__ void __launch_bounds__(256) bias_tanh_kernel(T *__restrict__ a, const T *__restrict__ bias, unsigned rows, unsigned cols)
{
    // 32-bit indices (the host checks rows * cols < 2^32): a 64-bit modulo per quad would cost more than the tanh
    const unsigned G = gridDim.x * blockDim.x, gid = blockIdx.x * blockDim.x + threadIdx.x;
    const bool vec = (cols % 4 == 0) && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(bias)) % (4 * sizeof(T)) == 0);
    if (vec) {
        struct alignas(4 * sizeof(T)) Q { T v[4]; };
        Q *aq = reinterpret_cast<Q *>(a);
        const Q *bq = reinterpret_cast<const Q *>(bias);
        const unsigned qpr = cols / 4, nq = rows * qpr;
        for (unsigned q = gid; q < nq; q += G) {
            Q x = aq[q];
            const Q b = bq[q % qpr];
#pragma unroll
            for (int j = 0; j < 4; ++j) x.v[j] = tanh_dev(x.v[j] + b.v[j]);
            aq[q] = x;
        }
    } else {
        const unsigned n = rows * cols;
        for (unsigned i = gid; i < n; i += G) a[i] = tanh_dev(a[i] + bias[i % cols]);
    }
}

constexpr int TSQ_SLICES = 16;
template <typename T>
__global__ void __launch_bounds__(256) tanh_rowdot_kernel(T *__restrict__ a, const T *__restrict__ w, size_t cols,
                                                          T *__restrict__ out, const double *__restrict__ stats_ws,
                                                          double *__restrict__ tsq_parts, const T *__restrict__ bias)
{
    __shared__ T lds[4];
    __shared__ double lds_d[4];
    const unsigned n_slices = gridDim.x < (unsigned)TSQ_SLICES ? gridDim.x : (unsigned)TSQ_SLICES;
    double tsq = 0.0;
    if (stats_ws != nullptr && blockIdx.x < n_slices) {
        const unsigned nparts = (unsigned)reinterpret_cast<const unsigned long long *>(stats_ws)[0];
        const double *__restrict__ p = stats_ws + 4;
        const unsigned len = (nparts + n_slices - 1) / n_slices;
        const unsigned lo = blockIdx.x * len, hi = (lo + len < nparts) ? lo + len : nparts;
        for (unsigned i = lo + threadIdx.x; i < hi; i += 256) tsq += p[4 * (size_t)i];      // statistic 0 of record i
    }
    T *row = a + (size_t)blockIdx.x * cols;
    T acc = T(0);
    const bool vec = (cols % 4 == 0) && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(w) |
                                          reinterpret_cast<uintptr_t>(bias)) % (4 * sizeof(T)) == 0);
    if (vec) {
        struct alignas(4 * sizeof(T)) Q { T v[4]; };
        Q *rq = reinterpret_cast<Q *>(row);
        const Q *wq = reinterpret_cast<const Q *>(w);
        const Q *bq = reinterpret_cast<const Q *>(bias);
        const Q zero = {{T(0), T(0), T(0), T(0)}};
        const size_t nq = cols / 4;
        size_t q = threadIdx.x;
        for (; q + 256 < nq; q += 512) {                      // two quads per lane in flight
            Q x0 = rq[q], x1 = rq[q + 256], w0 = wq[q], w1 = wq[q + 256];
            const Q b0 = bias ? bq[q] : zero, b1 = bias ? bq[q + 256] : zero;
#pragma unroll
            for (int j = 0; j < 4; ++j) { x0.v[j] = tanh_dev(x0.v[j] + b0.v[j]); x1.v[j] = tanh_dev(x1.v[j] + b1.v[j]); }
            rq[q] = x0; rq[q + 256] = x1;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc += x0.v[j] * w0.v[j];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc += x1.v[j] * w1.v[j];
        }
        for (; q < nq; q += 256) {
            Q x0 = rq[q], w0 = wq[q];
            const Q b0 = bias ? bq[q] : zero;
#pragma unroll
            for (int j = 0; j < 4; ++j) x0.v[j] = tanh_dev(x0.v[j] + b0.v[j]);
            rq[q] = x0;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc += x0.v[j] * w0.v[j];
        }
    } else {
        for (size_t c = threadIdx.x; c < cols; c += 256) {
            const T h = tanh_dev(row[c] + (bias ? bias[c] : T(0)));
            row[c] = h;
            acc += h * w[c];
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (stats_ws != nullptr && blockIdx.x < n_slices) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) tsq += __shfl_down(tsq, off, 64);
        if ((threadIdx.x & 63) == 0) lds_d[threadIdx.x >> 6] = tsq;
    }
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        out[blockIdx.x] = ((lds[0] + lds[1]) + lds[2]) + lds[3];
        if (stats_ws != nullptr && blockIdx.x < n_slices) tsq_parts[blockIdx.x] = ((lds_d[0] + lds_d[1]) + lds_d[2]) + lds_d[3];
    }
}

// Backward of a single-output last layer fused with the tanh backward of the layer below:
//   delta_prev[r][c] = dvec[r] * w[c] * (1 - h[r][c]^2)     (rank-1 back-propagation + tanh')
//   colsum[c]        = sum_r delta_prev[r][c] (+ beta * bias_prev[c])   bias gradient of the layer below
//   gw[c]            = sum_r h[r][c] * dvec[r] (+ beta * w[c])          weight gradient of the last layer
// Same block shape as tanh_backward_colsum_kernel (64 columns x 16 row-strided waves), deterministic.
template <typename T>
__global__ void __launch_bounds__(1024) last_layer_backward_kernel(const T *__restrict__ dvec, const T *__restrict__ w,
                                                                    const T *__restrict__ h, size_t rows, size_t cols,
                                                                    const T *__restrict__ bias_prev, T beta,
                                                                    T *__restrict__ delta_prev, T *__restrict__ colsum,
                                                                    T *__restrict__ gw)
{
    // same block shape as tanh_backward_colsum_kernel: 16 columns x 64 row lanes
    __shared__ T lds[2][16][CS_COLS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cl = lane & (CS_COLS - 1);
    const size_t c = (size_t)blockIdx.x * CS_COLS + cl;
    const size_t rl = (size_t)wave * 4 + (lane >> 4);
    T acc_b = T(0), acc_w = T(0);
    if (c < cols) {
        const T wc = w[c];
        size_t r = rl;
        for (; r + 192 < rows; r += 256) {
            T hv[4], dr[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { hv[u] = h[(r + 64 * u) * cols + c]; dr[u] = dvec[r + 64 * u]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const T d = (dr[u] * wc) * (T(1) - hv[u] * hv[u]);
                delta_prev[(r + 64 * u) * cols + c] = d;
                acc_b += d;
                acc_w += hv[u] * dr[u];
            }
        }
        for (; r < rows; r += 64) {
            const size_t i = r * cols + c;
            const T hv = h[i], dr = dvec[r];
            const T d = (dr * wc) * (T(1) - hv * hv);
            delta_prev[i] = d;
            acc_b += d;
            acc_w += hv * dr;
        }
    }
    acc_b += __shfl_xor(acc_b, 16, 64); acc_b += __shfl_xor(acc_b, 32, 64);
    acc_w += __shfl_xor(acc_w, 16, 64); acc_w += __shfl_xor(acc_w, 32, 64);
    if (lane < CS_COLS) { lds[0][wave][lane] = acc_b; lds[1][wave][lane] = acc_w; }
    __syncthreads();
    if (wave == 0 && lane < CS_COLS && c < cols) {
        T tb = T(0), tw = T(0);
#pragma unroll
        for (int k = 0; k < 16; ++k) { tb += lds[0][k][lane]; tw += lds[1][k][lane]; }
        colsum[c] = (beta != T(0)) ? tb + beta * bias_prev[c] : tb;
        gw[c] = (beta != T(0)) ? tw + beta * w[c] : tw;
    }
}

// The loss head (bnn_head_kernel) folded into the backward of a single-output last layer: dvec[r] = d cost / d mean_r is a
// function of the residual and the scalar log-variance only, so every workgroup forms it on the fly; one extra
// workgroup (the last of the grid, no columns of its own) reduces the residuals and writes the head's scalar outputs (cost,
// d cost/d log_var, mse, last bias gradient).
// sum(theta^2) arrives as the n_tsq slices tanh_rowdot_kernel left in tsq_parts. One launch less per step.
constexpr int HEAD_MAX_PART_ROWS = 1024;                     // batch rows when the mean arrives as partial dot products
template <typename T>
__global__ void __launch_bounds__(1024) head_last_layer_backward_kernel(
    const T *__restrict__ mean_parts, int n_mean_parts, const T *__restrict__ y, const T *__restrict__ s_ptr, const double *__restrict__ tsq_parts,
    int n_tsq, const T *__restrict__ last_bias, BnnHeadConsts k, T *__restrict__ cost_out, T *__restrict__ grad_s_out,
    T *__restrict__ grad_bias_out, T *__restrict__ mse_out, const T *__restrict__ w, const T *__restrict__ h, size_t rows,
    size_t cols, const T *__restrict__ bias_prev, T beta, T *__restrict__ delta_prev, T *__restrict__ colsum,
    T *__restrict__ gw)
{
    __shared__ T lds[2][16][CS_COLS];
    __shared__ double lds_h[2][16];
    __shared__ T mean_lds[HEAD_MAX_PART_ROWS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cl = lane & (CS_COLS - 1);
    // The LAST workgroup of the grid owns no columns: it does the head's own reductions and the scalar outputs (one thread's
    // ~1.5 us of dependent double-precision divisions at the end) next to the column workgroups instead of at the tail of one
    // of them (measured at 256 x 2048, round 4).
    const bool head_wg = blockIdx.x == gridDim.x - 1;
    const size_t c = head_wg ? cols : (size_t)blockIdx.x * CS_COLS + cl;
    const size_t rl = (size_t)wave * 4 + (lane >> 4);
    // Everything this lane will want from memory is requested FIRST -- its partial dot products, the activations and targets of
    // its first four rows, the scalars -- and the double-precision scalar chain (exp, reciprocal: ~1 us of dependent
    // instructions that used to start after the barrier) runs while those loads fly.
    // -- the output unit's pre-bias mean: a plain vector, or n_mean_parts partial dot products per row (what
    // sgmcmc_bnn_dense_tanh_f32 leaves: one per 64-column tile), added here in a fixed order by every workgroup: four adjacent
    // lanes per row, each adds a contiguous quarter of the parts (its loads issued together, not one dependent round trip per
    // part), then the quarters are added in lane order
    constexpr int PRE = 8;                                       // parts per lane requested ahead (32 parts: all of them)
    const int per = (n_mean_parts + 3) / 4;
    const size_t pr = threadIdx.x >> 2;                          // first trip: row and quarter of this lane
    const int pq = (int)(threadIdx.x & 3), plo = pq * per, phi = (plo + per < n_mean_parts) ? plo + per : n_mean_parts;
    const bool pre_parts = n_mean_parts > 1 && threadIdx.x < 4 * rows && per <= PRE;
    T pv[PRE];
    if (pre_parts) {
#pragma unroll
        for (int u = 0; u < PRE; ++u) pv[u] = (plo + u < phi) ? mean_parts[(size_t)(plo + u) * rows + pr] : T(0);
    }
    const bool pre_rows = c < cols && rl + 192 < rows;           // the first trip of the main loop
    T hv0[4], yv0[4];
    if (pre_rows) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { hv0[u] = h[(rl + 64 * u) * cols + c]; yv0[u] = y[rl + 64 * u]; }
    }
    const T wc = (c < cols) ? w[c] : T(0);
    const double s = (double)*s_ptr;
    const double bias_add = (k.add_last_bias && last_bias != nullptr) ? (double)*last_bias : 0.0;
    const double es = exp(s);
    const double inv = 1.0 / (es + 1e-16);                       // :369
    const double dscale = -(inv / k.batch_size);
    // (head workgroup, thread 0) whatever of the scalar outputs does not depend on the residuals: before the barriers, not after
    double tq = 0.0, lvp = 0.0, dlv = 0.0, prior_coef = 0.0;
    if (head_wg && threadIdx.x == 0) {
        for (int j = 0; j < n_tsq; ++j) tq += tsq_parts[j];
        dlv = s - k.ln_prior_mean;
        lvp = -(dlv * dlv) / k.lvp_den - 0.5 * k.ln_prior_var;                              // :102-107
        prior_coef = k.fold_prior_grad ? 0.0 : k.wdecay / (k.wp_den * k.n_examples);
    }
    const T *__restrict__ mean = mean_parts;
    if (n_mean_parts > 1) {
        for (size_t i = threadIdx.x; i < 4 * rows; i += blockDim.x) {   // rows <= 1024: whole waves enter each trip
            const size_t r = i >> 2;
            const int q = (int)(i & 3), lo = q * per, hi = (lo + per < n_mean_parts) ? lo + per : n_mean_parts;
            T m = T(0);
            int p = lo;
            if (pre_parts && i == threadIdx.x) {
                // the same left-to-right sum as the loop below
#pragma unroll
                for (int u = 0; u < PRE; ++u)
                    if (lo + u < hi) m += pv[u];
            } else {
                for (; p + 4 <= hi; p += 4) {
                    const T v0 = mean_parts[(size_t)p * rows + r], v1 = mean_parts[(size_t)(p + 1) * rows + r];
                    const T v2 = mean_parts[(size_t)(p + 2) * rows + r], v3 = mean_parts[(size_t)(p + 3) * rows + r];
                    m = (((m + v0) + v1) + v2) + v3;
                }
                for (; p < hi; ++p) m += mean_parts[(size_t)p * rows + r];
            }
            const T m1 = __shfl_down(m, 1, 64), m2 = __shfl_down(m, 2, 64), m3 = __shfl_down(m, 3, 64);
            if (q == 0) mean_lds[r] = ((m + m1) + m2) + m3;
        }
        __syncthreads();
        mean = mean_lds;
    }
    auto dvec_y = [&](T yr, size_t r) -> T { return (T)(((double)yr - ((double)mean[r] + bias_add)) * dscale); };
    auto dvec = [&](size_t r) -> T { return dvec_y(y[r], r); };
    T acc_b = T(0), acc_w = T(0);
    if (c < cols) {
        size_t r = rl;
        if (pre_rows) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const T dr = dvec_y(yv0[u], r + 64 * u);
                const T d = (dr * wc) * (T(1) - hv0[u] * hv0[u]);
                delta_prev[(r + 64 * u) * cols + c] = d;
                acc_b += d;
                acc_w += hv0[u] * dr;
            }
            r += 256;
        }
        for (; r + 192 < rows; r += 256) {
            T hv[4], dr[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { hv[u] = h[(r + 64 * u) * cols + c]; dr[u] = dvec(r + 64 * u); }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const T d = (dr[u] * wc) * (T(1) - hv[u] * hv[u]);
                delta_prev[(r + 64 * u) * cols + c] = d;
                acc_b += d;
                acc_w += hv[u] * dr[u];
            }
        }
        for (; r < rows; r += 64) {
            const size_t i = r * cols + c;
            const T hv = h[i], dr = dvec(r);
            const T d = (dr * wc) * (T(1) - hv * hv);
            delta_prev[i] = d;
            acc_b += d;
            acc_w += hv * dr;
        }
    }
    acc_b += __shfl_xor(acc_b, 16, 64); acc_b += __shfl_xor(acc_b, 32, 64);
    acc_w += __shfl_xor(acc_w, 16, 64); acc_w += __shfl_xor(acc_w, 32, 64);
    if (lane < CS_COLS) { lds[0][wave][lane] = acc_b; lds[1][wave][lane] = acc_w; }
    // the head's reductions (same arithmetic as bnn_head_kernel)
    double sse = 0.0, sumr = 0.0;
    if (head_wg) {
        for (size_t i = threadIdx.x; i < rows; i += blockDim.x) {
            double r = (double)y[i] - ((double)mean[i] + bias_add);
            sse += r * r;                                            // :370
            sumr += r;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { sse += __shfl_down(sse, off, 64); sumr += __shfl_down(sumr, off, 64); }
        if (lane == 0) { lds_h[0][wave] = sse; lds_h[1][wave] = sumr; }
    }
    __syncthreads();
    if (wave == 0 && lane < CS_COLS && c < cols) {
        T tb = T(0), tw = T(0);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) { tb += lds[0][kk][lane]; tw += lds[1][kk][lane]; }
        colsum[c] = (beta != T(0)) ? tb + beta * bias_prev[c] : tb;
        gw[c] = (beta != T(0)) ? tw + beta * w[c] : tw;
    }
    if (head_wg && threadIdx.x == 0) {
        double tot = 0.0, rs = 0.0;
        for (int wv = 0; wv < 16; ++wv) { tot += lds_h[0][wv]; rs += lds_h[1][wv]; }
        const double Bd = (double)rows;
        double log_like = (-(tot * (0.5 * inv)) - 0.5 * s * Bd) / k.batch_size;            // :371-377
        double wp = (-0.5 * k.wdecay) * tq / k.wp_den;                                      // :131-141
        double cost = -(log_like + lvp / k.n_examples + wp / k.n_examples);                 // :380-388
        double ds = -((tot * (0.5 * es * inv * inv) - 0.5 * Bd) / k.batch_size
                      + (-2.0 * dlv / k.lvp_den) / k.n_examples) + prior_coef * s;
        *cost_out = (T)cost;
        *grad_s_out = (T)ds;
        *mse_out = (T)(tot / Bd);
        if (grad_bias_out != nullptr) *grad_bias_out = (T)(rs * dscale + prior_coef * (double)*last_bias);
    }
}

// --------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------

}  // namespace

// error reporting shared by the translation units of the library (sgmcmc_host.hpp)
namespace sgmcmc_host {
thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
int hip_fail(hipError_t e, const char *what)
{
    return fail((int)e, "%s: %s", what, hipGetErrorString(e));
}
}  // namespace sgmcmc_host
using namespace sgmcmc_host;

namespace {

__global__ void counter_add_kernel(uint64_t *ctr, uint64_t inc) { *ctr += inc; }

// minibatch window [start, start + B) of the resident dataset into the static feed buffers
// (pysgmcmc/data_batches.py:118-123): x rows are contiguous, so the window is ONE contiguous range of X
template <typename T>
__global__ void window_gather_kernel(const T *__restrict__ X, const T *__restrict__ y, size_t start, size_t B, size_t D,
                                     T *__restrict__ xb, size_t ldx, T *__restrict__ yb)
{
    const size_t nx = B * D;
    const T *__restrict__ src = X + start * D;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, G = (size_t)gridDim.x * blockDim.x;
    // 16-byte copies when source window and destination are 16-byte aligned (4 elements of f32, 2 of f64 per access)
    constexpr size_t V = 16 / sizeof(T);
    struct alignas(16) Q { T v[V]; };
    if (ldx == D) {
        const bool vec = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(xb)) & 15u) == 0;
        const size_t nq = vec ? nx / V : 0;
        for (size_t q = gid; q < nq; q += G) reinterpret_cast<Q *>(xb)[q] = reinterpret_cast<const Q *>(src)[q];
        for (size_t i = nq * V + gid; i < nx; i += G) xb[i] = src[i];
    } else {
        // pitched destination (row stride ldx > D; the columns beyond D are the caller's): row by row
        const bool vec = D % V == 0 && ldx % V == 0 &&
                         ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(xb)) & 15u) == 0;
        if (vec) {
            const size_t qpr = D / V, lq = ldx / V;
            for (size_t q = gid; q < B * qpr; q += G) {
                const size_t r = q / qpr, c = q - r * qpr;
                reinterpret_cast<Q *>(xb)[r * lq + c] = reinterpret_cast<const Q *>(src)[q];
            }
        } else {
            for (size_t i = gid; i < nx; i += G) {
                const size_t r = i / D, c = i - r * D;
                xb[r * ldx + c] = src[i];
            }
        }
    }
    for (size_t i = gid; i < B; i += G) yb[i] = y[start + i];
}

template <typename T>
int summary(const T *x, size_t n, double *out4, void *ws, hipStream_t st)
{
    if (!x || !out4 || !ws) return fail(SGMCMC_EINVAL, "summary: NULL argument");
    size_t want = (n + SUMMARY_THREADS - 1) / SUMMARY_THREADS;
    int blocks = (int)(want < (size_t)SUMMARY_BLOCKS ? (want ? want : 1) : SUMMARY_BLOCKS);
    Summary *part = static_cast<Summary *>(ws);
    hipLaunchKernelGGL((summary_partial<T>), dim3(blocks), dim3(SUMMARY_THREADS), 0, st, x, n, part);
    hipLaunchKernelGGL(summary_final, dim3(1), dim3(SUMMARY_THREADS), 0, st, part, blocks, out4);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "launch summary");
}

inline unsigned small_grid(size_t n)
{
    size_t want = (n + 255) / 256;
    size_t cap = (size_t)1 << 20;
    return (unsigned)(want < cap ? (want ? want : 1) : cap);
}

template <typename T>
int bnn_head(const T *mean, const T *y, const T *s_ptr, const double *theta_sumsq, const void *stats_ws,
             const T *last_bias, size_t B,
                    double batch_size, double n_examples, double n_params, double wdecay, double prior_mean,
                    double prior_var, int fold_prior_grad, T *delta, T *cost_out, T *grad_s_out, T *grad_bias_out,
             T *mse_out, hipStream_t st)
{
    if (!mean || !y || !s_ptr || (!theta_sumsq && !stats_ws) || !delta || !cost_out || !grad_s_out || !mse_out || B == 0 ||
        (grad_bias_out && !last_bias))
        return fail(SGMCMC_EINVAL, "bnn_head: NULL argument or B == 0");
    BnnHeadConsts k;
    k.batch_size = batch_size; k.n_examples = n_examples; k.wdecay = wdecay;
    k.wp_den = n_params + (2.0 * 1e-16 + 1e-16);                 /* safe_divide, n_params > 0 */
    k.lvp_den = 2.0 * prior_var + (2.0 * 1e-16 + 1e-16);
    k.ln_prior_mean = std::log(prior_mean); k.ln_prior_var = std::log(prior_var);
    k.fold_prior_grad = (fold_prior_grad & 1) ? 1 : 0;
    k.add_last_bias = (fold_prior_grad & 2) ? 1 : 0;
    hipLaunchKernelGGL((bnn_head_kernel<T>), dim3(1), dim3(1024), 0, st, mean, y, s_ptr, theta_sumsq,
                       static_cast<const double *>(stats_ws), last_bias, B, k, delta, cost_out, grad_s_out, grad_bias_out,
                       mse_out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "launch bnn_head");
}


template <typename T>
int last_layer_backward_impl(const T *dvec, const T *w, const T *h, size_t rows, size_t cols, const T *bias_prev, T beta,
                             T *delta_prev, T *colsum, T *gw, hipStream_t st)
{
    if (rows == 0 || cols == 0) return 0;
    if (!dvec || !w || !h || !delta_prev || !colsum || !gw || (beta != T(0) && !bias_prev))
        return fail(SGMCMC_EINVAL, "last_layer_backward: NULL argument");
    hipLaunchKernelGGL((last_layer_backward_kernel<T>), dim3((unsigned)((cols + CS_COLS - 1) / CS_COLS)), dim3(1024), 0, st, dvec, w, h,
                       rows, cols, bias_prev, beta, delta_prev, colsum, gw);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "launch last_layer_backward");
}

template <typename T>
int head_last_layer_backward_impl(const T *mean, size_t n_mean_parts, const T *y, const T *s_ptr, const double *tsq_parts, const T *last_bias,
                                  size_t rows, size_t cols, double batch_size, double n_examples, double n_params,
                                  double wdecay, double prior_mean, double prior_var, int flags, const T *w, const T *h,
                                  const T *bias_prev, T beta, T *cost_out, T *grad_s_out, T *grad_bias_out, T *mse_out,
                                  T *delta_prev, T *colsum, T *gw, hipStream_t st)
{
    if (!mean || !y || !s_ptr || !tsq_parts || !w || !h || !cost_out || !grad_s_out || !mse_out || !delta_prev || !colsum ||
        !gw || rows == 0 || cols == 0 || (grad_bias_out && !last_bias) || (beta != T(0) && !bias_prev))
        return fail(SGMCMC_EINVAL, "bnn_head_last_layer_backward: NULL argument or empty matrix");
    if (n_mean_parts == 0 || n_mean_parts > 4096 || (n_mean_parts > 1 && rows > (size_t)HEAD_MAX_PART_ROWS))
        return fail(SGMCMC_EINVAL, "bnn_head_last_layer_backward: n_mean_parts must be 1 .. 4096 (and rows <= 1024 when > 1)");
    BnnHeadConsts k;
    k.batch_size = batch_size; k.n_examples = n_examples; k.wdecay = wdecay;
    k.wp_den = n_params + (2.0 * 1e-16 + 1e-16);                 /* safe_divide, n_params > 0 */
    k.lvp_den = 2.0 * prior_var + (2.0 * 1e-16 + 1e-16);
    k.ln_prior_mean = std::log(prior_mean); k.ln_prior_var = std::log(prior_var);
    k.fold_prior_grad = (flags & 1) ? 1 : 0;
    k.add_last_bias = (flags & 2) ? 1 : 0;
    // slices of sum(theta^2) the forward launch left: min(16, its workgroups) -- one workgroup per row (tanh_rowdot), or per
    // 32 x 64 output tile (bnn_dense_tanh, which callers use only with >= 16 tiles)
    const int n_tsq = (int)((n_mean_parts > 1 || rows >= (size_t)TSQ_SLICES) ? (size_t)TSQ_SLICES : rows);
    hipLaunchKernelGGL((head_last_layer_backward_kernel<T>), dim3((unsigned)((cols + CS_COLS - 1) / CS_COLS) + 1u), dim3(1024), 0, st,
                       mean, (int)n_mean_parts, y, s_ptr, tsq_parts, n_tsq, last_bias, k, cost_out, grad_s_out, grad_bias_out, mse_out, w, h,
                       rows, cols, bias_prev, beta, delta_prev, colsum, gw);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "launch head_last_layer_backward");
}

template <typename T>
int tanh_backward_colsum_impl(T *delta, const T *h, size_t rows, size_t cols, const T *bias, T beta, T *colsum,
                                     hipStream_t st)
{
    if (rows == 0 || cols == 0) return 0;
    if (!delta || !h || !colsum || (beta != T(0) && !bias))
        return fail(SGMCMC_EINVAL, "tanh_backward_colsum: NULL argument");
    hipLaunchKernelGGL((tanh_backward_colsum_kernel<T>), dim3((unsigned)((cols + CS_COLS - 1) / CS_COLS)), dim3(1024), 0, st, delta, h,
                       rows, cols, bias, beta, colsum);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "launch tanh_backward_colsum");
}

}  // namespace

// --------------------------------------------------------------------------
// C ABI
// --------------------------------------------------------------------------

extern "C" {

int sgmcmc_abi_version(void) { return SGMCMC_ABI_VERSION; }
const char *sgmcmc_last_error(void) { return g_err; }

int sgmcmc_event_create(void **event_out)
{
    if (!event_out) return fail(SGMCMC_EINVAL, "event_create: event_out is NULL");
    hipEvent_t ev = nullptr;
    hipError_t e = hipEventCreate(&ev);
    if (e != hipSuccess) return hip_fail(e, "hipEventCreate");
    *event_out = ev;
    return 0;
}
int sgmcmc_event_destroy(void *event)
{
    if (!event) return 0;
    hipError_t e = hipEventDestroy(static_cast<hipEvent_t>(event));
    return e == hipSuccess ? 0 : hip_fail(e, "hipEventDestroy");
}
int sgmcmc_event_elapsed_ms(void *start_event, void *stop_event, float *ms_out)
{
    if (!start_event || !stop_event || !ms_out) return fail(SGMCMC_EINVAL, "event_elapsed_ms: NULL argument");
    hipError_t e = hipEventElapsedTime(ms_out, static_cast<hipEvent_t>(start_event), static_cast<hipEvent_t>(stop_event));
    return e == hipSuccess ? 0 : hip_fail(e, "hipEventElapsedTime");
}

int sgmcmc_event_synchronize(void *event)
{
    if (!event) return fail(SGMCMC_EINVAL, "event_synchronize: NULL argument");
    hipError_t e = hipEventSynchronize(static_cast<hipEvent_t>(event));
    return e == hipSuccess ? 0 : hip_fail(e, "hipEventSynchronize");
}

int sgmcmc_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(SGMCMC_ENODEV, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    return n;
}

// header + one record per block of the smallest block size (64 lanes)
size_t sgmcmc_step_stats_workspace_bytes(size_t n) { return (max_grid_for(n) + 1) * 4 * sizeof(double); }
size_t sgmcmc_step_stats_records(size_t n, const sgmcmc_launch_t *launch_in)
{
    LaunchCfg cfg;
    if (!launch_in || resolve_launch(launch_in, cfg) != 0) return 0;
    if (cfg.block_threads <= 0) { fail(SGMCMC_EINVAL, "step_stats_records: launch.block_threads must be explicit"); return 0; }
    const size_t want = want_blocks(n, cfg.block_threads, cfg.qpt);
    return want < (size_t)cfg.max_blocks ? want : (size_t)cfg.max_blocks;
}
int sgmcmc_step_stats_finish(const void *stats_ws, double *stats_out, sgmcmc_stream_t stream)
{
    if (!stats_ws || !stats_out) return fail(SGMCMC_EINVAL, "step_stats_finish: NULL argument");
    hipLaunchKernelGGL(stats_final_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream),
                       static_cast<const double *>(stats_ws), stats_out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "launch stats_final");
}

int sgmcmc_philox_normal_f32(float *out, size_t n, uint64_t seed, uint64_t step, const uint64_t *step_dev,
                             const sgmcmc_launch_t *launch_cfg, sgmcmc_stream_t stream)
{
    if (n == 0) return 0;
    if (!out) return fail(SGMCMC_EINVAL, "philox_normal: out is NULL");
    NormalFillOp<float> op{out, make_key(seed, step, step_dev)};
    return launch(op, n, aligned16(out), sizeof(*out), launch_cfg, static_cast<hipStream_t>(stream));
}
int sgmcmc_philox_normal_f64(double *out, size_t n, uint64_t seed, uint64_t step, const uint64_t *step_dev,
                             const sgmcmc_launch_t *launch_cfg, sgmcmc_stream_t stream)
{
    if (n == 0) return 0;
    if (!out) return fail(SGMCMC_EINVAL, "philox_normal: out is NULL");
    NormalFillOp<double> op{out, make_key(seed, step, step_dev)};
    return launch(op, n, aligned16(out), sizeof(*out), launch_cfg, static_cast<hipStream_t>(stream));
}
int sgmcmc_philox_bits_u32(uint32_t *out, size_t n, uint64_t seed, uint64_t step, const uint64_t *step_dev, sgmcmc_stream_t stream)
{
    if (n == 0) return 0;
    if (!out) return fail(SGMCMC_EINVAL, "philox_bits: out is NULL");
    hipLaunchKernelGGL(philox_bits_kernel, dim3(small_grid((n + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       out, n, make_key(seed, step, step_dev));
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "launch philox_bits");
}

int sgmcmc_moments_update_f32(const float *theta, float *mean, float *m2, size_t n, uint64_t count,
                              const sgmcmc_launch_t *launch_cfg, sgmcmc_stream_t stream)
{
    if (n == 0) return 0;
    if (!theta || !mean || !m2 || count == 0) return fail(SGMCMC_EINVAL, "moments_update: NULL argument or count == 0");
    MomentsOp<float> op{theta, mean, m2, 1.0f / (float)count};
    return launch(op, n, aligned16(theta) && aligned16(mean) && aligned16(m2), 5 * sizeof(*theta), launch_cfg, static_cast<hipStream_t>(stream));
}
int sgmcmc_moments_update_f64(const double *theta, double *mean, double *m2, size_t n, uint64_t count,
                              const sgmcmc_launch_t *launch_cfg, sgmcmc_stream_t stream)
{
    if (n == 0) return 0;
    if (!theta || !mean || !m2 || count == 0) return fail(SGMCMC_EINVAL, "moments_update: NULL argument or count == 0");
    MomentsOp<double> op{theta, mean, m2, 1.0 / (double)count};
    return launch(op, n, aligned16(theta) && aligned16(mean) && aligned16(m2), 5 * sizeof(*theta), launch_cfg, static_cast<hipStream_t>(stream));
}

#define SGMCMC_RHAT(SFX, T)                                                                                           \
    int sgmcmc_rhat_pack_##SFX(const T *mean, const T *m2, size_t n, uint64_t count, size_t n_shards, size_t shard_len, \
                               T *out3, sgmcmc_stream_t stream)                                                       \
    {                                                                                                                 \
        if (n == 0) return 0;                                                                                         \
        if (!mean || !m2 || !out3 || count < 2) return fail(SGMCMC_EINVAL, "rhat_pack: NULL argument or count < 2"); \
        if (n_shards == 0 || shard_len == 0 || n_shards * shard_len < n || (n_shards - 1) * shard_len >= n)           \
            return fail(SGMCMC_EINVAL, "rhat_pack: n_shards * shard_len must cover n with no empty shard");          \
        const size_t total = n_shards * shard_len;                                                                    \
        hipLaunchKernelGGL((rhat_pack_kernel<T>), dim3(small_grid(total)), dim3(256), 0, static_cast<hipStream_t>(stream), \
                           mean, m2, n, T(1) / (T)(count - 1), shard_len, total, out3);                               \
        hipError_t e = hipGetLastError();                                                                             \
        return e == hipSuccess ? 0 : hip_fail(e, "launch rhat_pack");                                                 \
    }                                                                                                                 \
    int sgmcmc_rhat_finish_##SFX(const T *sum3, size_t n, size_t ld, int m_chains, uint64_t count, T *rhat,          \
                                 double *summary_out4, void *summary_ws, sgmcmc_stream_t stream)                      \
    {                                                                                                                 \
        if (n == 0) return 0;                                                                                         \
        if (!sum3 || !rhat || m_chains < 2 || count < 2 || ld < n)                                                    \
            return fail(SGMCMC_EINVAL, "rhat_finish: NULL argument, m_chains < 2, count < 2 or ld < n");              \
        if ((summary_out4 == nullptr) != (summary_ws == nullptr))                                                     \
            return fail(SGMCMC_EINVAL, "rhat_finish: summary_out4 and summary_ws go together");                       \
        hipLaunchKernelGGL((rhat_finish_kernel<T>), dim3(small_grid(n)), dim3(256), 0, static_cast<hipStream_t>(stream), \
                           sum3, n, ld, (T)m_chains, (T)count, rhat);                                                 \
        hipError_t e = hipGetLastError();                                                                             \
        if (e != hipSuccess) return hip_fail(e, "launch rhat_finish");                                                \
        /* device-side summary {sum, sum of squares, min, max} of R-hat: no host synchronisation on the path */      \
        return summary_out4 ? summary<T>(rhat, n, summary_out4, summary_ws, static_cast<hipStream_t>(stream)) : 0;   \
    }
SGMCMC_RHAT(f32, float)
SGMCMC_RHAT(f64, double)
#undef SGMCMC_RHAT

int sgmcmc_counter_add_u64(uint64_t *counter, uint64_t inc, sgmcmc_stream_t stream)
{
    if (!counter) return fail(SGMCMC_EINVAL, "counter_add: counter is NULL");
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), counter, inc);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "launch counter_add");
}

#define SGMCMC_WINDOW_GATHER(SFX, T)                                                                                  \
    int sgmcmc_window_gather_##SFX(const T *X, const T *y, size_t n_data, size_t start, size_t batch, size_t dim,   \
                                   T *x_out, size_t x_out_ld, T *y_out, sgmcmc_stream_t stream)                      \
    {                                                                                                                \
        if (!X || !y || !x_out || !y_out) return fail(SGMCMC_EINVAL, "window_gather: NULL argument");               \
        if (batch == 0 || start + batch > n_data) return fail(SGMCMC_EINVAL, "window_gather: window outside the data"); \
        if (x_out_ld < dim) return fail(SGMCMC_EINVAL, "window_gather: x_out_ld < dim");                             \
        const size_t total = (batch * dim) / (16 / sizeof(T)) + batch;    /* 16-byte copies: see the kernel */            \
        const size_t blocks = (total + 255) / 256;                                                                   \
        hipLaunchKernelGGL((window_gather_kernel<T>), dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, \
                           static_cast<hipStream_t>(stream), X, y, start, batch, dim, x_out, x_out_ld, y_out);       \
        hipError_t e = hipGetLastError();                                                                            \
        return e == hipSuccess ? 0 : hip_fail(e, "launch window_gather");                                            \
    }
SGMCMC_WINDOW_GATHER(f32, float)
SGMCMC_WINDOW_GATHER(f64, double)
#undef SGMCMC_WINDOW_GATHER

#define SGMCMC_TANH_ROWDOT(SFX, T)                                                                                   \
    int sgmcmc_bias_tanh_rowdot_##SFX(T *a, const T *bias, const T *w, size_t rows, size_t cols, T *out,             \
                                      const void *stats_ws, double *tsq_parts, sgmcmc_stream_t stream)               \
    {                                                                                                                \
        if (rows == 0 || cols == 0) return 0;                                                                        \
        if (!a || !w || !out) return fail(SGMCMC_EINVAL, "tanh_rowdot: NULL argument");                             \
        if ((stats_ws == nullptr) != (tsq_parts == nullptr))                                                         \
            return fail(SGMCMC_EINVAL, "tanh_rowdot: stats_ws and tsq_parts go together");                           \
        if (rows > 0x7fffffffull) return fail(SGMCMC_EINVAL, "tanh_rowdot: too many rows");                          \
        hipLaunchKernelGGL((tanh_rowdot_kernel<T>), dim3((unsigned)rows), dim3(256), 0, static_cast<hipStream_t>(stream), \
                           a, w, cols, out, static_cast<const double *>(stats_ws), tsq_parts, bias);                 \
        hipError_t e = hipGetLastError();                                                                            \
        return e == hipSuccess ? 0 : hip_fail(e, "launch tanh_rowdot");                                              \
    }                                                                                                                \
    int sgmcmc_bias_tanh_##SFX(T *a, const T *bias, size_t rows, size_t cols, sgmcmc_stream_t stream)                \
    {                                                                                                                \
        if (rows == 0 || cols == 0) return 0;                                                                        \
        if (!a || !bias) return fail(SGMCMC_EINVAL, "bias_tanh: NULL argument");                                     \
        if (rows * cols >= 0xffffffffull) return fail(SGMCMC_EINVAL, "bias_tanh: more than 2^32 - 1 elements");      \
        const size_t lanes = (rows * cols + 3) / 4, blocks = (lanes + 255) / 256;                                    \
        hipLaunchKernelGGL((bias_tanh_kernel<T>), dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0,   \
                           static_cast<hipStream_t>(stream), a, bias, (unsigned)rows, (unsigned)cols);               \
        hipError_t e = hipGetLastError();                                                                            \
        return e == hipSuccess ? 0 : hip_fail(e, "launch bias_tanh");                                                \
    }                                                                                                                \
    int sgmcmc_bnn_head_last_layer_backward_##SFX(                                                                   \
        const T *mean, size_t n_mean_parts, const T *y, const T *log_var, const double *tsq_parts, const T *last_bias, \
        size_t rows, size_t cols, double batch_size, double n_examples, double n_params, double wdecay, double prior_mean,        \
        double prior_var, int fold_prior_grad, const T *w, const T *h, const T *bias_prev, T beta, T *cost_out,      \
        T *grad_log_var_out, T *grad_last_bias_out, T *mse_out, T *delta_prev, T *colsum, T *gw,                     \
        sgmcmc_stream_t stream)                                                                                      \
    {                                                                                                                \
        return head_last_layer_backward_impl<T>(mean, n_mean_parts, y, log_var, tsq_parts, last_bias, rows, cols, batch_size, \
                                                n_examples, n_params, wdecay, prior_mean, prior_var, fold_prior_grad, \
                                                w, h, bias_prev, beta, cost_out, grad_log_var_out,                   \
                                                grad_last_bias_out, mse_out, delta_prev, colsum, gw,                 \
                                                static_cast<hipStream_t>(stream));                                   \
    }
SGMCMC_TANH_ROWDOT(f32, float)
SGMCMC_TANH_ROWDOT(f64, double)
#undef SGMCMC_TANH_ROWDOT

size_t sgmcmc_summary_workspace_bytes(void) { return sizeof(Summary) * SUMMARY_BLOCKS; }
int sgmcmc_summary_f32(const float *x, size_t n, double *out4, void *workspace, sgmcmc_stream_t stream)
{
    return summary<float>(x, n, out4, workspace, static_cast<hipStream_t>(stream));
}
int sgmcmc_summary_f64(const double *x, size_t n, double *out4, void *workspace, sgmcmc_stream_t stream)
{
    return summary<double>(x, n, out4, workspace, static_cast<hipStream_t>(stream));
}

int sgmcmc_bnn_head_f32(const float *mean, const float *y, const float *log_var, const double *theta_sumsq,
                        const void *stats_ws, const float *last_bias, size_t B,
                        double batch_size, double n_examples, double n_params, double wdecay, double prior_mean,
                        double prior_var, int fold_prior_grad, float *delta, float *cost_out, float *grad_log_var_out,
                        float *grad_last_bias_out, float *mse_out, sgmcmc_stream_t stream)
{
    return bnn_head<float>(mean, y, log_var, theta_sumsq, stats_ws, last_bias, B, batch_size, n_examples, n_params, wdecay, prior_mean,
                           prior_var, fold_prior_grad, delta, cost_out, grad_log_var_out, grad_last_bias_out, mse_out,
                           static_cast<hipStream_t>(stream));
}
int sgmcmc_bnn_head_f64(const double *mean, const double *y, const double *log_var, const double *theta_sumsq,
                        const void *stats_ws, const double *last_bias, size_t B,
                        double batch_size, double n_examples, double n_params, double wdecay, double prior_mean,
                        double prior_var, int fold_prior_grad, double *delta, double *cost_out, double *grad_log_var_out,
                        double *grad_last_bias_out, double *mse_out, sgmcmc_stream_t stream)
{
    return bnn_head<double>(mean, y, log_var, theta_sumsq, stats_ws, last_bias, B, batch_size, n_examples, n_params, wdecay, prior_mean,
                            prior_var, fold_prior_grad, delta, cost_out, grad_log_var_out, grad_last_bias_out, mse_out,
                           static_cast<hipStream_t>(stream));
}
int sgmcmc_tanh_backward_f32(float *delta, const float *h, size_t n, sgmcmc_stream_t stream)
{
    if (n == 0) return 0;
    if (!delta || !h) return fail(SGMCMC_EINVAL, "tanh_backward: NULL argument");
    hipLaunchKernelGGL((tanh_backward_kernel<float>), dim3(small_grid(n)), dim3(256), 0, static_cast<hipStream_t>(stream), delta, h, n);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "launch tanh_backward");
}
int sgmcmc_tanh_backward_colsum_f32(float *delta, const float *h, size_t rows, size_t cols, const float *bias, float beta,
                                    float *colsum, sgmcmc_stream_t stream)
{
    return tanh_backward_colsum_impl<float>(delta, h, rows, cols, bias, beta, colsum, static_cast<hipStream_t>(stream));
}
int sgmcmc_tanh_backward_colsum_f64(double *delta, const double *h, size_t rows, size_t cols, const double *bias,
                                    double beta, double *colsum, sgmcmc_stream_t stream)
{
    return tanh_backward_colsum_impl<double>(delta, h, rows, cols, bias, beta, colsum, static_cast<hipStream_t>(stream));
}
int sgmcmc_bnn_last_layer_backward_f32(const float *dvec, const float *w, const float *h, size_t rows, size_t cols,
                                       const float *bias_prev, float beta, float *delta_prev, float *colsum, float *gw,
                                       sgmcmc_stream_t stream)
{
    return last_layer_backward_impl<float>(dvec, w, h, rows, cols, bias_prev, beta, delta_prev, colsum, gw,
                                           static_cast<hipStream_t>(stream));
}
int sgmcmc_bnn_last_layer_backward_f64(const double *dvec, const double *w, const double *h, size_t rows, size_t cols,
                                       const double *bias_prev, double beta, double *delta_prev, double *colsum,
                                       double *gw, sgmcmc_stream_t stream)
{
    return last_layer_backward_impl<double>(dvec, w, h, rows, cols, bias_prev, beta, delta_prev, colsum, gw,
                                            static_cast<hipStream_t>(stream));
}
int sgmcmc_tanh_backward_f64(double *delta, const double *h, size_t n, sgmcmc_stream_t stream)
{
    if (n == 0) return 0;
    if (!delta || !h) return fail(SGMCMC_EINVAL, "tanh_backward: NULL argument");
    hipLaunchKernelGGL((tanh_backward_kernel<double>), dim3(small_grid(n)), dim3(256), 0, static_cast<hipStream_t>(stream), delta, h, n);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "launch tanh_backward");
}

}  // extern "C"
