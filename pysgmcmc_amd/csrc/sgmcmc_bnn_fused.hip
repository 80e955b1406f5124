// sgmcmc_bnn_fused.hip -- whole SGHMC steps of a SMALL tanh-MLP BNN in one kernel.
//
// BASELINE.json configs[1] (the reference's default model: 3x50 tanh net, 5 252 parameters,
// minibatch 20) is launch-bound on any GPU: one `next(sampler)` is ~20 launches of ~3-5 us even
// when replayed from a hipGraph. This kernel runs `n_steps` COMPLETE steps -- minibatch window,
// forward, loss head (pysgmcmc/models/bayesian_neural_network.py:365-388), analytic backward into
// the gradient row, fused SGHMC update (pysgmcmc/samplers/sghmc.py:165-251, the same quad operator
// and the same Philox stream as kernel K1), sum(theta^2) for the next cost -- with ONE workgroup of
// 512 lanes per chain and `__syncthreads()` between phases. Activations, deltas and a copy of the
// parameters live in LDS (the dot-product loops never wait on global memory); the sampler state stays
// in its arena rows (L2-resident at this size) and is streamed once per step by the update phase, which
// also drops theta' straight into the LDS copy for the next step.
// Measured (MI355X, 3x50 net, batch 20): 20.0 us per step (round 5; 27.2 before) = forward 5.2 + weight / bias gradients 3.2 +
// delta products 4.5 + update 3.2 + head / sums / barriers 3.9, bound by the instruction and LDS latency of ONE workgroup on one CU,
// not by memory: 256 chains in one launch take the same time per step. What round 5 changed: 512 lanes instead of 1024 (the
// 128-register cap of a 1024-lane workgroup spilled: 24.3 -> 20.4 us), pairs of adjacent outputs per lane with 8-byte LDS
// accesses (2 x 2 register tiles for the weight gradients, the bias gradients riding in the same loop), theta' written to the
// LDS copy by the update (no reload through L2), the next step's minibatch window requested a step ahead. Every output keeps
// its k-ordered fma chain, so the results are bit-identical to the scalar loops. (Tried and dropped: the update reading theta and
// the gradient from LDS copies through flat accesses -- 20.4 us for one chain, 256 chains 32 -> 37 us per step.) blockIdx.x is
// the chain: independent chains (seed = seed_base + chain, own state rows, own window stream) run
// concurrently on other CUs at no extra cost.
//
// The update arithmetic is the shared SghmcOp (bit-identical to K1 given the same gradient); the
// matrix products are plain fp32/fp64 dot products in k order, so a fused chain tracks the
// GEMM-based path to rounding (tests: 2e-4 relative over 12 steps, like the GEMM path itself
// against the fp64 golden trajectory).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "sgmcmc_hip.h"

#pragma clang fp contract(off)

#include "sgmcmc_device.hpp"
#include "sgmcmc_host.hpp"

using namespace sgmcmc_host;

namespace {

constexpr int FUSED_MAX_LAYERS = 8;
constexpr int FUSED_THREADS = 512;       // 8 waves: 2 per SIMD, 256 registers each (1024 lanes spilled registers: 24.3 vs 20.4 us per step)

__device__ __forceinline__ float tanh_t(float x) { return tanhf(x); }
__device__ __forceinline__ double tanh_t(double x) { return tanh(x); }
// dot products accumulate with fused multiply-add (a matrix product has no reference rounding order;
// GEMM libraries fuse too); the UPDATE arithmetic below stays one rounding per reference op
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }

// pairs of consecutive elements as ONE LDS access (ds_read_b64 / b128): half the LDS instructions of the dot-product loops
template <typename T> struct Pair;
template <> struct Pair<float> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct Pair<double> { typedef double type __attribute__((ext_vector_type(2))); };
template <typename T>
__device__ __forceinline__ typename Pair<T>::type ld2(const T *p) { return *reinterpret_cast<const typename Pair<T>::type *>(p); }
template <typename T>
__device__ __forceinline__ void st2(T *p, typename Pair<T>::type v) { *reinterpret_cast<typename Pair<T>::type *>(p) = v; }

template <typename T>
struct FusedArgs {
    T *theta, *V, *grad, *tau, *g, *vh, *minv;          // chain c at + c * chain_stride
    size_t n_params, chain_stride;
    int n_layers;                                        // number of weight layers L
    int sizes[FUSED_MAX_LAYERS + 1];                     // sizes[0] = inputs, sizes[L] = 1
    size_t off_w[FUSED_MAX_LAYERS + 1], off_b[FUSED_MAX_LAYERS + 1];   // parameter offsets of layer l (1-based)
    size_t act_off[FUSED_MAX_LAYERS + 1], del_off[FUSED_MAX_LAYERS + 1];  // LDS element offsets
    size_t lds_y;                                        // LDS element offset of the target window
    size_t lds_w;                                        // LDS element offset of the parameter copy
    const T *X, *y;
    size_t n_data;
    const int *starts;                                   // [n_chains][n_steps]
    int batch;
    double batch_size, n_examples, wp_den, lvp_den, ln_prior_mean, ln_prior_var, wdecay;
    T eps_e2, c1, c3, e4, mdecay, grad_decay;            // host-derived scalars of K1
    T sgld_eps, sgld_A, sgld_a_eff, sgld_two_eps, sgld_den;   // host-derived scalars of K2 (SGLD chains)
    uint64_t first_step, n_steps, burn_in_steps, seed_base;
    const T *xi;                                         // nullable: [n_steps][n_params], chain 0
    T *cost_out;                                         // [n_chains][n_steps]
};

// block-wide sum of one double per lane; every lane returns the total. red: 17 doubles of LDS.
__device__ __forceinline__ double block_sum(double v, double *red)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_sum_dpp_lane63(v);
    __syncthreads();                                      // red may still be read from the previous call
    if (lane == 63) red[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
        red[16] = t;
    }
    __syncthreads();
    return red[16];
}

// wl: the LDS copy of the parameters the next step's forward pass reads -- theta' goes there straight from the registers, so the
// next step does not wait for a round trip through L2 to get it back
template <typename Op>
__device__ __forceinline__ double run_update(Op &op, size_t n_params, typename Op::real *wl)
{
    const size_t nq_full = n_params / 4;
    const int tail = (int)(n_params % 4);
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (size_t q = threadIdx.x; q < nq_full; q += blockDim.x) {
        typename Op::Regs R;
        op.template load_vec<false>(q, R);
        op.compute(q, R);
        op.template store_vec<false>(q, R);
#pragma unroll
        for (int j = 0; j < 4; ++j) wl[4 * q + j] = R.th[j];
        op.template accumulate<true>(R, 4, acc);
    }
    if (tail && threadIdx.x == blockDim.x - 1) {
        typename Op::Regs R;
        op.load_part_(nq_full, tail, R);
        op.compute(nq_full, R);
        op.store_part_(nq_full, tail, R);
        for (int j = 0; j < tail; ++j) wl[4 * nq_full + j] = R.th[j];
        op.template accumulate<true>(R, tail, acc);
    }
    return acc[0];                                        // this lane's share of sum(theta'^2)
}

// KIND 0: SGHMC (K1's operator, sghmc.py:165-251); KIND 1: preconditioned SGLD (K2's operator, sgld.py:149-211)
template <typename T, int KIND, bool ADAPT, bool INJECT>
__device__ __forceinline__ double update_phase(const FusedArgs<T> &a, T *theta, T *V, const T *grad, T *tau, T *g, T *vh,
                                               T *minv, const T *xi, uint64_t seed, uint64_t step, T *wl)
{
    NoiseKey nk;
    nk.k0 = (uint32_t)seed; nk.k1 = (uint32_t)(seed >> 32);
    nk.s0 = (uint32_t)step; nk.s1 = (uint32_t)(step >> 32);
    nk.step_dev = nullptr;
    if (KIND == 0) {
        SghmcOp<T, ADAPT, INJECT> op{theta, V, grad, tau, g, vh, minv, nullptr, xi,
                                     a.eps_e2, a.c1, a.c3, a.e4, a.mdecay, a.grad_decay, nk, nullptr};
        return run_update(op, a.n_params, wl);
    } else {
        SgldOp<T, ADAPT, INJECT> op{theta, grad, tau, g, vh, minv, nullptr, xi, a.sgld_eps, a.sgld_A, a.sgld_a_eff,
                                    a.sgld_two_eps, a.sgld_den, a.grad_decay, nk, nullptr};
        return run_update(op, a.n_params, wl);
    }
}

template <typename T, int KIND>
__global__ void __launch_bounds__(FUSED_THREADS) bnn_fused_sghmc_kernel(const FusedArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double *red = reinterpret_cast<double *>(smem_raw);           // 17 doubles (+ pad to 160 B)
    T *lds = reinterpret_cast<T *>(smem_raw + 160);
    const int chain = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const size_t cs = (size_t)chain * a.chain_stride;
    T *theta = a.theta + cs, *V = KIND == 0 ? a.V + cs : nullptr, *grad = a.grad + cs;
    T *tau = a.tau + cs, *g = a.g + cs, *vh = a.vh + cs, *minv = a.minv + cs;
    const int L = a.n_layers, B = a.batch;
    const uint64_t seed = a.seed_base + (uint64_t)chain;
    T *yb = lds + a.lds_y;
    T *wl = lds + a.lds_w;                                // this step's parameters

    // sum(theta^2) of the starting point (weight-prior value of the first cost)
    double part = 0.0;
    for (size_t i = tid; i < a.n_params; i += nt) { double v = (double)theta[i]; part += v * v; }
    double tsq = block_sum(part, red);

    // The next step's minibatch window is requested a whole step ahead (its start index, then one window element per lane, into
    // registers): the dependent pair of global loads is off the step's critical path. Windows larger than the workgroup are
    // loaded in place as before.
    const int D0 = a.sizes[0];
    const bool prefetch = (size_t)B * D0 <= (size_t)nt;
    T x_next = T(0), y_next = T(0);
    auto fetch_window = [&](uint64_t tt) {
        const size_t st = (size_t)a.starts[(size_t)chain * a.n_steps + tt];
        if (tid < B * D0) x_next = a.X[st * D0 + tid];
        if (tid < B) y_next = a.y[st + tid];
    };
    if (prefetch) fetch_window(0);
    for (uint64_t t = 0; t < a.n_steps; ++t) {
        const uint64_t step = a.first_step + t;
        // ---- parameters into LDS (first step of the launch: later ones find theta' there, written by the update phase);
        // minibatch window [start, start + B) (pysgmcmc/data_batches.py:118-123)
        {
            if (t == 0) {
#pragma unroll 4
                for (size_t i = tid; i < a.n_params; i += nt) wl[i] = theta[i];
            }
            T *x0 = lds + a.act_off[0];
            if (prefetch) {
                if (tid < B * D0) x0[tid] = x_next;
                if (tid < B) yb[tid] = y_next;
                if (t + 1 < a.n_steps) fetch_window(t + 1);
            } else {
                const size_t start = (size_t)a.starts[(size_t)chain * a.n_steps + t];
                for (int i = tid; i < B * D0; i += nt) x0[i] = a.X[start * D0 + i];
                for (int i = tid; i < B; i += nt) yb[i] = a.y[start + i];
            }
        }
        __syncthreads();
        // ---- forward
        for (int l = 1; l <= L; ++l) {
            const int nin = a.sizes[l - 1], nout = a.sizes[l];
            const T *W = wl + a.off_w[l], *bias = wl + a.off_b[l];
            const T *hin = lds + a.act_off[l - 1];
            T *hout = lds + a.act_off[l];
            // two adjacent outputs per lane where the layout allows pair accesses (even width, even offsets): the weight row and
            // the bias come as pairs; every output keeps its own k-ordered fma chain (same bits as the scalar form)
            const bool pairs = (nout % 2 == 0) && ((a.off_w[l] | a.off_b[l] | a.act_off[l] | a.lds_w) % 2 == 0);
            if (pairs) {
                const int half = nout / 2;
                for (int idx = tid; idx < B * half; idx += nt) {
                    const int b = idx / half, j = 2 * (idx - b * half);
                    typename Pair<T>::type acc = ld2(bias + j);
#pragma unroll 8
                    for (int k = 0; k < nin; ++k) {
                        const T h = hin[b * nin + k];
                        const typename Pair<T>::type w = ld2(W + (size_t)k * nout + j);
                        acc.x = fma_t(h, w.x, acc.x);
                        acc.y = fma_t(h, w.y, acc.y);
                    }
                    if (l < L) { acc.x = tanh_t(acc.x); acc.y = tanh_t(acc.y); }
                    st2(hout + b * nout + j, acc);
                }
            } else {
                for (int idx = tid; idx < B * nout; idx += nt) {
                    const int b = idx / nout, j = idx - b * nout;
                    T acc = bias[j];
#pragma unroll 8
                    for (int k = 0; k < nin; ++k) acc = fma_t(hin[b * nin + k], W[(size_t)k * nout + j], acc);
                    hout[idx] = (l < L) ? tanh_t(acc) : acc;
                }
            }
            __syncthreads();
        }
        // ---- loss head (bayesian_neural_network.py:365-388)
        const double s = (double)wl[a.n_params - 1];
        const double es = exp(s), inv = 1.0 / (es + 1e-16), dscale = -(inv / a.batch_size);
        {
            const T *mean = lds + a.act_off[L];
            T *dL = lds + a.del_off[L];
            double sse = 0.0;
            for (int i = tid; i < B; i += nt) {
                double r = (double)yb[i] - (double)mean[i];
                sse += r * r;
                dL[i] = (T)(r * dscale);
            }
            const double tot = block_sum(sse, red);       // ends with a barrier: dL is visible
            if (tid == 0) {
                const double Bd = (double)B;
                double log_like = (-(tot * (0.5 * inv)) - 0.5 * s * Bd) / a.batch_size;
                double d = s - a.ln_prior_mean;
                double lvp = -(d * d) / a.lvp_den - 0.5 * a.ln_prior_var;
                double wp = (-0.5 * a.wdecay) * tsq / a.wp_den;
                a.cost_out[(size_t)chain * a.n_steps + t] = (T)(-(log_like + lvp / a.n_examples + wp / a.n_examples));
                // d NLL / d log_var; its weight-prior term is added by the update (grad_decay)
                grad[a.n_params - 1] = (T)(-((tot * (0.5 * es * inv * inv) - 0.5 * Bd) / a.batch_size
                                             + (-2.0 * d / a.lvp_den) / a.n_examples));
            }
        }
        // ---- backward
        for (int l = L; l >= 1; --l) {
            const int nin = a.sizes[l - 1], nout = a.sizes[l];
            const T *W = wl + a.off_w[l];
            const T *hin = lds + a.act_off[l - 1];
            const T *dl = lds + a.del_off[l];
            T *gW = grad + a.off_w[l], *gb = grad + a.off_b[l];
            // gW = h_{l-1}^T delta_l. 2 x 2 outputs per lane where the layout allows pair accesses: two pair loads feed four
            // independent b-ordered fma chains (same bits as the scalar form, a quarter of its LDS instructions)
            const bool pj = (nout % 2 == 0) && ((a.off_w[l] | a.del_off[l]) % 2 == 0);
            const bool pk = (nin % 2 == 0) && (a.act_off[l - 1] % 2 == 0);
            if (pj && pk && a.off_b[l] % 2 == 0) {
                const int hj = nout / 2, items = (nin / 2) * hj;
                for (int idx = tid; idx < items; idx += nt) {
                    const int k = 2 * (idx / hj), j = 2 * (idx % hj);
                    typename Pair<T>::type a0 = {T(0), T(0)}, a1 = {T(0), T(0)}, sb = {T(0), T(0)};
#pragma unroll 4
                    for (int b = 0; b < B; ++b) {
                        const typename Pair<T>::type h = ld2(hin + b * nin + k), d = ld2(dl + b * nout + j);
                        a0.x = fma_t(h.x, d.x, a0.x); a0.y = fma_t(h.x, d.y, a0.y);
                        a1.x = fma_t(h.y, d.x, a1.x); a1.y = fma_t(h.y, d.y, a1.y);
                        sb.x += d.x; sb.y += d.y;                 // gb = delta_l^T 1 rides along (b order, as the loop below)
                    }
                    st2(gW + (size_t)k * nout + j, a0);
                    st2(gW + (size_t)(k + 1) * nout + j, a1);
                    if (k == 0) st2(gb + j, sb);                  // (the lanes of the first row pair keep it)
                }
            } else {
                for (int idx = tid; idx < nin * nout; idx += nt) {
                    const int k = idx / nout, j = idx - k * nout;
                    T acc = T(0);
#pragma unroll 8
                    for (int b = 0; b < B; ++b) acc = fma_t(hin[b * nin + k], dl[b * nout + j], acc);
                    gW[idx] = acc;
                }
            }
            if (!(pj && pk && a.off_b[l] % 2 == 0)) {
                for (int j = tid; j < nout; j += nt) {                    // gb = delta_l^T 1
                    T acc = T(0);
#pragma unroll 8
                    for (int b = 0; b < B; ++b) acc += dl[b * nout + j];
                    gb[j] = acc;
                }
            }
            if (l > 1) {                                                  // delta_{l-1} = (delta_l W^T) (1 - h^2)
                T *dprev = lds + a.del_off[l - 1];
                if (pj && pk && a.del_off[l - 1] % 2 == 0) {
                    // two adjacent k per lane, j in pairs: three pair loads per four fmas; each output keeps its j-ordered chain
                    const int hk = nin / 2;
                    for (int idx = tid; idx < B * hk; idx += nt) {
                        const int b = idx / hk, k = 2 * (idx - b * hk);
                        T acc0 = T(0), acc1 = T(0);
#pragma unroll 4
                        for (int j = 0; j < nout; j += 2) {
                            const typename Pair<T>::type d = ld2(dl + b * nout + j);
                            const typename Pair<T>::type w0 = ld2(W + (size_t)k * nout + j), w1 = ld2(W + (size_t)(k + 1) * nout + j);
                            acc0 = fma_t(d.x, w0.x, acc0); acc0 = fma_t(d.y, w0.y, acc0);
                            acc1 = fma_t(d.x, w1.x, acc1); acc1 = fma_t(d.y, w1.y, acc1);
                        }
                        const typename Pair<T>::type hv = ld2(hin + b * nin + k);
                        typename Pair<T>::type out = {acc0 * (T(1) - hv.x * hv.x), acc1 * (T(1) - hv.y * hv.y)};
                        st2(dprev + b * nin + k, out);
                    }
                } else {
                    for (int idx = tid; idx < B * nin; idx += nt) {
                        const int b = idx / nin, k = idx - b * nin;
                        T acc = T(0);
#pragma unroll 8
                        for (int j = 0; j < nout; ++j) acc = fma_t(dl[b * nout + j], W[(size_t)k * nout + j], acc);
                        const T hv = hin[idx];
                        dprev[idx] = acc * (T(1) - hv * hv);
                    }
                }
            }
            __syncthreads();
        }
        // ---- fused update (K1's or K2's operator) + sum(theta'^2)
        const bool adapt = step < a.burn_in_steps || a.burn_in_steps == 0;
        const T *xi = (a.xi != nullptr && chain == 0) ? a.xi + (size_t)t * a.n_params : nullptr;
        double share;
        if (adapt) {
            share = xi ? update_phase<T, KIND, true, true>(a, theta, V, grad, tau, g, vh, minv, xi, seed, step, wl)
                       : update_phase<T, KIND, true, false>(a, theta, V, grad, tau, g, vh, minv, xi, seed, step, wl);
        } else {
            share = xi ? update_phase<T, KIND, false, true>(a, theta, V, grad, tau, g, vh, minv, xi, seed, step, wl)
                       : update_phase<T, KIND, false, false>(a, theta, V, grad, tau, g, vh, minv, xi, seed, step, wl);
        }
        __threadfence_block();
        tsq = block_sum(share, red);                      // barriers inside: the new theta is visible to the block
    }
}

template <typename T, int KIND>
int bnn_fused_steps(T *theta, T *V, T *grad, T *tau, T *g, T *v_hat, T *minv, size_t n_params, size_t chain_stride,
                    int n_chains, const int *layer_sizes, int n_layers, const T *X, const T *y, size_t n_data,
                    const int *window_starts, int batch, double batch_size, double n_examples, double wdecay,
                    double prior_mean, double prior_var, T eps, T scale_grad, T mdecay /* SGLD: A */, uint64_t first_step,
                    uint64_t n_steps, uint64_t burn_in_steps, uint64_t seed_base, const T *xi, T *cost_out, hipStream_t st)
{
    if (n_steps == 0 || n_chains == 0) return 0;
    if (!theta || (KIND == 0 && !V) || !grad || !tau || !g || !v_hat || !minv || !layer_sizes || !X || !y || !window_starts ||
        !cost_out)
        return fail(SGMCMC_EINVAL, "bnn_fused_sghmc_steps: NULL argument");
    if (n_layers < 1 || n_layers > FUSED_MAX_LAYERS) return fail(SGMCMC_EINVAL, "bnn_fused_sghmc_steps: 1..8 layers");
    if (layer_sizes[n_layers] != 1) return fail(SGMCMC_EINVAL, "bnn_fused_sghmc_steps: the last layer must have one unit");
    if (batch < 1 || (size_t)batch > n_data) return fail(SGMCMC_EINVAL, "bnn_fused_sghmc_steps: bad batch");
    if (xi && (n_params % 4) != 0) return fail(SGMCMC_EINVAL, "bnn_fused_sghmc_steps: injected xi needs n_params %% 4 == 0");
    if (n_chains > 1 && (chain_stride < n_params || (chain_stride % 4) != 0))
        return fail(SGMCMC_EINVAL, "bnn_fused_sghmc_steps: chain_stride must be >= n_params and a multiple of 4");
    T *rows[7] = {theta, KIND == 0 ? V : theta, grad, tau, g, v_hat, minv};
    for (T *p : rows)
        if (reinterpret_cast<uintptr_t>(p) & 15u) return fail(SGMCMC_EINVAL, "bnn_fused_sghmc_steps: rows must be 16-B aligned");
    FusedArgs<T> a;
    a.theta = theta; a.V = V; a.grad = grad; a.tau = tau; a.g = g; a.vh = v_hat; a.minv = minv;
    a.n_params = n_params; a.chain_stride = chain_stride; a.n_layers = n_layers;
    size_t off = 0, lds_elems = 0;
    for (int l = 0; l <= n_layers; ++l) {
        if (layer_sizes[l] < 1) return fail(SGMCMC_EINVAL, "bnn_fused_sghmc_steps: bad layer size");
        a.sizes[l] = layer_sizes[l];
    }
    for (int l = 1; l <= n_layers; ++l) {                  // parameter order: W1, b1, ..., WL, bL, log_var
        a.off_w[l] = off; off += (size_t)a.sizes[l - 1] * a.sizes[l];
        a.off_b[l] = off; off += (size_t)a.sizes[l];
    }
    if (off + 1 != n_params) return fail(SGMCMC_EINVAL, "bnn_fused_sghmc_steps: n_params does not match the layer sizes");
    for (int l = 0; l <= n_layers; ++l) { a.act_off[l] = lds_elems; lds_elems += (size_t)batch * a.sizes[l]; }
    a.del_off[0] = 0;
    for (int l = 1; l <= n_layers; ++l) { a.del_off[l] = lds_elems; lds_elems += (size_t)batch * a.sizes[l]; }
    a.lds_y = lds_elems; lds_elems += (size_t)batch;
    lds_elems = (lds_elems + 3) & ~(size_t)3;
    a.lds_w = lds_elems; lds_elems += n_params;
    const size_t lds_bytes = 160 + lds_elems * sizeof(T);
    if (lds_bytes > 160 * 1024) return fail(SGMCMC_EINVAL, "bnn_fused_sghmc_steps: activations need %zu B of LDS (> 160 KiB); "
                                            "use the GEMM path", lds_bytes);
    a.X = X; a.y = y; a.n_data = n_data; a.starts = window_starts; a.batch = batch;
    a.batch_size = batch_size; a.n_examples = n_examples; a.wdecay = wdecay;
    a.wp_den = (double)n_params + (2.0 * 1e-16 + 1e-16);
    a.lvp_den = 2.0 * prior_var + (2.0 * 1e-16 + 1e-16);
    a.ln_prior_mean = std::log(prior_mean); a.ln_prior_var = std::log(prior_var);
    // K1's host-derived scalars (sghmc.py:111-117,211-217,235), same op order
    T eps_s = eps / std::sqrt(scale_grad);
    a.eps_e2 = std::pow(eps, T(2));
    a.c1 = (T(2) * std::pow(eps_s, T(2))) * mdecay;
    a.c3 = T(2) * std::pow(eps_s, T(3));
    a.e4 = std::pow(eps_s, T(4));
    a.mdecay = mdecay;
    // K2's host-derived scalars (sgld.py:106-108,186-191), same op order as sgld_step in sgmcmc_kernels.hip
    {
        const T A = mdecay;                               // the SGLD entry passes A in this slot
        const T sgn = (scale_grad > T(0)) ? T(1) : ((scale_grad < T(0)) ? T(-1) : T(0));
        a.sgld_den = scale_grad + ((T(2) * sgn) * T(1e-16) + T(1e-16));
        a.sgld_two_eps = T(2) * eps;
        a.sgld_a_eff = A - T(0);
        a.sgld_eps = eps;
        a.sgld_A = A;
    }
    a.grad_decay = (T)(wdecay / (a.wp_den * n_examples));   // weight-prior gradient, folded into the update
    a.first_step = first_step; a.n_steps = n_steps; a.burn_in_steps = burn_in_steps; a.seed_base = seed_base;
    a.xi = xi; a.cost_out = cost_out;
    if (lds_bytes > 64 * 1024) {
        hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void *>(&bnn_fused_sghmc_kernel<T, KIND>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e0 != hipSuccess) return hip_fail(e0, "hipFuncSetAttribute(bnn_fused_sghmc_kernel)");
    }
    hipLaunchKernelGGL((bnn_fused_sghmc_kernel<T, KIND>), dim3((unsigned)n_chains), dim3(FUSED_THREADS), lds_bytes, st, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "launch bnn_fused_sghmc_kernel");
}

}  // namespace

extern "C" {

int sgmcmc_bnn_fused_sghmc_steps_f32(float *theta, float *V, float *grad, float *tau, float *g, float *v_hat, float *minv,
                                     size_t n_params, size_t chain_stride, int n_chains, const int *layer_sizes,
                                     int n_layers, const float *X, const float *y, size_t n_data,
                                     const int *window_starts, int batch, double batch_size, double n_examples,
                                     double wdecay, double prior_mean, double prior_var, float eps, float scale_grad,
                                     float mdecay, uint64_t first_step, uint64_t n_steps, uint64_t burn_in_steps,
                                     uint64_t seed_base, const float *xi, float *cost_out, sgmcmc_stream_t stream)
{
    return bnn_fused_steps<float, 0>(theta, V, grad, tau, g, v_hat, minv, n_params, chain_stride, n_chains, layer_sizes,
                                  n_layers, X, y, n_data, window_starts, batch, batch_size, n_examples, wdecay,
                                  prior_mean, prior_var, eps, scale_grad, mdecay, first_step, n_steps, burn_in_steps,
                                  seed_base, xi, cost_out, static_cast<hipStream_t>(stream));
}
int sgmcmc_bnn_fused_sghmc_steps_f64(double *theta, double *V, double *grad, double *tau, double *g, double *v_hat,
                                     double *minv, size_t n_params, size_t chain_stride, int n_chains,
                                     const int *layer_sizes, int n_layers, const double *X, const double *y,
                                     size_t n_data, const int *window_starts, int batch, double batch_size,
                                     double n_examples, double wdecay, double prior_mean, double prior_var, double eps,
                                     double scale_grad, double mdecay, uint64_t first_step, uint64_t n_steps,
                                     uint64_t burn_in_steps, uint64_t seed_base, const double *xi, double *cost_out,
                                     sgmcmc_stream_t stream)
{
    return bnn_fused_steps<double, 0>(theta, V, grad, tau, g, v_hat, minv, n_params, chain_stride, n_chains, layer_sizes,
                                   n_layers, X, y, n_data, window_starts, batch, batch_size, n_examples, wdecay,
                                   prior_mean, prior_var, eps, scale_grad, mdecay, first_step, n_steps, burn_in_steps,
                                   seed_base, xi, cost_out, static_cast<hipStream_t>(stream));
}

int sgmcmc_bnn_fused_sgld_steps_f32(float *theta, float *grad, float *tau, float *g, float *v_hat, float *minv,
                                    size_t n_params, size_t chain_stride, int n_chains, const int *layer_sizes,
                                    int n_layers, const float *X, const float *y, size_t n_data,
                                    const int *window_starts, int batch, double batch_size, double n_examples,
                                    double wdecay, double prior_mean, double prior_var, float eps, float scale_grad,
                                    float A, uint64_t first_step, uint64_t n_steps, uint64_t burn_in_steps,
                                    uint64_t seed_base, const float *xi, float *cost_out, sgmcmc_stream_t stream)
{
    return bnn_fused_steps<float, 1>(theta, nullptr, grad, tau, g, v_hat, minv, n_params, chain_stride, n_chains,
                                     layer_sizes, n_layers, X, y, n_data, window_starts, batch, batch_size, n_examples,
                                     wdecay, prior_mean, prior_var, eps, scale_grad, A, first_step, n_steps, burn_in_steps,
                                     seed_base, xi, cost_out, static_cast<hipStream_t>(stream));
}
int sgmcmc_bnn_fused_sgld_steps_f64(double *theta, double *grad, double *tau, double *g, double *v_hat, double *minv,
                                    size_t n_params, size_t chain_stride, int n_chains, const int *layer_sizes,
                                    int n_layers, const double *X, const double *y, size_t n_data,
                                    const int *window_starts, int batch, double batch_size, double n_examples,
                                    double wdecay, double prior_mean, double prior_var, double eps, double scale_grad,
                                    double A, uint64_t first_step, uint64_t n_steps, uint64_t burn_in_steps,
                                    uint64_t seed_base, const double *xi, double *cost_out, sgmcmc_stream_t stream)
{
    return bnn_fused_steps<double, 1>(theta, nullptr, grad, tau, g, v_hat, minv, n_params, chain_stride, n_chains,
                                      layer_sizes, n_layers, X, y, n_data, window_starts, batch, batch_size, n_examples,
                                      wdecay, prior_mean, prior_var, eps, scale_grad, A, first_step, n_steps,
                                      burn_in_steps, seed_base, xi, cost_out, static_cast<hipStream_t>(stream));
}

}  // extern "C"
