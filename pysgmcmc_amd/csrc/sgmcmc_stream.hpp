// sgmcmc_stream.hpp -- the ONE streaming kernel shape every per-element operator of libsgmcmc_hip.so shares
// (stream_quads_vec / stream_quads_scalar), its fused statistics reduction, the fused Welford moments, and the host-side
// launch logic. Included by the per-sampler translation units (sgmcmc_sghmc.hip, sgmcmc_sgld.hip, sgmcmc_rsghmc.hip)
// and by sgmcmc_kernels.hip; everything lives in an anonymous namespace (one copy per translation unit, compiled side
// by side by `make -j`).
//
// HBM-bound elementwise pass, no contraction => no MFMA, no LDS staging of the streamed arrays. Work unit = one
// "quad" of 4 consecutive elements per lane: 16 B per lane per array (global_load/store_dwordx4, 1 KiB per wave
// instruction) and exactly one Philox4x32-10 call. All OLD state is read into registers before anything is written
// (the tf.control_dependencies contract of pysgmcmc/samplers/sghmc.py:170-200 made structural).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdint>
#include <type_traits>

#include "sgmcmc_hip.h"

#pragma clang fp contract(off)

#include "sgmcmc_device.hpp"
#include "sgmcmc_host.hpp"

namespace {

using sgmcmc_host::fail;
using sgmcmc_host::hip_fail;

// --------------------------------------------------------------------------
// fused step statistics
// --------------------------------------------------------------------------

// Per-launch extras of a step kernel (by value): where this launch's per-block statistics go, and the Welford
// moments a step may update in the same pass (K4 fused into the step, sgmcmc_step_opts_t).
// Side job of a step launch (sgmcmc_step_opts_t.gather_*): the NEXT step's minibatch window -- `rows` consecutive rows of the
// dataset, contiguous in memory -- copied into the feed buffers by the first `copy_blocks` workgroups of the launch, which do
// nothing else. 16-byte units: row_quads per row, destination rows dst_pitch_quads apart; y as `y_words` 32-bit words.
struct SideCopy {
    const void *xsrc = nullptr;
    void *xdst = nullptr;
    const void *ysrc = nullptr;
    void *ydst = nullptr;
    unsigned rows = 0, row_quads = 0, dst_pitch_quads = 0, y_words = 0;
};

__device__ __forceinline__ void side_copy(const SideCopy &c, unsigned block, unsigned blocks)
{
    struct alignas(16) Q { unsigned v[4]; };
    const size_t G = (size_t)blocks * blockDim.x, gid = (size_t)block * blockDim.x + threadIdx.x;
    const size_t nq = (size_t)c.rows * c.row_quads;
    const Q *__restrict__ src = static_cast<const Q *>(c.xsrc);
    Q *__restrict__ dst = static_cast<Q *>(c.xdst);
    for (size_t q = gid; q < nq; q += G) {
        const size_t r = q / c.row_quads, col = q - r * c.row_quads;
        dst[r * c.dst_pitch_quads + col] = src[q];
    }
    const unsigned *__restrict__ ys = static_cast<const unsigned *>(c.ysrc);
    unsigned *__restrict__ yd = static_cast<unsigned *>(c.ydst);
    for (size_t i = gid; i < c.y_words; i += G) yd[i] = ys[i];
}

__global__ void __launch_bounds__(256) __attribute__((unused)) side_copy_kernel(const SideCopy c) { side_copy(c, blockIdx.x, gridDim.x); }

template <typename T>
struct StreamExtras {
    unsigned copy_blocks = 0;     // single-pass vector variant: workgroups [0, copy_blocks) run the side job `cp` and nothing else
    SideCopy cp;
    unsigned part_base = 0;       // block b writes statistics record part_base + b
    unsigned part_total = 0;      // record count written to the workspace header (0 = this launch's grid)
    T *mom_mean = nullptr;        // Welford running mean / sum of squared deviations (MOM variants only)
    T *mom_m2 = nullptr;
    T mom_inv = T(0);             // 1 / count (count includes this sample)
};

// Fused step statistics ("LDS-staged reduction, wavefront shuffles for the partial sums"): every lane holds the sums of
// its quad in registers, a wave combines them with DPP lane moves (row shifts + row broadcasts across the 64 lanes, VALU
// only), the waves of a block meet in LDS, and lanes 0..3 write ONE 32-byte record {sum theta'^2, sum V'^2, sum minv,
// sum minv^2} per block -- block-major, i.e. a single 32-byte sector write. sgmcmc_step_stats_finish adds the records in
// block order: bit-reproducible for a given launch geometry, no extra HBM pass.
// History (profiles/r03_stats_variant_cost.txt): round 2 wrote statistic-major partials (four 8-byte sector writes per
// block: +0.25 B/param of write traffic at 128-lane blocks). One record per WAVE without LDS and barrier was tried in
// round 3 and was no faster (relativistic step at 49.8 M: 169-172 vs 165-168 us) while doubling the record bytes.
// T = the kernel's dtype: f32 kernels reduce across the wave in f32 (6 DPP adds per statistic), f64 kernels in f64.
// MASK = the statistics reduced (the others are written as 0 without any reduction work).
template <typename T, unsigned MASK, typename ACC>
__device__ __forceinline__ void stats_block_write(ACC (&acc)[4], double *__restrict__ part, unsigned part_base,
                                                  unsigned part_total, unsigned block, unsigned blocks)
{
    __shared__ T lds[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if ((MASK >> k) & 1u) {
            T v = wave_sum_dpp_lane63((T)acc[k]);
            if (lane == 63) lds[wave][k] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int nw = blockDim.x >> 6;
        double v = 0.0;
        if ((MASK >> threadIdx.x) & 1u)
            for (int w = 0; w < nw; ++w) v += (double)lds[w][threadIdx.x];
        // workspace = 32-byte header {number of records} + block-major records [nrecords][4]
        part[4 + 4 * ((size_t)part_base + block) + threadIdx.x] = v;
    }
    if (block == 0 && threadIdx.x == 0)
        reinterpret_cast<unsigned long long *>(part)[0] = part_total ? part_total : blocks;
}

// Welford update of (mean, m2) with the sample x, one IEEE rounding per op (the arithmetic of MomentsOp / K4, so the
// fused form equals the separate launch bit for bit)
template <typename T>
__device__ __forceinline__ void welford_quad(const T (&x)[4], T (&mu)[4], T (&m2)[4], T inv)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        T d = x[j] - mu[j];
        T m = mu[j] + d * inv;
        mu[j] = m;
        m2[j] = m2[j] + d * (x[j] - m);
    }
}

// VEC: arrays are 16-B aligned; quads [0, nq_full) go through dwordx4 accesses, QPT quads in flight per lane; the
//      ragged tail (n % 4 elements) is done element-wise by one lane.
// STATS: 0 = none (the plain variant carries no reduction code), 1 = every statistic of the operator, 2 = sum theta'^2 only.
// MOM:  also fold theta' into the Welford moments ex.mom_mean / ex.mom_m2 (single-pass variant only).
template <typename Op, int QPT, bool NT, int STATS, bool LOOP, bool MOM>
__global__ void __launch_bounds__(256) stream_quads_vec(const Op op_in, size_t nq_full, int tail_cnt,
                                                        const StreamExtras<typename Op::real> ex)
{
    typedef typename Op::real T;
    unsigned block = blockIdx.x, blocks = gridDim.x;
    if constexpr (!LOOP) {
        // side job (uniform per workgroup): the first copy_blocks workgroups copy the next minibatch window and leave; the update
        // runs on the others exactly as without it (same quads per block, same statistics records)
        if (ex.copy_blocks != 0) {
            if (block < ex.copy_blocks) {
                side_copy(ex.cp, block, ex.copy_blocks);
                return;
            }
            block -= ex.copy_blocks;
            blocks -= ex.copy_blocks;
        }
    }
    Op op = op_in;
    op.prepare();
    const size_t G = (size_t)blocks * blockDim.x;
    const size_t gid = (size_t)block * blockDim.x + threadIdx.x;
    // running sums of this lane: the single-pass variant adds at most one quad (+ the ragged tail) -> the kernel's own
    // dtype, no f64 round trip in front of the wave reduction; the looping variants add many quads -> double
    typedef typename std::conditional<LOOP, double, T>::type acc_t;
    acc_t acc[4] = {acc_t(0), acc_t(0), acc_t(0), acc_t(0)};
    constexpr bool stats = STATS != 0;
    constexpr bool tsq_only = STATS == 2;
    if constexpr (!LOOP) {
        // the grid covers every quad (the default geometry): straight-line code, no loop-carried
        // scalar state -> fewer SGPRs/VGPRs -> one more resident block per CU
        static_assert(QPT == 1, "single-pass variant is one quad per lane");
        if (gid < nq_full) {
            typename Op::Regs R;
            T mu[4], m2[4];
            op.template load_vec<NT>(gid, R);
            if constexpr (MOM) { load_quad<NT>(ex.mom_mean, gid, mu); load_quad<NT>(ex.mom_m2, gid, m2); }
            // every load is ISSUED before any arithmetic: Philox + Box-Muller (which need no loaded value) then run under
            // the memory latency. Without the fence the scheduler sank the loads below the Philox rounds in some
            // variants (relativistic step with statistics: 168 instead of 158 us at 49.8 M parameters).
            __builtin_amdgcn_sched_barrier(0);
            op.compute(gid, R);
            op.template store_vec<NT>(gid, R);
            if constexpr (MOM) {
                welford_quad<T>(R.th, mu, m2, ex.mom_inv);
                store_quad<NT>(ex.mom_mean, gid, mu); store_quad<NT>(ex.mom_m2, gid, m2);
            }
            if constexpr (stats) op.template accumulate<tsq_only>(R, 4, acc);
        }
    } else {
        static_assert(!MOM, "fused moments ride on the single-pass variant");
        for (size_t base = gid; base < nq_full; base += G * QPT) {
            typename Op::Regs R[QPT];
#pragma unroll
            for (int u = 0; u < QPT; ++u) {
                size_t q = base + (size_t)u * G;
                if (q < nq_full) op.template load_vec<NT>(q, R[u]);
            }
#pragma unroll
            for (int u = 0; u < QPT; ++u) {
                size_t q = base + (size_t)u * G;
                if (q < nq_full) op.compute(q, R[u]);
            }
#pragma unroll
            for (int u = 0; u < QPT; ++u) {
                size_t q = base + (size_t)u * G;
                if (q < nq_full) {
                    op.template store_vec<NT>(q, R[u]);
                    if constexpr (stats) op.template accumulate<tsq_only>(R[u], 4, acc);
                }
            }
        }
    }
    if (tail_cnt && gid == G - 1) {
        typename Op::Regs R;
        T mu[4], m2[4];
        op.load_part_(nq_full, tail_cnt, R);
        if constexpr (MOM) { load_part(ex.mom_mean, nq_full, tail_cnt, mu); load_part(ex.mom_m2, nq_full, tail_cnt, m2); }
        op.compute(nq_full, R);
        op.store_part_(nq_full, tail_cnt, R);
        if constexpr (MOM) {
            welford_quad<T>(R.th, mu, m2, ex.mom_inv);
            store_part(ex.mom_mean, nq_full, tail_cnt, mu); store_part(ex.mom_m2, nq_full, tail_cnt, m2);
        }
        if constexpr (stats) op.template accumulate<tsq_only>(R, tail_cnt, acc);
    }
    if constexpr (stats)
        stats_block_write<T, tsq_only ? (Op::stats_mask & 1u) : Op::stats_mask>(acc, op.stats_part, ex.part_base, ex.part_total, block, blocks);
}

// element-wise path for misaligned arrays: same quads, same results
template <typename Op, bool STATS>
__global__ void __launch_bounds__(256) stream_quads_scalar(const Op op_in, size_t n, const StreamExtras<typename Op::real> ex)
{
    Op op = op_in;
    op.prepare();
    const size_t G = (size_t)gridDim.x * blockDim.x;
    const size_t nq = (n + 3) / 4;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    constexpr bool stats = STATS;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += G) {
        size_t left = n - 4 * q;
        int cnt = left >= 4 ? 4 : (int)left;
        typename Op::Regs R;
        op.load_part_(q, cnt, R);
        op.compute(q, R);
        op.store_part_(q, cnt, R);
        if constexpr (stats) op.template accumulate<false>(R, cnt, acc);
    }
    if constexpr (stats)
        stats_block_write<typename Op::real, Op::stats_mask>(acc, op.stats_part, ex.part_base, ex.part_total, blockIdx.x, gridDim.x);
}

// --------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------

// Launch geometry of ONE call (sgmcmc_launch_t in the header; NULL = these defaults). The library keeps
// no mutable state: two host threads may step two chains with different geometries concurrently.
struct LaunchCfg {
    int block_threads = -1;       // -1 = auto: 128 lanes when a launch streams > NT_AUTO_BYTES, else 256
    int qpt = 1;
    int max_blocks = 1 << 20;
    int nt = 2;                   // 0 = plain, 1 = nt, 2 = auto by working-set size
    int bt = 256;                 // resolved block size of this launch (set by launch())
    hipEvent_t ev0 = nullptr, ev1 = nullptr;   // kernel start / stop timestamps (hipExtLaunchKernel), both or none
};
// hipLaunchKernelGGL, or the timestamping launch when the caller passed events
#define SGMCMC_LAUNCH(KERNEL, GRID, BLOCK, STREAM, CFG, ...)                                                         \
    do {                                                                                                             \
        if ((CFG).ev0 != nullptr || (CFG).ev1 != nullptr)                                                            \
            hipExtLaunchKernelGGL(KERNEL, dim3(GRID), dim3(BLOCK), 0, STREAM, (CFG).ev0, (CFG).ev1, 0, __VA_ARGS__); \
        else                                                                                                         \
            hipLaunchKernelGGL(KERNEL, dim3(GRID), dim3(BLOCK), 0, STREAM, __VA_ARGS__);                             \
    } while (0)
// validates *in (0 / -1 fields keep the default); returns 0 or SGMCMC_EINVAL
inline int resolve_launch(const sgmcmc_launch_t *in, LaunchCfg &c)
{
    if (!in) return 0;
    if (in->block_threads != 0) {
        if (in->block_threads != -1 && (in->block_threads < 64 || in->block_threads > 256 || (in->block_threads % 64) != 0))
            return fail(SGMCMC_EINVAL, "launch.block_threads must be 64, 128, 192, 256, 0 (default) or -1 (auto)");
        c.block_threads = in->block_threads;
    }
    if (in->quads_per_thread != 0) {
        if (in->quads_per_thread != 1 && in->quads_per_thread != 2 && in->quads_per_thread != 4)
            return fail(SGMCMC_EINVAL, "launch.quads_per_thread must be 0 (default), 1, 2 or 4");
        c.qpt = in->quads_per_thread;
    }
    if (in->max_blocks != 0) {
        if (in->max_blocks < 1) return fail(SGMCMC_EINVAL, "launch.max_blocks must be 0 (default) or >= 1");
        c.max_blocks = in->max_blocks;
    }
    if (in->nontemporal != -1) {
        if (in->nontemporal < 0 || in->nontemporal > 2)
            return fail(SGMCMC_EINVAL, "launch.nontemporal must be -1 (default), 0 (off), 1 (on) or 2 (auto)");
        c.nt = in->nontemporal;
    }
    c.ev0 = static_cast<hipEvent_t>(in->start_event);
    c.ev1 = static_cast<hipEvent_t>(in->stop_event);
    return 0;
}
// Above this many bytes touched per launch the arrays cannot stay in the 256 MiB
// Infinity Cache between steps and nt accesses win (+6..7 % at 1.2 GB); below it
// plain accesses win (the cache holds part of the working set across steps:
// -5..-12 % with nt at 240 MB and 480 MB; +8 % at 800 MB). Measured on MI355X, profiles/r01_tune_*.txt.
constexpr size_t NT_AUTO_BYTES = (size_t)640 << 20;

inline bool aligned16(const void *p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline NoiseKey make_key(uint64_t seed, uint64_t step, const uint64_t *step_dev, uint64_t first_element = 0)
{
    NoiseKey nk;
    nk.k0 = (uint32_t)seed; nk.k1 = (uint32_t)(seed >> 32);
    nk.s0 = (uint32_t)step; nk.s1 = (uint32_t)(step >> 32);
    nk.step_dev = step_dev;
    nk.q0 = first_element / 4;
    return nk;
}

// Upper bound of the grid any launch of n elements can use (sizes the stats workspace).
inline size_t max_grid_for(size_t n)
{
    size_t nq = (n + 3) / 4;
    size_t want = (nq + 63) / 64;                      // smallest block (64 threads), 1 quad per lane
    size_t cap = (size_t)1 << 20;
    return want < cap ? (want ? want : 1) : cap;
}

// What a step launch was asked to do besides the update (resolved sgmcmc_step_opts_t)
template <typename T>
struct StepExtras {
    StreamExtras<T> ex;
    int stats_mode = 1;            // with a stats workspace: 1 = all statistics of the operator, 2 = sum theta'^2 only
    bool want_moments = false;
    bool big_hint = false;         // the launch is a slice of a working set larger than NT_AUTO_BYTES
    bool want_copy = false;        // ex.cp holds a window to gather (opts.gather_*)
    bool *copy_done = nullptr;     // set by launch_vec when the side job rode in the step launch
};

constexpr unsigned SIDE_COPY_MAX_BLOCKS = 256;      // workgroups of 256 lanes x 16 bytes a window copy is spread over

inline unsigned side_copy_blocks(const SideCopy &c, int bt)
{
    const size_t units = (size_t)c.rows * c.row_quads + c.y_words;
    const size_t want = (units + (size_t)bt - 1) / (size_t)bt;
    return (unsigned)(want < SIDE_COPY_MAX_BLOCKS ? (want ? want : 1) : SIDE_COPY_MAX_BLOCKS);
}

// grid of a vector launch of n elements with bt-lane blocks and QPT quads per lane (before the max_blocks cap)
inline size_t want_blocks(size_t n, int bt, int qpt)
{
    const size_t nq_full = n / 4;
    const size_t per_block = (size_t)bt * qpt;
    size_t want = (nq_full + per_block - 1) / per_block;
    return want ? want : 1;
}

// Returns 0, an error code, or (when `moments_done` is given) reports whether the fused-moments variant ran.
template <typename Op, int QPT, bool NT, bool FAST_VARIANTS>
int launch_vec(const Op &op, size_t n, const LaunchCfg &cfg, const StepExtras<typename Op::real> &se, bool *moments_done,
               hipStream_t st)
{
    const int bt = cfg.bt;
    const size_t nq_full = n / 4;
    const int tail = (int)(n % 4);
    const size_t want = want_blocks(n, bt, QPT);
    size_t cap = (size_t)cfg.max_blocks;
    unsigned grid = (unsigned)(want < cap ? want : cap);
    const bool with_stats = op.stats_part != nullptr;
    StreamExtras<typename Op::real> ex = se.ex;
    ex.copy_blocks = 0;
    if (moments_done) *moments_done = false;
    if constexpr (QPT == 1) {
        if (want <= cap) {                                 // one quad per lane, whole array in one pass
            const int smode = with_stats ? (FAST_VARIANTS ? se.stats_mode : 1) : 0;
            const bool mom = FAST_VARIANTS && se.want_moments;
            if (se.want_copy && se.copy_done != nullptr) { // the window gather rides in this launch: extra workgroups in front
                ex.copy_blocks = side_copy_blocks(ex.cp, bt);
                grid += ex.copy_blocks;
                *se.copy_done = true;
            }
#define SGMCMC_FAST(SM, MOMV) SGMCMC_LAUNCH((stream_quads_vec<Op, 1, NT, SM, false, MOMV>), grid, bt, st, cfg, op, nq_full, tail, ex)
            if constexpr (FAST_VARIANTS) {
                if (mom) {
                    if (smode == 2) SGMCMC_FAST(2, true); else if (smode == 1) SGMCMC_FAST(1, true); else SGMCMC_FAST(0, true);
                    if (moments_done) *moments_done = true;
                } else {
                    if (smode == 2) SGMCMC_FAST(2, false); else if (smode == 1) SGMCMC_FAST(1, false); else SGMCMC_FAST(0, false);
                }
            } else {
                if (smode) SGMCMC_FAST(1, false); else SGMCMC_FAST(0, false);
            }
#undef SGMCMC_FAST
            hipError_t e1 = hipGetLastError();
            return e1 == hipSuccess ? 0 : hip_fail(e1, "launch stream_quads_vec");
        }
    }
    if (with_stats)
        SGMCMC_LAUNCH((stream_quads_vec<Op, QPT, NT, 1, true, false>), grid, bt, st, cfg, op, nq_full, tail, ex);
    else
        SGMCMC_LAUNCH((stream_quads_vec<Op, QPT, NT, 0, true, false>), grid, bt, st, cfg, op, nq_full, tail, ex);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "launch stream_quads_vec");
}
template <typename Op>
int launch_scalar(const Op &op, size_t n, const LaunchCfg &cfg, const StepExtras<typename Op::real> &se, hipStream_t st)
{
    const int bt = cfg.bt;
    size_t nq = (n + 3) / 4;
    size_t want = (nq + bt - 1) / bt;
    if (want == 0) want = 1;
    size_t cap = (size_t)cfg.max_blocks;
    unsigned grid = (unsigned)(want < cap ? want : cap);
    if (op.stats_part != nullptr)
        SGMCMC_LAUNCH((stream_quads_scalar<Op, true>), grid, bt, st, cfg, op, n, se.ex);
    else
        SGMCMC_LAUNCH((stream_quads_scalar<Op, false>), grid, bt, st, cfg, op, n, se.ex);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "launch stream_quads_scalar");
}

// f32 ops honour the (quads_per_thread, nontemporal) knobs; f64 ops (a quad is already 32 B per lane per array) use one
// quad per lane. FAST_VARIANTS: instantiate the sum-theta'^2-only and fused-moments forms of the single-pass kernel (the
// in-register-noise step operators; injected-noise and utility operators keep the two basic forms).
template <typename Op, bool FAST_VARIANTS = false>
int launch(const Op &op, size_t n, bool vec_ok, size_t bytes_per_elem, const sgmcmc_launch_t *launch_in,
           const StepExtras<typename Op::real> &se, bool *moments_done, hipStream_t st)
{
    if (moments_done) *moments_done = false;
    if (n == 0) return 0;
    LaunchCfg cfg;
    if (int rc = resolve_launch(launch_in, cfg)) return rc;
    const bool big = se.big_hint || n * bytes_per_elem > NT_AUTO_BYTES;     // cannot stay in the Infinity Cache between steps
    // measured (profiles/r01_block_sweep.txt): 128-lane blocks +7 % at 50 M params (HBM-resident), 256 +2 % at 10 M
    cfg.bt = cfg.block_threads > 0 ? cfg.block_threads : (big ? 128 : 256);
    if (!vec_ok) return launch_scalar<Op>(op, n, cfg, se, st);
    // auto (2): nt for f32 launches that cannot stay in the Infinity Cache. f64 launches never: with 32 B per lane per array plain
    // accesses win at every size (49.8 M parameters: K1 frozen 412.8 vs 439.0 us, K1 burn-in 796 vs 870, K2 283 vs 295, K3 376 vs 385;
    // profiles/r05_tune_50m_f64.txt)
    const bool nt = cfg.nt == 2 ? (big && sizeof(typename Op::real) == 4) : (cfg.nt != 0);
    if (sizeof(typename Op::real) == 8) {
        return nt ? launch_vec<Op, 1, true, FAST_VARIANTS>(op, n, cfg, se, moments_done, st)
                  : launch_vec<Op, 1, false, FAST_VARIANTS>(op, n, cfg, se, moments_done, st);
    }
    const int qpt = cfg.qpt;
    if (nt) {
        if (qpt >= 4) return launch_vec<Op, 4, true, false>(op, n, cfg, se, moments_done, st);
        if (qpt == 2) return launch_vec<Op, 2, true, false>(op, n, cfg, se, moments_done, st);
        return launch_vec<Op, 1, true, FAST_VARIANTS>(op, n, cfg, se, moments_done, st);
    }
    if (qpt >= 4) return launch_vec<Op, 4, false, false>(op, n, cfg, se, moments_done, st);
    if (qpt == 2) return launch_vec<Op, 2, false, false>(op, n, cfg, se, moments_done, st);
    return launch_vec<Op, 1, false, FAST_VARIANTS>(op, n, cfg, se, moments_done, st);
}
// operators without step extras (K4, K5)
template <typename Op>
int launch(const Op &op, size_t n, bool vec_ok, size_t bytes_per_elem, const sgmcmc_launch_t *launch_in, hipStream_t st)
{
    StepExtras<typename Op::real> se;
    return launch<Op, false>(op, n, vec_ok, bytes_per_elem, launch_in, se, nullptr, st);
}

// Resolve the optional sgmcmc_step_opts_t of a step call. `n` = elements of this launch.
template <typename T>
int resolve_step_opts(const sgmcmc_step_opts_t *o, size_t n, void *stats_ws, StepExtras<T> &se, uint64_t &first_element,
                      const char *who)
{
    first_element = 0;
    if (!o) return 0;
    if (o->first_element % 4 != 0)
        return fail(SGMCMC_EINVAL, "%s: opts.first_element must be a multiple of 4 (slices start on a quad)", who);
    first_element = o->first_element;
    if (o->stats_select != 0 && o->stats_select != SGMCMC_STATS_THETA_SQ)
        return fail(SGMCMC_EINVAL, "%s: opts.stats_select must be 0 (all) or SGMCMC_STATS_THETA_SQ", who);
    se.stats_mode = o->stats_select == SGMCMC_STATS_THETA_SQ ? 2 : 1;
    if ((o->stats_record_base || o->stats_record_total) && !stats_ws)
        return fail(SGMCMC_EINVAL, "%s: opts.stats_record_* without a stats workspace", who);
    se.ex.part_base = o->stats_record_base;
    se.ex.part_total = o->stats_record_total;
    if ((o->moments_mean == nullptr) != (o->moments_m2 == nullptr))
        return fail(SGMCMC_EINVAL, "%s: opts.moments_mean and opts.moments_m2 go together", who);
    if (o->moments_mean) {
        if (o->moments_count == 0) return fail(SGMCMC_EINVAL, "%s: opts.moments_count must be >= 1", who);
        se.want_moments = true;
        se.ex.mom_mean = static_cast<T *>(o->moments_mean);
        se.ex.mom_m2 = static_cast<T *>(o->moments_m2);
        se.ex.mom_inv = T(1) / (T)o->moments_count;
    }
    se.big_hint = (o->flags & SGMCMC_STEP_HBM_RESIDENT) != 0;
    if (o->gather_x != nullptr || o->gather_x_out != nullptr || o->gather_batch != 0) {
        // the next step's minibatch window (sgmcmc_window_gather_*'s job) as a side job of this launch
        const size_t es = sizeof(T);
        if (!o->gather_x || !o->gather_x_out || !o->gather_y || !o->gather_y_out || o->gather_batch == 0 || o->gather_dim == 0 ||
            o->gather_x_out_ld < o->gather_dim)
            return fail(SGMCMC_EINVAL, "%s: opts.gather_* needs X, y, both outputs, batch > 0, dim > 0 and x_out_ld >= dim", who);
        const unsigned char *src = static_cast<const unsigned char *>(o->gather_x) + (size_t)o->gather_start * o->gather_dim * es;
        if (((size_t)o->gather_dim * es) % 16 || ((size_t)o->gather_x_out_ld * es) % 16 ||
            ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(o->gather_x_out)) & 15u) ||
            (size_t)o->gather_batch * o->gather_dim * es / 16 >= 0xffffffffull)
            return fail(SGMCMC_EINVAL, "%s: opts.gather_* needs rows of a multiple of 16 bytes, 16-byte aligned source window and "
                                       "destination (use sgmcmc_window_gather_* otherwise)", who);
        se.want_copy = true;
        se.ex.cp.xsrc = src;
        se.ex.cp.xdst = o->gather_x_out;
        se.ex.cp.ysrc = static_cast<const unsigned char *>(o->gather_y) + (size_t)o->gather_start * es;
        se.ex.cp.ydst = o->gather_y_out;
        se.ex.cp.rows = o->gather_batch;
        se.ex.cp.row_quads = (unsigned)((size_t)o->gather_dim * es / 16);
        se.ex.cp.dst_pitch_quads = (unsigned)((size_t)o->gather_x_out_ld * es / 16);
        se.ex.cp.y_words = (unsigned)((size_t)o->gather_batch * es / 4);
    }
    (void)n;
    return 0;
}

// The window gather of a step call whose launch had no fused form for it (looping / element-wise variants): its own small launch
// behind the step on the same stream -- same bytes either way.
template <typename T>
int finish_side_copy(const StepExtras<T> &se, bool rode_in_the_step, hipStream_t st)
{
    if (!se.want_copy || rode_in_the_step) return 0;
    hipLaunchKernelGGL(side_copy_kernel, dim3(side_copy_blocks(se.ex.cp, 256)), dim3(256), 0, st, se.ex.cp);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "launch side_copy");
}

}  // namespace
