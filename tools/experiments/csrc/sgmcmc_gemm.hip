// sgmcmc_gemm.hip -- the weight-gradient GEMM of a dense layer on the fp32 matrix cores, with the SGHMC update of that
// layer's weights as its epilogue.
//
//   gW[m][n] = sum_k A[k][m] * B[k][n]        A = the layer's input activations h_{l-1}  [K = batch][M = fan_in]
//                                             B = the layer's back-propagated deltas     [K = batch][N = fan_out]
//
// is the last thing the backward pass computes for a layer, and the ONLY consumer of gW is the sampler's update of W
// (pysgmcmc/samplers/sghmc.py:211-251). Fused form (gemm_tn_sghmc_kernel): the tile of gW a workgroup has just accumulated
// never goes to HBM -- the workgroup loads the same tile of theta, V and minv, draws the tile's Philox normals (the stream of
// the streaming kernel K1: counter = (step, global quad index)), applies the frozen SGHMC update (SghmcOp::compute, one IEEE
// rounding per reference op) and writes theta', V'. Per parameter that is 20 B of HBM traffic instead of 4 (GEMM writes gW)
// + 24 (K1) and one launch less. What it does NOT do is hide the update's arithmetic: fp32 MFMA and vector-ALU work of
// different waves on one SIMD take the sum of their times on gfx950 (tools/gpu/mfma_valu_overlap.hip), so the fused kernel
// costs product time + ~8 us of Philox / Box-Muller / update ALU work per 2048 x 2048 layer, and in the sampler it is a draw
// with library GEMM + one K1 launch (profiles/r03_gemm_fusion_probe.txt). It stays opt-in (sampler.fuse_update_into_gemm).
//
// Tiling for gfx950: 64 x 64 output tile per 256-lane workgroup (4 waves in 2 x 2, each one v_mfma_f32_32x32x2_f32 tile = 16
// accumulator registers), four workgroups per CU. Operand fragments are single floats per lane (A[m = lane & 31][k = lane
// >> 5], B[k][n = lane & 31]) read from k-major LDS rows: conflict-free ds_read_b32. The K loop of the default variant
// (mainloop_dma) requests its operands with global_load_lds_dwordx4 -- 16 bytes per lane from global memory straight into LDS
// -- in chunks of 16, two chunks ahead, into a ring of three stages, with an explicit vmcnt wait and ONE bare s_barrier per
// chunk; it runs at the MFMA rate (12.9 us of a 22.0 us launch at 2048 x 2048 x 256; the rest is launch, first-chunk latency,
// tail and the output store). The other variants (Tile<...>/mainloop: operands staged through registers, bigger tiles) are
// kept for tools/gemm_probe2.py. The accumulators leave through LDS so that stores (and the fused update) work on row-major
// quads: a lane owns 4 consecutive columns of a row = one Philox quad and one 16-byte access per array.
//
// fp32 MFMA is an exact fmaf chain in k order (MI355X_MICROARCH.md): the product differs from a library GEMM only in
// summation order. The update arithmetic is bit-identical to K1 applied to the same gW (tests write gW out and check).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "sgmcmc_gemm_experiments.h"

#pragma clang fp contract(off)

#include "sgmcmc_device.hpp"
#include "sgmcmc_host.hpp"

using namespace sgmcmc_host;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));


struct GemmArgs {
    const float *A, *B;       // [K][lda], [K][ldb]
    float *C;                 // [M][ldc] (plain mode, or the optional gradient copy of the fused mode)
    int M, N, K, lda, ldb, ldc;
    int probe = 0;            // timing experiments (tools/gemm_probe2.py): 1 = no output store
    int *phase_counters = nullptr;   // nullable: 2048 ints, one per hardware CU (see cu_arrival_slot)
    int phase_sleep = 0;             // the k-th workgroup a CU receives starts k * phase_sleep * 64 cycles late
};

// Which workgroup of its CU is this one (0, 1, 2, ...)? Counted per hardware CU id with one atomic per workgroup.
// Co-resident workgroups that start together run their K chunks in lockstep -- all of a SIMD's waves reach the
// "write the next chunk to LDS, barrier, read fragments" phase of a chunk at the same time and the matrix pipe idles for that
// phase of every chunk. A start delay of a fraction of a chunk per slot keeps them out of phase for the whole K loop.
__device__ __forceinline__ int cu_arrival_slot(int *counters)
{
    __shared__ int slot;
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // HW_REG_XCC_ID [3:0]
        const unsigned key = ((xcc & 7u) << 8) | (((hw >> 13) & 7u) << 5) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u);
        slot = atomicAdd(&counters[key], 1);
    }
    __syncthreads();
    return slot;
}

// One workgroup = WM x WN waves, each wave owns TM x TN MFMA tiles of 32 x 32: block tile BM = 32 TM WM by BN = 32 TN WN.
// K advances in chunks of BK through a double-buffered LDS stage; inside a chunk the MFMAs run in sub-batches of 8 k-steps
// whose operand fragments are read from LDS one sub-batch ahead.
// SUB: k-steps whose fragments are read ahead; FENCE: pin "read the next sub-batch, then issue this one's MFMAs" with
// sched_barriers (measured: the compiler's own interleave is 2-3 us faster at 2048 x 2048 x 256, so the default is off;
// s_setprio around the MFMAs costs 1.5-2.5 us).
template <int TM, int TN, int WM, int WN, int BK_, int SUB_ = 4, bool FENCE_ = false>
struct Tile {
    static constexpr int BK = BK_, SUB = SUB_;
    static constexpr bool FENCE = FENCE_;
    static constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, NT = 64 * WM * WN;
    static constexpr int LA = BK * BM / 4 / NT, LB = BK * BN / 4 / NT;      // float4 loads per lane per chunk
    static_assert(LA >= 1 && LB >= 1 && LA * NT * 4 == BK * BM && LB * NT * 4 == BK * BN, "chunk must divide over the lanes");
};

typedef float f32x4_t __attribute__((ext_vector_type(4)));

// LDS of one workgroup: two stages of a BK x BM and a BK x BN chunk
template <typename TL>
struct Stages {
    float A[2][TL::BK][TL::BM];
    float B[2][TL::BK][TL::BN];
};

// The register-staged K loop (probe variants 1-9): acc[i][j] = 32 x 32 tile (rows +32 i, cols +32 j) of this wave's part of
// the block tile at (m0, n0); the last chunk is peeled out of the loop (no fetch of a chunk that does not exist).
template <typename TL, int TM, int TN>
__device__ __forceinline__ void mainloop(const GemmArgs &g, Stages<TL> &lds, int m0, int n0, int wm, int wn,
                                         f32x16 (&acc)[TM][TN])
{
    constexpr int BK = TL::BK, BM = TL::BM, BN = TL::BN, NT = TL::NT, LA = TL::LA, LB = TL::LB;
    constexpr int SUB = TL::SUB, NSUB = BK / 2 / SUB;       // k-steps (of 2) per sub-batch, sub-batches per chunk
    static_assert(NSUB >= 1 && NSUB * SUB * 2 == BK, "BK must be a multiple of 2 SUB");
    const int tid = threadIdx.x, lane = tid & 63;
    const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4_t ra[LA], rb[LB];
    // lane -> (row, float4 column) of a BK x BM (BK x BN) chunk, 512-byte rows read by consecutive lanes
    auto fetch = [&](int kc) {
        const size_t ko = (size_t)kc * BK;
#pragma unroll
        for (int u = 0; u < LA; ++u) {
            const int f = tid + u * NT, r = f / (BM / 4), c = (f % (BM / 4)) * 4;
            ra[u] = (m0 + c < g.M) ? *reinterpret_cast<const f32x4_t *>(g.A + (ko + r) * g.lda + m0 + c) : zero4;   // M % 4 == 0
        }
#pragma unroll
        for (int u = 0; u < LB; ++u) {
            const int f = tid + u * NT, r = f / (BN / 4), c = (f % (BN / 4)) * 4;
            rb[u] = *reinterpret_cast<const f32x4_t *>(g.B + (ko + r) * g.ldb + n0 + c);
        }
    };
    auto stash = [&](int s) {
#pragma unroll
        for (int u = 0; u < LA; ++u) {
            const int f = tid + u * NT, r = f / (BM / 4), c = (f % (BM / 4)) * 4;
            *reinterpret_cast<f32x4_t *>(&lds.A[s][r][c]) = ra[u];
        }
#pragma unroll
        for (int u = 0; u < LB; ++u) {
            const int f = tid + u * NT, r = f / (BN / 4), c = (f % (BN / 4)) * 4;
            *reinterpret_cast<f32x4_t *>(&lds.B[s][r][c]) = rb[u];
        }
    };
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nk = g.K / BK;
    fetch(0);
    stash(0);
    __syncthreads();
    const int kl = lane >> 5, cl = lane & 31;
    float af[2][SUB][TM], bf[2][SUB][TN];                  // operand fragments, double-buffered over sub-batches
    auto frags = [&](int s, int sub, int buf) {
#pragma unroll
        for (int kk = 0; kk < SUB; ++kk) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[buf][kk][i] = lds.A[s][2 * (sub * SUB + kk) + kl][wm + 32 * i + cl];
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[buf][kk][j] = lds.B[s][2 * (sub * SUB + kk) + kl][wn + 32 * j + cl];
        }
    };
    auto mfmas = [&](int s) {
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            if (sub + 1 < NSUB) frags(s, sub + 1, (sub + 1) & 1);   // next sub-batch's fragments are requested first ...
            if (TL::FENCE) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < SUB; ++kk)                         // ... and land under these MFMAs
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[sub & 1][kk][i], bf[sub & 1][kk][j], acc[i][j], 0, 0, 0);
            if (TL::FENCE) __builtin_amdgcn_sched_barrier(0);
        }
    };
    if (g.probe & 6) {
        // timing experiments only (results are wrong): 2 = operand fragments read from LDS once, 4 = also no global fetch,
        // LDS write and barrier per chunk -- what is left is the bare MFMA chain
        frags(0, 0, 0);
        for (int kc = 0; kc + 1 < nk; ++kc) {
            if (!(g.probe & 4)) fetch(kc + 1);
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
                for (int kk = 0; kk < SUB; ++kk)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][kk][i], bf[0][kk][j], acc[i][j], 0, 0, 0);
            if (!(g.probe & 4)) { stash((kc & 1) ^ 1); __syncthreads(); }
        }
        return;
    }
    for (int kc = 0; kc + 1 < nk; ++kc) {
        const int s = kc & 1;
        frags(s, 0, 0);
        fetch(kc + 1);                                     // global loads of the next chunk fly under this chunk's MFMAs
        mfmas(s);
        stash(s ^ 1);                                      // stage s ^ 1 was last read before the previous barrier
        __syncthreads();
    }
    {
        const int s = (nk - 1) & 1;
        frags(s, 0, 0);
        mfmas(s);
        __syncthreads();                                   // the stages may be overwritten by the next tile
    }
}

// ------------------------------------------------------------------------------------------------------------------
// K loop with DIRECT-TO-LDS operand loads (gfx950: global_load_lds_dwordx4)
// ------------------------------------------------------------------------------------------------------------------
// Timing the pieces of the loop above (tools/gemm_probe2.py) showed where it loses to the library: the bare MFMA chain takes
// 14 us at 2048 x 2048 x 256, the LDS fragment reads cost nothing, and staging the next chunk through registers (global load ->
// s_waitcnt -> ds_write -> barrier) costs 6 us whatever the chunk size. Here a lane's 16 bytes go from global memory straight
// into LDS (a wave fills 1 KiB of contiguous LDS per instruction = 4 rows of a 64-float chunk row), two chunks ahead, into a
// ring of NS stages: no staging registers, no ds_write, one barrier per chunk.
template <int BK, int BM, int BN, int NS>
struct Ring {
    float A[NS][BK][BM];
    float B[NS][BK][BN];
};

// one chunk (BK rows of A and of B) into ring stage st: a wave fills rows 16 h + 4 wave ... + 3 of both operands per h
template <int BK, int NS>
__device__ __forceinline__ void dma_issue(Ring<BK, 64, 64, NS> &lds, const float *ga, const float *gb, int lda, int ldb, int kc, int st, int wave)
{
#pragma unroll
    for (int h = 0; h < BK / 16; ++h) {
#if defined(__HIP_DEVICE_COMPILE__)                       /* the host pass has no declaration of this builtin */
        const size_t ko = (size_t)kc * BK + 16 * h;
        __builtin_amdgcn_global_load_lds(ga + ko * lda, &lds.A[st][16 * h + wave * 4][0], 16, 0, 0);
        __builtin_amdgcn_global_load_lds(gb + ko * ldb, &lds.B[st][16 * h + wave * 4][0], 16, 0, 0);
#endif
    }
}

struct NoSideLoads {
    __device__ __forceinline__ void operator()() const {}
};

// SIDE: a functor that issues exactly EXTRA more vector-memory loads (into registers) right after the first two chunks have been
// requested -- the fused kernel's theta/V/minv quads, which then travel under the whole K loop. vmcnt retires in order, so the
// waits for chunks 0 and 1 (older than the side loads) allow EXTRA more outstanding loads, and the wait for chunk 2 (younger)
// is also the wait for the side loads.
template <int BK, int NS, int EXTRA = 0, class SIDE = NoSideLoads>
__device__ __forceinline__ void mainloop_dma(const GemmArgs &g, Ring<BK, 64, 64, NS> &lds, int m0, int n0, int wm, int wn, f32x16 &acc,
                                             const SIDE &side = SIDE())
{
    static_assert(2 * (BK / 16) + EXTRA <= 15, "the wait count must fit the low vmcnt field");
    static_assert(BK % 16 == 0, "a 256-lane workgroup fills 16 rows of a 64-wide chunk per instruction");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // this lane's 16 bytes of a 16-row slab: row (wave * 4 + lane / 16), floats (lane % 16) * 4 ...; columns beyond M are clamped
    // (they only feed output rows >= M, which are never stored)
    const int lr = wave * 4 + (lane >> 4);
    int ca = m0 + (lane & 15) * 4;
    if (ca + 4 > g.M) ca = g.M - 4;
    const float *ga = g.A + (size_t)lr * g.lda + ca;
    const float *gb = g.B + (size_t)lr * g.ldb + n0 + (lane & 15) * 4;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nk = g.K / BK;
    const int kl = lane >> 5, cl = lane & 31;
    dma_issue<BK, NS>(lds, ga, gb, g.lda, g.ldb, 0, 0, wave);
    if (nk > 1) dma_issue<BK, NS>(lds, ga, gb, g.lda, g.ldb, 1, 1, wave);
    if (EXTRA != 0) {
        __builtin_amdgcn_sched_barrier(0);                 // the wait counts above depend on this issue order
        side();
        __builtin_amdgcn_sched_barrier(0);
    }
    for (int kc = 0; kc < nk; ++kc) {
        // chunk kc has landed when at most the loads of chunk kc + 1 are outstanding
        if (EXTRA != 0 && kc < 2) {
            if (kc + 1 < nk) __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * (BK / 16) + EXTRA));
            else __builtin_amdgcn_s_waitcnt(0x0F70 | EXTRA);
        } else {
            if (kc + 1 < nk) __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * (BK / 16)));
            else __builtin_amdgcn_s_waitcnt(0x0F70);
        }
        // bare s_barrier: __syncthreads() carries a workgroup fence that drains EVERY outstanding load (vmcnt(0)), i.e. also the
        // chunk that was requested one iteration ago; the explicit count above is the only wait this pipeline needs
        __builtin_amdgcn_s_barrier();                      // ... for every wave; and everyone is done with chunk kc - 1
        const int st = kc % NS;
        float af[BK / 2], bf[BK / 2];
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            af[kk] = lds.A[st][2 * kk + kl][wm + cl];
            bf[kk] = lds.B[st][2 * kk + kl][wn + cl];
        }
        if (kc + 2 < nk) dma_issue<BK, NS>(lds, ga, gb, g.lda, g.ldb, kc + 2, (kc + 2) % NS, wave);    // stage (kc + 2) % NS == (kc - 1) % NS for NS = 3: free since the barrier
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk], bf[kk], acc, 0, 0, 0);
    }
    __syncthreads();
}

template <int BK, int NS, bool PAD>
__global__ void __launch_bounds__(256) gemm_tn_dma_kernel(const GemmArgs g)
{
    constexpr int P = 64 + 4;                              // pitch of the output tile in LDS (floats)
    __shared__ union {
        Ring<BK, 64, 64, NS> ring;
        float T[64 * P];
        char at_most_4_per_cu[PAD ? 36 * 1024 : 4];        // 160 KB / 36 KB: with 6 (24 KB) the dispatcher packs some CUs and starves others
    } lds;
    static_assert(sizeof(Ring<BK, 64, 64, NS>) >= sizeof(float) * 64 * P, "the output tile reuses the ring");
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    f32x16 acc;
    mainloop_dma<BK, NS>(g, lds.ring, m0, n0, wm, wn, acc);                // ends with a barrier: the ring is free
    // accumulator layout (one column per lane) -> row-major through LDS: 16-byte stores, 16 lanes per 256-byte row segment
    // (dword stores straight from the accumulators cost 2.7 us of the 22.4 at 2048 x 2048 x 256)
#pragma unroll
    for (int r = 0; r < 16; ++r) lds.T[(wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * P + wn + (lane & 31)] = acc[r];
    __syncthreads();
    if (g.probe & 1) return;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int q = (int)threadIdx.x + 256 * u, row = q >> 4, c4 = (q & 15) * 4;
        if (m0 + row < g.M)
            *reinterpret_cast<f32x4_t *>(g.C + (size_t)(m0 + row) * g.ldc + n0 + c4) = *reinterpret_cast<const f32x4_t *>(&lds.T[row * P + c4]);
    }
}

template <int BK, int NS, bool PAD = false>
int launch_dma(const GemmArgs &g, hipStream_t st)
{
    if (g.K % BK || g.N % 64 || g.M < 4 || g.ldc % 4 || (reinterpret_cast<uintptr_t>(g.C) & 15u))
        return fail(SGMCMC_EINVAL, "gemm_tn: K must be a multiple of the variant's chunk, C 16-byte aligned with ldc %% 4 == 0");
    hipLaunchKernelGGL((gemm_tn_dma_kernel<BK, NS, PAD>), dim3(g.N / 64, (g.M + 63) / 64), dim3(256), 0, st, g);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "launch gemm_tn_dma");
}

template <int TM, int TN, int WM, int WN, int BK, int SUB = 4, bool FENCE = false>
__global__ void __launch_bounds__(64 * WM * WN) gemm_tn_kernel(const GemmArgs g)
{
    typedef Tile<TM, TN, WM, WN, BK, SUB, FENCE> TL;
    const int m0 = blockIdx.y * TL::BM, n0 = blockIdx.x * TL::BN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave / WN) * 32 * TM, wn = (wave % WN) * 32 * TN;
    __shared__ Stages<TL> lds;
    if (g.phase_counters != nullptr) {
        const int late = (cu_arrival_slot(g.phase_counters) & 3) * g.phase_sleep;
        for (int i = 0; i < late; ++i) __builtin_amdgcn_s_sleep(1);          // 64 cycles each
    }
    f32x16 acc[TM][TN];
    mainloop<TL, TM, TN>(g, lds, m0, n0, wm, wn, acc);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int col = n0 + wn + 32 * j + (lane & 31);
                if (row < g.M) g.C[(size_t)row * g.ldc + col] = acc[i][j][r];
            }
}

// ------------------------------------------------------------------------------------------------------------------
// fused: weight-gradient GEMM + frozen SGHMC update of [W | the parameters that follow W in the arena]
// ------------------------------------------------------------------------------------------------------------------

struct FusedArgs {
    GemmArgs g;                        // g.C: nullable, receives gW (tests); the product is [M][N] with ldc = N
    float *theta, *V;                  // the layer's slice of the arena: W [M][N] dense, then n_tail more parameters
    const float *minv;
    const float *grad_tail;            // gradient of those n_tail parameters (bias gradient etc.), already computed
    size_t n_tail;
    float e2, c1, c3, e4, mdecay, grad_decay;
    NoiseKey nk;                       // nk.q0 = global quad index of W[0][0] within the chain's parameter vector
    double *stats;                     // nullable: one {sum theta'^2, 0, 0, 0} record per workgroup
    unsigned rec_base, rec_total;
    int n_gemm_blocks;                 // persistent workgroups of the product; blocks beyond them update the tail
    int *phase_counters;               // nullable: 2048 ints, one per CU (see the kernel): every second workgroup a CU receives starts late
    int phase_sleeps;                  // ... by this many s_sleep(127) (8128 cycles each)
};

template <int CTRL>
__device__ __forceinline__ float quad_bcast(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

// 64 x 64 output tile, 4 waves of 32 x 32, K in chunks of 16: a workgroup walks over tiles blockIdx.x, + gridDim, ...
// (persistent), so the asynchronous theta'/V' stores of one tile drain under the MFMAs of the next.
typedef Tile<1, 1, 2, 2, 16> FT;

// 256-lane blocks for the n_tail parameters after W (one quad per lane per trip; at least one block if there are any)
inline unsigned tail_blocks_for(size_t n_tail)
{
    if (n_tail == 0) return 0;
    const size_t blocks = (n_tail / 4 + 255) / 256;
    return (unsigned)(blocks ? blocks : 1);
}

// One 64 x 64 tile: product, then the update of the tile. The accumulators go through LDS so that the update runs on
// ROW-MAJOR QUADS exactly like the streaming kernel K1: a lane owns 4 consecutive
// columns of a row = one Philox quad and one 16-byte access per array (16 lanes cover a 256-byte row segment), the arithmetic
// is SghmcOp::compute itself. (A first version kept the MFMA accumulator layout -- one column per lane, dword accesses, a
// 4 x 4 DPP transpose of the normals over each lane quad: 32.8 us at 2048 x 2048 against this version's figure in
// profiles/r03_gemm_fusion_probe.txt.) Rows beyond M (last tile of a ragged layer) are skipped.
constexpr int TP = FT::BN + 4;                             // LDS pitch of the accumulator tile (floats)

template <int NS>
union FusedLds {
    Ring<16, FT::BM, FT::BN, NS> ring;                     // NS = 3: 24 KB; the accumulator tile (17 KB) reuses it after the K loop
    float T[FT::BM * TP];
};

typedef SghmcOp<float, false, false> FusedOp;

// the theta/V/minv quads of quads [U0, U1) of a lane's 4 quads of the tile (3 loads of 16 bytes each)
template <int U0, int U1>
struct FusedStateLoads {
    const FusedOp &op;
    FusedOp::Regs (&R)[4];
    const unsigned (&qg)[4];
    __device__ __forceinline__ void operator()() const
    {
#pragma unroll
        for (int u = U0; u < U1; ++u) {
            load_quad<false>(op.theta, qg[u], R[u].th);
            load_quad<false>(op.V, qg[u], R[u].v);
            load_quad<false>(op.minv, qg[u], R[u].mi);
        }
    }
};

template <int NS, int PRE>
__device__ __forceinline__ void fused_tile(const FusedArgs &a, FusedLds<NS> &lds, const FusedOp &op, int m0, int n0,
                                           int wm, int wn, int lane, float &tsq)
{
    float *T = lds.T;
    const unsigned N = (unsigned)a.g.N;
    constexpr int QPR = FT::BN / 4;                        // quads per tile row
    FusedOp::Regs R[4];
    unsigned qg[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int q = (int)threadIdx.x + 256 * u, row = q / QPR, c4 = (q % QPR) * 4;
        ok[u] = m0 + row < a.g.M;
        qg[u] = ((unsigned)(m0 + (ok[u] ? row : 0)) * N + (unsigned)(n0 + c4)) >> 2;          // quad index within W
    }
    f32x16 acc;
    // the state of the first PRE quads is requested before the K loop and arrives under it, the rest right after the loop
    mainloop_dma<16, NS, 3 * PRE>(a.g, lds.ring, m0, n0, wm, wn, acc, FusedStateLoads<0, PRE>{op, R, qg});   // ends with a barrier: the ring is free
    FusedStateLoads<PRE, 4>{op, R, qg}();
#pragma unroll
    for (int r = 0; r < 16; ++r)
        T[(wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * TP + wn + (lane & 31)] = acc[r];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int q = (int)threadIdx.x + 256 * u, row = q / QPR, c4 = (q % QPR) * 4;
        const f32x4_t gq = *reinterpret_cast<const f32x4_t *>(&T[row * TP + c4]);
        R[u].gr[0] = gq.x; R[u].gr[1] = gq.y; R[u].gr[2] = gq.z; R[u].gr[3] = gq.w;
        if (a.g.C != nullptr && ok[u]) store_quad<false>(a.g.C, qg[u], R[u].gr);
        op.compute(qg[u], R[u]);
        if (ok[u]) {
            store_quad<false>(op.theta, qg[u], R[u].th);
            store_quad<false>(op.V, qg[u], R[u].v);
            tsq += ((R[u].th[0] * R[u].th[0] + R[u].th[1] * R[u].th[1]) + R[u].th[2] * R[u].th[2]) + R[u].th[3] * R[u].th[3];
        }
    }
    __syncthreads();                                       // T = the ring of the next tile's K loop
}

template <int NS, int PRE, int OCC>
__global__ void __launch_bounds__(256, OCC) gemm_tn_sghmc_kernel(const FusedArgs a)
{
    __shared__ FusedLds<NS> lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    NoiseKey nk = a.nk;
    nk.resolve();
    float tsq = 0.f;                                       // this lane's share of sum theta'^2
    SghmcOp<float, false, false> op{a.theta, a.V, nullptr, nullptr, nullptr, nullptr, const_cast<float *>(a.minv), nullptr, nullptr,
                                    a.e2, a.c1, a.c3, a.e4, a.mdecay, a.grad_decay, nk, nullptr};
    if ((int)blockIdx.x >= a.n_gemm_blocks) {
        // ---- the parameters after W (bias, ...): the streaming update K1 on their quads, same Philox stream
        const size_t base = (size_t)a.g.M * a.g.N;         // multiple of 4 (N % 128 == 0)
        op.theta += base; op.V += base; op.minv += base; op.grad = a.grad_tail;
        op.nk.q0 = nk.q0 + base / 4;
        const size_t nq_full = a.n_tail / 4;
        const int tail = (int)(a.n_tail % 4);
        const size_t G = (size_t)(gridDim.x - a.n_gemm_blocks) * blockDim.x;
        const size_t gid = (size_t)(blockIdx.x - a.n_gemm_blocks) * blockDim.x + tid;
        float acc4[4] = {0.f, 0.f, 0.f, 0.f};
        for (size_t q = gid; q < nq_full; q += G) {
            SghmcOp<float, false, false>::Regs R;
            op.load_vec<false>(q, R);
            op.compute(q, R);
            op.store_vec<false>(q, R);
            op.accumulate<true>(R, 4, acc4);
        }
        if (tail && gid == G - 1) {
            SghmcOp<float, false, false>::Regs R;
            op.load_part_(nq_full, tail, R);
            op.compute(nq_full, R);
            op.store_part_(nq_full, tail, R);
            op.accumulate<true>(R, tail, acc4);
        }
        tsq = acc4[0];
    } else {
        const int tiles_n = a.g.N / FT::BN, tiles_m = (a.g.M + FT::BM - 1) / FT::BM;
        const int n_tiles = tiles_m * tiles_n;
        const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
        if (a.phase_counters != nullptr) {
            // De-phasing. Workgroups that share a CU and start together stay in lockstep -- all in the product (matrix cores
            // busy, HBM idle), then all in the update (HBM busy, matrix cores idle). Every second workgroup that arrives on a
            // CU (counted per hardware CU id) therefore starts one update-phase late, so that one half updates while the other
            // half multiplies.
            __shared__ int late;
            if (tid == 0) {
                const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
                const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // HW_REG_XCC_ID [3:0]
                const unsigned key = ((xcc & 7u) << 8) | (((hw >> 13) & 7u) << 5) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u);
                late = atomicAdd(&a.phase_counters[key], 1) & 1;
            }
            __syncthreads();
            if (late)
                for (int i = 0; i < a.phase_sleeps; ++i) __builtin_amdgcn_s_sleep(127);
        }
        for (int t = blockIdx.x; t < n_tiles; t += a.n_gemm_blocks) {
            const int m0 = (t / tiles_n) * FT::BM, n0 = (t % tiles_n) * FT::BN;
            fused_tile<NS, PRE>(a, lds, op, m0, n0, wm, wn, lane, tsq);
        }
    }
    if (a.stats != nullptr) {
        __shared__ float red[4];
        const float w = wave_sum_dpp_lane63(tsq);
        if (lane == 63) red[wave] = w;
        __syncthreads();
        if (tid < 4) {
            double val = 0.0;
            if (tid == 0) val = (((double)red[0] + (double)red[1]) + (double)red[2]) + (double)red[3];
            a.stats[4 + 4 * ((size_t)a.rec_base + blockIdx.x) + tid] = val;
        }
        if (blockIdx.x == 0 && tid == 0)
            reinterpret_cast<unsigned long long *>(a.stats)[0] = a.rec_total ? a.rec_total : gridDim.x;
    }
}

template <int TM, int TN, int WM, int WN, int BK, int SUB = 4, bool FENCE = false>
int launch_gemm(const GemmArgs &g, hipStream_t st)
{
    typedef Tile<TM, TN, WM, WN, BK, SUB, FENCE> TL;
    if (g.K % BK) return fail(SGMCMC_EINVAL, "gemm_tn: K must be a multiple of the variant's chunk");
    hipLaunchKernelGGL((gemm_tn_kernel<TM, TN, WM, WN, BK, SUB, FENCE>), dim3(g.N / TL::BN, (g.M + TL::BM - 1) / TL::BM), dim3(TL::NT), 0, st, g);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "launch gemm_tn");
}

}  // namespace

extern "C" {

/* C[M][N] = A^T B with A [K][lda] (M columns used), B [K][ldb] (N columns used): the weight-gradient product of a dense
 * layer (A = input activations, B = deltas), fp32 on the matrix cores. N % 128 == 0, K % 16 == 0, M % 4 == 0.        */
int sgmcmc_gemm_tn_f32(const float *A, const float *B, float *C, int M, int N, int K, int lda, int ldb, int ldc,
                       int variant, int *phase_counters, int phase_sleep, sgmcmc_stream_t stream)
{
    if (!A || !B || !C) return fail(SGMCMC_EINVAL, "gemm_tn: NULL argument");
    if (M <= 0 || N <= 0 || K <= 0 || N % 128 || K % 16 || M % 4 || lda < M || ldb < N || ldc < N || lda % 4 || ldb % 4 ||
        ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15u))
        return fail(SGMCMC_EINVAL, "gemm_tn: needs N %% 128 == 0, K %% 16 == 0, M %% 4 == 0, 16-byte aligned operands");
    GemmArgs g{A, B, C, M, N, K, lda, ldb, ldc};
    g.probe = ((variant >> 8) & 1) | (((variant >> 10) & 3) << 1);
    g.phase_counters = phase_counters; g.phase_sleep = phase_sleep;
    if ((variant >> 9) & 1) g.K = 64;                     // probe: a quarter of the K loop
    variant &= 0xff;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (variant) {
    case 0: return launch_dma<16, 3, true>(g, st);                  //  64 x  64, 4 waves of 32 x 32, direct-to-LDS loads, K chunks of 16, ring of 3 (the fused kernel's K loop)
    case 1: return launch_gemm<1, 1, 2, 2, 32>(g, st);          //                                              32
    case 2: return launch_gemm<1, 1, 2, 2, 64>(g, st);          //                                              64
    case 3: return launch_gemm<1, 2, 4, 2, 32>(g, st);          // 128 x 128, 8 waves of 32 x 64
    case 4: return launch_gemm<1, 2, 4, 2, 64>(g, st);
    case 5: return launch_gemm<2, 2, 2, 2, 32>(g, st);          // 128 x 128, 4 waves of 64 x 64
    case 6: return launch_gemm<1, 2, 2, 2, 32>(g, st);          //  64 x 128, 4 waves of 32 x 64
    case 7: return launch_gemm<2, 1, 2, 2, 32>(g, st);          // 128 x  64, 4 waves of 64 x 32
    case 8: return launch_gemm<1, 1, 2, 2, 32, 4, true>(g, st); //  64 x  64 with the scheduling fences
    case 9: return launch_gemm<1, 1, 2, 2, 16>(g, st);          //  64 x  64, operands staged through registers, K chunks of 16
    case 10: return launch_dma<32, 3>(g, st);                   //  direct-to-LDS, chunks of 32
    case 11: return launch_dma<16, 4>(g, st);                   //  direct-to-LDS, ring of 4
    case 12: return launch_dma<16, 3, false>(g, st);            //  variant 0 without its LDS padding (6 workgroups per CU fit: 24.7 us instead of 22.0)
    default: return fail(SGMCMC_EINVAL, "gemm_tn: unknown variant");
    }
}

/* see include/sgmcmc_hip.h */
int sgmcmc_gemm_tn_sghmc_f32(const float *A, const float *B, int M, int N, int K, int lda, int ldb, float *theta, float *V,
                             const float *minv, const float *grad_tail, size_t n_tail, float *grad_out, float eps,
                             float scale_grad, float mdecay, float grad_decay, uint64_t seed, uint64_t step,
                             const uint64_t *step_dev, uint64_t first_element, void *stats_ws, uint32_t stats_record_base,
                             uint32_t stats_record_total, int gemm_blocks, int *phase_counters, int phase_sleeps,
                             sgmcmc_stream_t stream)
{
    if (!A || !B || !theta || !V || !minv || (n_tail && !grad_tail)) return fail(SGMCMC_EINVAL, "gemm_tn_sghmc: NULL argument");
    if (M <= 0 || N <= 0 || K <= 0 || N % 128 || K % FT::BK || M % 4 || lda < M || ldb < N || lda % 4 || ldb % 4 ||
        ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15u))
        return fail(SGMCMC_EINVAL, "gemm_tn_sghmc: needs N %% 128 == 0, K %% 16 == 0, M %% 4 == 0, 16-byte aligned operands");
    if (first_element % 4 || ((reinterpret_cast<uintptr_t>(theta) | reinterpret_cast<uintptr_t>(V) |
                               reinterpret_cast<uintptr_t>(minv) | reinterpret_cast<uintptr_t>(grad_tail)) & 15u))
        return fail(SGMCMC_EINVAL, "gemm_tn_sghmc: the slice must start on a quad (first_element %% 4 == 0, 16-byte aligned arrays)");
    FusedArgs a;
    a.g = GemmArgs{A, B, grad_out, M, N, K, lda, ldb, N};
    a.theta = theta; a.V = V; a.minv = minv; a.grad_tail = grad_tail; a.n_tail = n_tail;
    // derived scalars exactly as sgmcmc_sghmc_step_f32 forms them (sghmc.py:111-117,211-217,235)
    const float eps_s = eps / std::sqrt(scale_grad);
    a.e2 = std::pow(eps, 2.0f);
    a.c1 = (2.0f * std::pow(eps_s, 2.0f)) * mdecay;
    a.c3 = 2.0f * std::pow(eps_s, 3.0f);
    a.e4 = std::pow(eps_s, 4.0f);
    a.mdecay = mdecay; a.grad_decay = grad_decay;
    a.nk.k0 = (uint32_t)seed; a.nk.k1 = (uint32_t)(seed >> 32);
    a.nk.s0 = (uint32_t)step; a.nk.s1 = (uint32_t)(step >> 32);
    a.nk.step_dev = step_dev; a.nk.q0 = first_element / 4;
    a.stats = static_cast<double *>(stats_ws); a.rec_base = stats_record_base; a.rec_total = stats_record_total;
    const int n_tiles = ((M + FT::BM - 1) / FT::BM) * (N / FT::BN);
    const int flavour = gemm_blocks > 0 ? gemm_blocks >> 16 : 0;          // probe: see the header
    gemm_blocks &= 0xffff;
    if (gemm_blocks <= 0) gemm_blocks = 1024;             // 4 workgroups per CU
    a.n_gemm_blocks = gemm_blocks < n_tiles ? gemm_blocks : n_tiles;
    a.phase_counters = phase_counters; a.phase_sleeps = phase_sleeps;
    const dim3 grid(a.n_gemm_blocks + tail_blocks_for(n_tail));
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (flavour) {
    case 0: hipLaunchKernelGGL((gemm_tn_sghmc_kernel<3, 0, 4>), grid, dim3(256), 0, st, a); break;   // state loaded after the K loop
    case 1: hipLaunchKernelGGL((gemm_tn_sghmc_kernel<3, 2, 4>), grid, dim3(256), 0, st, a); break;   // 2 of a lane's 4 quads requested before it
    case 2: hipLaunchKernelGGL((gemm_tn_sghmc_kernel<3, 4, 3>), grid, dim3(256), 0, st, a); break;   // all 4 (3 workgroups per CU)
    default: return fail(SGMCMC_EINVAL, "gemm_tn_sghmc: unknown kernel flavour");
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "launch gemm_tn_sghmc");
}

/* number of workgroups (= statistics records) sgmcmc_gemm_tn_sghmc_f32 launches for these sizes */
int sgmcmc_gemm_tn_sghmc_blocks(int M, int N, size_t n_tail, int gemm_blocks)
{
    const int n_tiles = ((M + FT::BM - 1) / FT::BM) * (N / FT::BN);
    gemm_blocks = gemm_blocks > 0 ? gemm_blocks & 0xffff : 0;
    if (gemm_blocks <= 0) gemm_blocks = 1024;
    const int gb = gemm_blocks < n_tiles ? gemm_blocks : n_tiles;
    return gb + (int)tail_blocks_for(n_tail);
}

}  // extern "C"
