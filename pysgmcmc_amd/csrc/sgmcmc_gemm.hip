// sgmcmc_gemm.hip -- the weight-gradient GEMM of a dense layer on the fp32 matrix cores, with the SGHMC update of that
// layer's weights as its epilogue.
//
//   gW[m][n] = sum_k A[k][m] * B[k][n]        A = the layer's input activations h_{l-1}  [K = batch][M = fan_in]
//                                             B = the layer's back-propagated deltas     [K = batch][N = fan_out]
//
// is the last thing the backward pass computes for a layer, and the ONLY consumer of gW is the sampler's update of W
// (pysgmcmc/samplers/sghmc.py:211-251). Fused form: the tile of gW a workgroup has just accumulated in its MFMA
// accumulators never goes to HBM -- the workgroup loads the same tile of theta, V and minv (prefetched under the K loop),
// draws the tile's Philox normals (the stream of the streaming kernel K1: counter = (step, global quad index)), applies
// the frozen SGHMC update arithmetic (SghmcOp, one IEEE rounding per reference op) and writes theta', V'. Per parameter
// that is 20 B of HBM traffic instead of 4 (GEMM writes gW) + 24 (K1), and the HBM-bound update hides under the
// matrix-core-bound product instead of running after it (second-stream and any-order overlap do not work on this
// stack: profiles/r03_overlap_probe.txt).
//
// Tiling for gfx950: 128 x 128 output tile per 256-lane workgroup (4 waves in 2 x 2, each 64 x 64 = 2 x 2
// v_mfma_f32_32x32x2_f32 tiles, 64 accumulator registers), K in chunks of 16 through a double-buffered LDS stage
// (2 x 16 KB). Operand fragments are single floats per lane (A[m = lane & 31][k = lane >> 5], B[k][n = lane & 31]) read
// from k-major LDS rows: conflict-free ds_read_b32. The accumulator map (row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5),
// col = lane & 31) makes every accumulator register of a wave two 128-byte row segments: dword accesses of the epilogue
// are fully coalesced. A Philox quad = 4 consecutive parameters = the same row in 4 adjacent lanes; lane t of each lane
// quad draws the quad of row t and a 4 x 4 DPP transpose hands every lane its own column.
//
// fp32 MFMA is an exact fmaf chain in k order (MI355X_MICROARCH.md): the product differs from a library GEMM only in
// summation order. The update arithmetic is bit-identical to K1 applied to the same gW (tests write gW out and check).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "sgmcmc_hip.h"

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
};

// One workgroup = WM x WN waves, each wave owns TM x TN MFMA tiles of 32 x 32: block tile BM = 32 TM WM by BN = 32 TN WN.
// K advances in chunks of BK through a double-buffered LDS stage; inside a chunk the MFMAs run in sub-batches of 8 k-steps
// whose operand fragments are read from LDS one sub-batch ahead.
template <int TM, int TN, int WM, int WN, int BK_>
struct Tile {
    static constexpr int BK = BK_;
    static constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, NT = 64 * WM * WN;
    static constexpr int LA = BK * BM / 4 / NT, LB = BK * BN / 4 / NT;      // float4 loads per lane per chunk
    static_assert(LA >= 1 && LB >= 1 && LA * NT * 4 == BK * BM && LB * NT * 4 == BK * BN, "chunk must divide over the lanes");
};

typedef float f32x4_t __attribute__((ext_vector_type(4)));

// acc[i][j] = 32 x 32 tile (rows +32 i, cols +32 j) of this wave's part of the block tile at (m0, n0)
template <typename TL, int TM, int TN>
__device__ __forceinline__ void mainloop(const GemmArgs &g, int m0, int n0, int wm, int wn, f32x16 (&acc)[TM][TN])
{
    constexpr int BK = TL::BK, BM = TL::BM, BN = TL::BN, NT = TL::NT, LA = TL::LA, LB = TL::LB;
    constexpr int SUB = 8, NSUB = BK / 2 / SUB;             // k-steps (of 2) per sub-batch, sub-batches per chunk
    static_assert(NSUB >= 1 && NSUB * SUB * 2 == BK, "BK must be a multiple of 16");
    __shared__ float As[2][BK][BM];
    __shared__ float Bs[2][BK][BN];
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
            *reinterpret_cast<f32x4_t *>(&As[s][r][c]) = ra[u];
        }
#pragma unroll
        for (int u = 0; u < LB; ++u) {
            const int f = tid + u * NT, r = f / (BN / 4), c = (f % (BN / 4)) * 4;
            *reinterpret_cast<f32x4_t *>(&Bs[s][r][c]) = rb[u];
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
            for (int i = 0; i < TM; ++i) af[buf][kk][i] = As[s][2 * (sub * SUB + kk) + kl][wm + 32 * i + cl];
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[buf][kk][j] = Bs[s][2 * (sub * SUB + kk) + kl][wn + 32 * j + cl];
        }
    };
    for (int kc = 0; kc < nk; ++kc) {
        const int s = kc & 1;
        frags(s, 0, 0);
        if (kc + 1 < nk) fetch(kc + 1);                    // global loads of the next chunk fly under this chunk's MFMAs
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            if (sub + 1 < NSUB) frags(s, sub + 1, (sub + 1) & 1);   // next sub-batch's fragments are requested first ...
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < SUB; ++kk)                         // ... and land under these MFMAs
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[sub & 1][kk][i], bf[sub & 1][kk][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (kc + 1 < nk) stash(s ^ 1);                     // stage s ^ 1 was last read before the previous barrier
        __syncthreads();
    }
}

template <int TM, int TN, int WM, int WN, int BK>
__global__ void __launch_bounds__(64 * WM * WN) gemm_tn_kernel(const GemmArgs g)
{
    typedef Tile<TM, TN, WM, WN, BK> TL;
    const int m0 = blockIdx.y * TL::BM, n0 = blockIdx.x * TL::BN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave / WN) * 32 * TM, wn = (wave % WN) * 32 * TN;
    f32x16 acc[TM][TN];
    mainloop<TL, TM, TN>(g, m0, n0, wm, wn, acc);
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

template <int TM, int TN, int WM, int WN, int BK>
int launch_gemm(const GemmArgs &g, hipStream_t st)
{
    typedef Tile<TM, TN, WM, WN, BK> TL;
    if (g.K % BK) return fail(SGMCMC_EINVAL, "gemm_tn: K must be a multiple of the variant's chunk");
    hipLaunchKernelGGL((gemm_tn_kernel<TM, TN, WM, WN, BK>), dim3(g.N / TL::BN, (g.M + TL::BM - 1) / TL::BM), dim3(TL::NT), 0, st, g);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "launch gemm_tn");
}

}  // namespace

extern "C" {

/* C[M][N] = A^T B with A [K][lda] (M columns used), B [K][ldb] (N columns used): the weight-gradient product of a dense
 * layer (A = input activations, B = deltas), fp32 on the matrix cores. N % 128 == 0, K % 16 == 0, M % 4 == 0.        */
int sgmcmc_gemm_tn_f32(const float *A, const float *B, float *C, int M, int N, int K, int lda, int ldb, int ldc,
                       int variant, sgmcmc_stream_t stream)
{
    if (!A || !B || !C) return fail(SGMCMC_EINVAL, "gemm_tn: NULL argument");
    if (M <= 0 || N <= 0 || K <= 0 || N % 128 || K % 16 || M % 4 || lda < M || ldb < N || ldc < N || lda % 4 || ldb % 4 ||
        ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15u))
        return fail(SGMCMC_EINVAL, "gemm_tn: needs N %% 128 == 0, K %% 16 == 0, M %% 4 == 0, 16-byte aligned operands");
    GemmArgs g{A, B, C, M, N, K, lda, ldb, ldc};
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (variant) {
    case 0: return launch_gemm<1, 2, 4, 2, 16>(g, st);    // 128 x 128, 8 waves of 32 x 64, K chunks of 16
    case 1: return launch_gemm<1, 2, 4, 2, 32>(g, st);    //                                          ... of 32
    case 2: return launch_gemm<1, 2, 4, 2, 64>(g, st);    //                                          ... of 64
    case 3: return launch_gemm<2, 2, 2, 2, 32>(g, st);    // 128 x 128, 4 waves of 64 x 64, 32
    case 4: return launch_gemm<2, 2, 2, 2, 64>(g, st);    //                                 64
    case 5: return launch_gemm<1, 1, 2, 2, 32>(g, st);    //  64 x  64, 4 waves of 32 x 32, 32
    case 6: return launch_gemm<1, 1, 2, 2, 64>(g, st);    //                                 64
    case 7: return launch_gemm<1, 2, 2, 2, 32>(g, st);    //  64 x 128, 4 waves of 32 x 64, 32
    case 8: return launch_gemm<1, 2, 2, 2, 64>(g, st);    //                                 64
    default: return fail(SGMCMC_EINVAL, "gemm_tn: unknown variant");
    }
}

}  // extern "C"
