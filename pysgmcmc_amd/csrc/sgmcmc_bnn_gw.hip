// sgmcmc_bnn_gw.hip -- weight gradients of the BNN's wide dense layers, gW = h^T delta (what tf.gradients builds for
// pysgmcmc/models/bayesian_neural_network.py:30-56, reached from pysgmcmc/samplers/sghmc.py:121-122), at fp32 accuracy on the bf16
// matrix pipe.
//
// Both operands of this product are ACTIVATIONS of the step ([batch][features]; the batch is the contraction index). Each is
// written once more as three exact bf16 planes
//     x = x0 + x1 + x2,  x0 = top 16 bits of x, x1 = top 16 bits of (x - x0), x2 = x - x0 - x1      (8 + 8 + 8 significant bits, no rounding)
// (sgmcmc_bnn_split_planes_f32), and the product loop is direct-to-LDS loads + six v_mfma_f32_32x32x16_bf16 per 16 batch rows -- the
// partial products of order <= 2^-16: a0 b0 | a0 b1, a1 b0 | a0 b2, a1 b1, a2 b0 -- with NO vector-ALU work. Error against fp64:
// rms 1.2e-7 of the rms value, against 2.9e-7 for the library's fp32 product on the same operands (the a0 b0 sums and the five small
// products have accumulators of their own); accumulation order over the batch is fixed: bit-reproducible.
//
// Where it pays (profiles/r06_gw_gate.txt, r06_gw_planes_step.txt): the matrix pipe clocks down under these MFMAs (1.7-2.0 GHz) and
// a launch has ~3.5 us of ramp, so the two 2048 x 2048 products of the 10 M-parameter net take 26 us against 33 for the library --
// nothing once the planes have to be written -- but the two 4864 x 4864 products of configs[4] take 131 us against 188. BNNCost's
// plan selects it by tile count.
//
// Plane layout: plane p of X [M][N] is P[p][m / 8][n][m % 8] (bf16): 16 bytes = the 8 batch rows one lane feeds one MFMA for feature
// n, so a fragment is ONE conflict-free ds_read_b128 per lane and a direct-to-LDS piece (64 lanes x 16 B) is 1 KiB contiguous in
// memory and in LDS.
// Decomposition: 128 x 128 output tile per workgroup of 4 waves (64 x 64 = 2 x 2 MFMA tiles each), up to 3 workgroups per CU;
// a chunk = 16 batch rows = ONE MFMA k-step = 24 KiB of planes (2 operands x 3 planes x 2 x 128 x 16 B); ring of 2 stages.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "sgmcmc_hip.h"

#pragma clang fp contract(off)

#include "sgmcmc_host.hpp"

using namespace sgmcmc_host;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int TILE = 128;
[[maybe_unused]] constexpr int KC = 16, STAGE_BYTES = 24 * 1024, NS = 2;

struct GwArgs {
    const unsigned char *A, *B;         // plane sets of product 0: layer inputs [M][nA] (rows of gW), deltas [M][nB] (columns)
    float *C;                           // gradient of product 0 [nA][nB], pitch ldc (a slice of the gradient arena)
    size_t a_stride, b_stride;          // bytes between the plane sets of consecutive products
    size_t c_stride;                    // elements between their gradients
    int nA, nB, M, ldc;
    unsigned plane_a_bytes, plane_b_bytes;      // bytes between the planes of a set (M * n * 2)
    int tiles_i, tiles_j;
};

[[maybe_unused]] constexpr int vmcnt_imm(int n) { return (n & 15) | (7 << 4) | (15 << 8) | ((n >> 4) << 14); }
template <int N>
__device__ __forceinline__ void wait_vm()
{
    __builtin_amdgcn_s_waitcnt(vmcnt_imm(N));
}

__global__ void __launch_bounds__(256, 2) gw_bf16x3_kernel(const GwArgs g)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NS][STAGE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 1, wj = wave & 1;
    const int per = g.tiles_i * g.tiles_j;
    int id = blockIdx.x, z, ti, tj;
    {
        // workgroup b runs on XCD b % 8: XCD x gets the CONTIGUOUS range of tile numbers [start_x, start_x + count_x) ...
        const int T = (int)gridDim.x, q8 = T >> 3, r8 = T & 7, x = id & 7;
        id = x * q8 + (x < r8 ? x : r8) + (id >> 3);
        z = id / per;
        // ... and tile numbers walk a gradient in panels of 8 tile rows, rows fastest: 64 consecutive numbers = an 8 x 8 block of
        // tiles = 16 operand slices of 64 KiB x 3 planes in the XCD's L2
        const int r = id - z * per, panel = 8 * g.tiles_j, gidx = r / panel, rows = (g.tiles_i - 8 * gidx) < 8 ? (g.tiles_i - 8 * gidx) : 8;
        const int w = r - gidx * panel;
        ti = 8 * gidx + w % rows;
        tj = w / rows;
    }
    const int i0 = ti * TILE, j0 = tj * TILE;
    const int nk = g.M / KC;
    // ---- direct loads: a chunk is 24 pieces of 1 KiB (operand, plane, m8 of the chunk, half of the tile's 128 features); wave w
    // requests pieces 3 w .. 3 w + 2 of A and of B
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char *>(g.A + (size_t)z * g.a_stride), 0, (int)(3u * g.plane_a_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char *>(g.B + (size_t)z * g.b_stride), 0, (int)(3u * g.plane_b_bytes), 0x00020000);
    // lanes whose feature lies beyond the operand must not read the next m8 row: the range check covers the per-lane offset only
    // (the scalar offset is excluded from it), so those lanes get an offset beyond the buffer and load zeros
    const unsigned big = 0x7ffffff0u;
    unsigned soff_a[3], soff_b[3], vo_a[3], vo_b[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int q = 3 * wave + u, p = q >> 2, m8 = (q >> 1) & 1, hf = q & 1;
        soff_a[u] = (unsigned)p * g.plane_a_bytes + (unsigned)(m8 * g.nA + i0 + 64 * hf) * 16u;
        soff_b[u] = (unsigned)p * g.plane_b_bytes + (unsigned)(m8 * g.nB + j0 + 64 * hf) * 16u;
        vo_a[u] = (i0 + 64 * hf + lane < g.nA) ? (unsigned)lane * 16u : big;
        vo_b[u] = (j0 + 64 * hf + lane < g.nB) ? (unsigned)lane * 16u : big;
    }
    const unsigned chunk_a = 2u * (unsigned)g.nA * 16u, chunk_b = 2u * (unsigned)g.nB * 16u;
    auto issue = [&](int c, int st) {
#pragma unroll
        for (int u = 0; u < 3; ++u)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, &lds[st][(3 * wave + u) * 1024], 16, vo_a[u], soff_a[u] + (unsigned)c * chunk_a, 0, 0);
#pragma unroll
        for (int u = 0; u < 3; ++u)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, &lds[st][(12 + 3 * wave + u) * 1024], 16, vo_b[u], soff_b[u] + (unsigned)c * chunk_b, 0, 0);
    };
    // ---- fragments: lane (fm, kh) holds batch rows 8 kh .. 8 kh + 7 of the chunk for feature fm of a 32-block
    const int fm = lane & 31, kh = lane >> 5;
    const int offA = (kh * 128 + wi * 64 + fm) * 16, offB = 12 * 1024 + (kh * 128 + wj * 64 + fm) * 16;
    f32x16 accm[2][2], accs[2][2];                          // a0 b0 sums | the five small partial products
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) accm[a][b][r] = accs[a][b][r] = 0.f;
    auto chunk_mfmas = [&](int st) {
        const unsigned char *S = &lds[st][0];
        u32x4 fa[2][3], fb[2][3];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                fa[x][p] = *reinterpret_cast<const u32x4 *>(S + offA + p * 4096 + x * 512);
                fb[x][p] = *reinterpret_cast<const u32x4 *>(S + offB + p * 4096 + x * 512);
            }
#define GW_MF(I, J, PA, PB, ACC)                                                                                                    \
    ACC[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[I][PA]), __builtin_bit_cast(bf16x8, fb[J][PB]), \
                                                        ACC[I][J], 0, 0, 0)
#define GW_ALL(PA, PB, ACC) GW_MF(0, 0, PA, PB, ACC); GW_MF(0, 1, PA, PB, ACC); GW_MF(1, 0, PA, PB, ACC); GW_MF(1, 1, PA, PB, ACC)
        GW_ALL(0, 0, accm);
        GW_ALL(0, 2, accs); GW_ALL(2, 0, accs); GW_ALL(1, 1, accs); GW_ALL(0, 1, accs); GW_ALL(1, 0, accs);
#undef GW_ALL
#undef GW_MF
    };
    // ---- ring of two stages: chunk c + 1 is requested when chunk c's stage is about to be multiplied
    issue(0, 0);
    int st = 0;
    for (int c = 0; c < nk; ++c) {
        wait_vm<0>();                                       // this wave's pieces of chunk c have landed ...
        __builtin_amdgcn_s_barrier();                       // ... and every wave's; the stage of chunk c - 1 is free
        if (c + 1 < nk) issue(c + 1, st ^ 1);
        chunk_mfmas(st);
        st ^= 1;
    }
    // ---- epilogue: straight from the accumulators. MFMA tile (x, y): lane (fm, kh) register r holds row 32 x + (r & 3) +
    // 8 (r >> 2) + 4 kh, column 32 y + fm of the wave's 64 x 64: 32 consecutive floats (one 128-byte line) per half wave and store
    float *__restrict__ C = g.C + (size_t)z * g.c_stride;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const int col = j0 + wj * 64 + 32 * y + fm;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wi * 64 + 32 * x + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (row < g.nA && col < g.nB) C[(size_t)row * g.ldc + col] = accm[x][y][r] + accs[x][y][r];
            }
        }
#endif
}

// exact three-way split of two consecutive batch rows (x = row m, y = row m + 1), packed low half = x
struct Pair3 {
    unsigned p0, p1, p2;
};
__device__ __forceinline__ Pair3 split_pair(float x, float y)
{
    const unsigned xb = __float_as_uint(x), yb = __float_as_uint(y);
    const unsigned p0 = __builtin_amdgcn_perm(yb, xb, 0x07060302u);
    const float xr = x - __uint_as_float(xb & 0xffff0000u), yr = y - __uint_as_float(yb & 0xffff0000u);
    const unsigned xrb = __float_as_uint(xr), yrb = __float_as_uint(yr);
    const unsigned p1 = __builtin_amdgcn_perm(yrb, xrb, 0x07060302u);
    const float xl = xr - __uint_as_float(xrb & 0xffff0000u), yl = yr - __uint_as_float(yrb & 0xffff0000u);
    const unsigned p2 = __builtin_amdgcn_perm(__float_as_uint(yl), __float_as_uint(xl), 0x07060302u);
    return {p0, p1, p2};
}

// X [M][N] (pitch ldx) of `count` matrices -> three bf16 planes each. One lane = 8 batch rows of one feature: reads are coalesced
// along the features, the three 16-byte stores too.
__global__ void __launch_bounds__(256) split_planes_kernel(const float *__restrict__ X, size_t x_stride, int M, int N, int ldx,
                                                           unsigned char *__restrict__ planes, size_t planes_stride, unsigned plane_bytes)
{
    const int n = (int)(blockIdx.x * 256u + threadIdx.x), m8 = (int)blockIdx.y, z = (int)blockIdx.z;
    if (n >= N) return;
    const float *__restrict__ x = X + (size_t)z * x_stride;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = x[(size_t)(8 * m8 + j) * ldx + n];
    u32x4 p0, p1, p2;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const Pair3 s = split_pair(v[2 * q], v[2 * q + 1]);
        p0[q] = s.p0; p1[q] = s.p1; p2[q] = s.p2;
    }
    unsigned char *base = planes + (size_t)z * planes_stride + ((size_t)m8 * N + n) * 16u;
    *reinterpret_cast<u32x4 *>(base) = p0;
    *reinterpret_cast<u32x4 *>(base + plane_bytes) = p1;
    *reinterpret_cast<u32x4 *>(base + 2 * (size_t)plane_bytes) = p2;
}

}  // namespace

extern "C" {

/* see include/sgmcmc_hip.h */
size_t sgmcmc_bnn_planes_bytes(int M, int N)
{
    if (M <= 0 || N <= 0 || M % 16) return 0;
    return 3 * (size_t)M * (size_t)N * 2;
}

/* see include/sgmcmc_hip.h */
int sgmcmc_bnn_split_planes_f32(const float *X, int count, size_t x_stride, int M, int N, int ldx, void *planes, size_t planes_stride,
                                sgmcmc_stream_t stream)
{
    if (count <= 0) return 0;
    if (!X || !planes) return fail(SGMCMC_EINVAL, "bnn_split_planes: NULL argument");
    if (M <= 0 || N <= 0 || M % 16 || ldx < N || count > 65535 || (reinterpret_cast<uintptr_t>(planes) & 15u) || (planes_stride & 15u))
        return fail(SGMCMC_EINVAL, "bnn_split_planes: needs M %% 16 == 0, ldx >= N, 16-byte aligned plane sets, at most 65535 matrices");
    const size_t one = sgmcmc_bnn_planes_bytes(M, N);
    if (one >= 2147483648ull || (count > 1 && planes_stride < one))
        return fail(SGMCMC_EINVAL, "bnn_split_planes: a plane set spans 2 GiB or more, or the sets overlap");
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)(M / 8), (unsigned)count), dim3(256), 0,
                       static_cast<hipStream_t>(stream), X, x_stride, M, N, ldx, static_cast<unsigned char *>(planes), planes_stride,
                       (unsigned)(one / 3));
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "launch bnn_split_planes");
}

/* see include/sgmcmc_hip.h */
int sgmcmc_bnn_gw_planes_f32(const void *a_planes, size_t a_stride, const void *b_planes, size_t b_stride, float *C, size_t c_stride,
                             int count, int M, int nA, int nB, int ldc, sgmcmc_stream_t stream)
{
    if (count <= 0) return 0;
    if (!a_planes || !b_planes || !C) return fail(SGMCMC_EINVAL, "bnn_gw_planes: NULL argument");
    if (M <= 0 || nA <= 0 || nB <= 0 || M % 16 || ldc < nB ||
        ((reinterpret_cast<uintptr_t>(a_planes) | reinterpret_cast<uintptr_t>(b_planes) | a_stride | b_stride) & 15u))
        return fail(SGMCMC_EINVAL, "bnn_gw_planes: needs M %% 16 == 0, ldc >= nB, 16-byte aligned plane sets");
    const size_t sa = sgmcmc_bnn_planes_bytes(M, nA), sb = sgmcmc_bnn_planes_bytes(M, nB);
    if (sa >= 2147483648ull || sb >= 2147483648ull) return fail(SGMCMC_EINVAL, "bnn_gw_planes: a plane set spans 2 GiB or more");
    GwArgs g{};
    g.A = static_cast<const unsigned char *>(a_planes); g.B = static_cast<const unsigned char *>(b_planes); g.C = C;
    g.a_stride = a_stride; g.b_stride = b_stride; g.c_stride = c_stride;
    g.nA = nA; g.nB = nB; g.M = M; g.ldc = ldc;
    g.plane_a_bytes = (unsigned)(sa / 3); g.plane_b_bytes = (unsigned)(sb / 3);
    g.tiles_i = (nA + TILE - 1) / TILE; g.tiles_j = (nB + TILE - 1) / TILE;
    const double tiles = (double)count * g.tiles_i * g.tiles_j;
    if (tiles >= 2147483648.0) return fail(SGMCMC_EINVAL, "bnn_gw_planes: too many tiles");
    hipLaunchKernelGGL(gw_bf16x3_kernel, dim3((unsigned)tiles), dim3(256), 0, static_cast<hipStream_t>(stream), g);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "launch bnn_gw_planes");
}

}  // extern "C"
