// bnn_dense_reg.hip -- EXPERIMENT (round 4), not part of libsgmcmc_hip.so: the hidden-layer products of the BNN
// (pysgmcmc/models/bayesian_neural_network.py:30-52) with the operands going from global memory straight into REGISTERS -- no LDS in
// the K loop, no barriers -- against the product's LDS-ring kernel (pysgmcmc_amd/csrc/sgmcmc_bnn_gemm.hip).
//
// Why try: in the ring kernel a 64-deep chunk moves 24 KB INTO LDS (direct loads) and 32 KB OUT of it (fragment reads of 8 waves) per 512
// MFMA cycles = 112 of the 128 B/clk an LDS delivers; and B is not shared inside the workgroup at all (every (column half, K quarter)
// is read by ONE wave), A by two waves only. Here a wave owns the whole 32 x 64 output tile for ITS eighth of every chunk:
//   wave q, chunk c: k in [64 c + 8 q, 64 c + 8 q + 8);  lane (m = lane % 32, kl = lane / 32), MFMA j = 0..3 uses k = 64 c + 8 q + 4 kl + j
//   A [M][K] (k contiguous):  ONE buffer_load_dwordx4 per lane and chunk = A[m][k .. k + 3]
//   B forward  W [K][N] (n contiguous): four buffer_load_dwordx2, load j = W[k_j][2 l', 2 l' + 1]  (l' = lane % 32): accumulator 0 owns the
//     even columns of the tile, accumulator 1 the odd ones
//   B backward W [N][K] (k contiguous): two buffer_load_dwordx4, W[l' + 32 t][k .. k + 3]: accumulator t owns columns 32 t + l'
// i.e. 8 MFMAs per 5 (3) vector-memory instructions, 12 VGPRs per chunk in flight, a register ring of P chunks; the eight K slices meet
// in LDS once, in the epilogue (fixed order). Build: make -C tools/gpu; run: tools/dense_reg_probe.py.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>

#pragma clang fp contract(off)

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

constexpr int BM = 32, BN = 64, BK = 64, NW = 8;
constexpr int TP = BN + 4;

struct Args {
    const float *A, *B, *bias;
    float *out;
    const float *act;           // BWD: tanh outputs of this layer
    int M, N, K, lda, ldb, ldo, ldact;
};

struct __attribute__((aligned(16))) Lds {
    float T[NW][BM][TP];
};

template <bool BWD>
struct Chunk {
    f32x4_t a;
    f32x2_t b2[BWD ? 1 : 4];
    f32x4_t b4[BWD ? 2 : 1];
};

template <int P, bool BWD>
__global__ void __launch_bounds__(512) dense_reg_kernel(const Args g)
{
    __shared__ Lds lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_m = g.M / BM, tiles = tiles_m * (g.N / BN);
    int t = blockIdx.x;
    if (tiles % 8 == 0) t = (t & 7) * (tiles >> 3) + (t >> 3);
    const int n0 = (t / tiles_m) * BN, m0 = (t % tiles_m) * BM;
    const int m = lane & 31, kl = lane >> 5;
    // chunks in which this wave's eight k lie below K (K % 8 == 0: all or none of them)
    const int nkw = (g.K - 8 * q + BK - 1) / BK;
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(g.A + (size_t)m0 * g.lda + 8 * q), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(BWD ? g.B + (size_t)n0 * g.ldb + 8 * q : g.B + (size_t)(8 * q) * g.ldb + n0), 0, 0x7fffffff, 0x00020000);
#endif
    const unsigned a_lane = (unsigned)(m * g.lda + 4 * kl) * 4u;
    const unsigned b_lane = BWD ? (unsigned)(m * g.ldb + 4 * kl) * 4u : (unsigned)(4 * kl * g.ldb + 2 * m) * 4u;
    const unsigned b_row = (unsigned)g.ldb * 4u;              // forward: next k; backward: (x 32) the second accumulator's rows
    const unsigned b_chunk = BWD ? (unsigned)BK * 4u : (unsigned)BK * (unsigned)g.ldb * 4u;
    auto load = [&](int c, Chunk<BWD> &r) {
#if defined(__HIP_DEVICE_COMPILE__)
        const unsigned sa = (unsigned)c * (BK * 4), sb = (unsigned)c * b_chunk;
        r.a = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, a_lane, sa, 0));
        if constexpr (BWD) {
            r.b4[0] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, b_lane, sb, 0));
            r.b4[1] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, b_lane, sb + 32u * b_row, 0));
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                r.b2[j] = __builtin_bit_cast(f32x2_t, __builtin_amdgcn_raw_buffer_load_b64(rsrc_b, b_lane, sb + (unsigned)j * b_row, 0));
        }
#endif
    };
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    auto compute = [&](const Chunk<BWD> &r) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float b0 = BWD ? r.b4[0][j] : r.b2[j][0], b1 = BWD ? r.b4[1][j] : r.b2[j][1];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(r.a[j], b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(r.a[j], b1, acc1, 0, 0, 0);
        }
    };
    f32x4_t actv = {0.f, 0.f, 0.f, 0.f};
    if constexpr (BWD) actv = *reinterpret_cast<const f32x4_t *>(g.act + (size_t)(m0 + (tid >> 4)) * g.ldact + n0 + (tid & 15) * 4);
    Chunk<BWD> ring[P];
#pragma unroll
    for (int s = 0; s < P; ++s)
        if (s < nkw) load(s, ring[s]);
    int c = 0;
    for (; c + 2 * P <= nkw; c += P) {
#pragma unroll
        for (int s = 0; s < P; ++s) {
            compute(ring[s]);
            load(c + s + P, ring[s]);
        }
    }
#pragma unroll
    for (int round = 0; round < 2; ++round) {
#pragma unroll
        for (int s = 0; s < P; ++s) {
            if (c + s < nkw) compute(ring[s]);
            if (c + s + P < nkw) load(c + s + P, ring[s]);
        }
        c += P;
    }
    // ---- epilogue: the eight K slices meet in LDS, fixed order
    const int fm = lane & 31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * kl;
        if constexpr (BWD) {
            lds.T[q][row][fm] = acc0[r];
            lds.T[q][row][32 + fm] = acc1[r];
        } else {
            *reinterpret_cast<f32x2_t *>(&lds.T[q][row][2 * fm]) = f32x2_t{acc0[r], acc1[r]};
        }
    }
    __syncthreads();
    {
        const int row = tid >> 4, c4 = (tid & 15) * 4;
        f32x4_t s = *reinterpret_cast<const f32x4_t *>(&lds.T[0][row][c4]);
#pragma unroll
        for (int p = 1; p < NW; ++p) {
            const f32x4_t sp = *reinterpret_cast<const f32x4_t *>(&lds.T[p][row][c4]);
#pragma unroll
            for (int j = 0; j < 4; ++j) s[j] += sp[j];
        }
        f32x4_t v;
        if constexpr (BWD) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = s[j] * (1.f - actv[j] * actv[j]);
        } else {
            const f32x4_t b = *reinterpret_cast<const f32x4_t *>(g.bias + n0 + c4);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = tanhf(s[j] + b[j]);
        }
        *reinterpret_cast<f32x4_t *>(g.out + (size_t)(m0 + row) * g.ldo + n0 + c4) = v;
    }
}

template <bool BWD>
int launch(const Args &g, int P, hipStream_t st)
{
    const int tiles = (g.M / BM) * (g.N / BN);
    switch (P) {
    case 4: hipLaunchKernelGGL((dense_reg_kernel<4, BWD>), dim3(tiles), dim3(512), 0, st, g); break;
    case 6: hipLaunchKernelGGL((dense_reg_kernel<6, BWD>), dim3(tiles), dim3(512), 0, st, g); break;
    case 8: hipLaunchKernelGGL((dense_reg_kernel<8, BWD>), dim3(tiles), dim3(512), 0, st, g); break;
    case 12: hipLaunchKernelGGL((dense_reg_kernel<12, BWD>), dim3(tiles), dim3(512), 0, st, g); break;
    default: return -2;
    }
    return (int)hipGetLastError();
}

}  // namespace

extern "C" {

/* forward: out = tanh(A[M][K] W[K][N] + bias); 0, negative (bad arguments) or a hipError_t. P = chunks in flight (4, 6, 8, 12). */
int bnn_dense_reg_forward_f32(const float *A, const float *W, const float *bias, float *out, int M, int N, int K, int lda, int ldw,
                              int ldo, int P, void *stream)
{
    if (!A || !W || !bias || !out || M % BM || N % BN || K % 8 || K < 64 || lda % 4 || ldw % 2 || ldo % 4) return -1;
    Args g{A, W, bias, out, nullptr, M, N, K, lda, ldw, ldo, 0};
    return launch<false>(g, P, static_cast<hipStream_t>(stream));
}

/* backward: out = (delta[M][K] W[N][K]^T) * (1 - act^2) */
int bnn_dense_reg_backward_f32(const float *delta, const float *W, const float *act, float *out, int M, int N, int K, int ldd, int ldw,
                               int lda, int ldo, int P, void *stream)
{
    if (!delta || !W || !act || !out || M % BM || N % BN || K % 8 || K < 64 || ldd % 4 || ldw % 4 || ldo % 4 || lda % 4) return -1;
    Args g{delta, W, nullptr, out, act, M, N, K, ldd, ldw, ldo, lda};
    return launch<true>(g, P, static_cast<hipStream_t>(stream));
}

}  // extern "C"
