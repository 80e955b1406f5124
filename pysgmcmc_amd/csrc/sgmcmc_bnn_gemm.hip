// sgmcmc_bnn_gemm.hip -- a hidden layer of the BNN (pysgmcmc/models/bayesian_neural_network.py:30-52) as ONE launch per direction:
// fp32 matrix-core products with the layer's elementwise work as their epilogue.
//
// forward (sgmcmc_bnn_dense_tanh_f32): bias + tanh, and for the last hidden layer the single output unit's dot product
//   out[m][n] = tanh( sum_k h[m][k] W[k][n] + b[n] )            h [M = batch][K], W [K][N] row-major (the arena's layout)
//   dot_parts[t][m] = sum_{n in column tile t} out[m][n] w_next[n]   (optional; the Dense(1) layer of :53-56, added up by the loss head)
// backward (sgmcmc_bnn_dense_tanh_backward_f32; what tf.gradients builds for the same lines): tanh' and the bias gradient's column sums
//   out[m][n] = ( sum_k delta[m][k] W[n][k] ) (1 - act[m][n]^2)  delta [M][K], W [N][K]: the same W, read along its rows
//   colsum_parts[t][n] = sum of out[m][n] over row tile t        (optional; row tiles added up by the next launch, see the kernel)
//
// Replaces library GEMM + sgmcmc_bias_tanh_f32 (+ sgmcmc_bias_tanh_rowdot_f32 for the last hidden layer). In isolation it
// only ties that pair on the 2048 x 2048 layer (19.5-20.2 vs 21.1-21.5 us; it was an experiment that missed its gate, see
// profiles/r04_fwd_epilogue_probe.txt for everything tried on the way), but in the sampler's
// step two launches per layer become one and the 10 M-parameter chain goes from 196.2 to 186.0 us per step with its three
// hidden layers on it.
//
// Decomposition for M = 256: the output is 512 MFMA tiles of 32 x 32 for 1024 SIMDs, so K must be split; a workgroup (ONE per
// CU, 8 waves = 2 per SIMD) owns a 32 x 64 output tile = 2 MFMA tiles x 4 quarters of every 64-deep K chunk and adds the
// quarters through LDS in a fixed order before the epilogue -- the sums are complete inside the workgroup, which is what lets
// the activation ride in the launch (and keeps the result bit-reproducible).
// Operands go from global memory DIRECTLY into LDS (buffer_load_dwordx4 ... lds) into a ring of NS stages (A 32 x 64, B 64 x 64
// floats = 24 KB), NS - 2 chunks in flight across bare s_barriers with counted vmcnt waits; the fragments of chunk c + 1 are
// read from LDS while the MFMAs of chunk c issue.
//   * fp32 MFMA and vector-ALU instructions share the SIMD's lanes: every VALU instruction in the K loop is MFMA time. The
//     loads are therefore BUFFER loads -- one constant 32-bit per-lane offset + a scalar chunk offset + a scalar descriptor: no
//     per-lane address arithmetic at all (64-bit global addresses cost 3.5 us of 23.9) -- and the loop is unrolled by the ring
//     depth, so LDS addresses are a per-lane register + an immediate and M0 a constant.
//   * A (k contiguous in memory): LDS image [m][64 k], 16-byte quads XOR-swizzled with m & 15 -- applied to the per-lane GLOBAL
//     offset, the LDS side of a direct load is lane-linear -- so that ds_read_b128 of 4 k values per lane is conflict-free; a
//     lane's 8 k values feed 8 MFMAs (the k order inside a chunk is permuted, a sum over k does not care).
//   * B (n contiguous): LDS image [k][64 n], one ds_read_b32 per MFMA, 32 consecutive lanes = 32 consecutive banks.
//   * two accumulators per wave (alternating MFMAs, added in the epilogue): consecutive MFMAs are independent.
// Workgroup -> tile map is XCD-aware (workgroup b runs on XCD b % 8): every XCD owns a contiguous range of column tiles, so
// each slice of W is pulled into exactly one XCD's L2.
// More tiles than compute units (round 5): the grid simply runs in rounds of workgroups; when the last round would be less than
// half full, its columns are cut into 32 x 32 HALF tiles (template parameter BN = 32: 8 waves = 1 MFMA tile x 8 K parts, a second
// launch on the same stream, see plan_columns()) -- 256 x 4864 outputs = 608 tiles run as 2 rounds + 192 half tiles on 256 CUs.
// Correct (tests/test_bnn_dense_gpu.py) but at that shape slower in the step than the library's stream-K product + activation
// launch (profiles/r05_dense_rounds.txt), so BNNCost's plan fuses a layer only while its launch is ONE round.
//
// fp32 MFMA is an exact fmaf chain (MI355X_MICROARCH.md): the product differs from a library GEMM in summation order only.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdint>
#include <type_traits>

#include "sgmcmc_hip.h"

#pragma clang fp contract(off)

#include "sgmcmc_host.hpp"

using namespace sgmcmc_host;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int BM = 32, BK = 64;
// BN (template parameter of the kernel): 64 = the tile the pipeline was built for -- 8 waves = 2 MFMA tiles x 4 K quarters; 32 = the
// HALF tile of a launch's last, partly filled round of workgroups -- 8 waves = 1 MFMA tile x 8 K eighths (see the entry points)
constexpr int TSQ_SLICES = 16;                              // as sgmcmc_kernels.hip: slices of the sum(theta^2) partials

// s_waitcnt immediate on gfx9: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14; expcnt / lgkmcnt = no wait
constexpr int vmcnt_imm(int n) { return (n & 15) | (7 << 4) | (15 << 8) | ((n >> 4) << 14); }

template <int N>
__device__ __forceinline__ void wait_vm()
{
    __builtin_amdgcn_s_waitcnt(vmcnt_imm(N));
}

// at most `chunks` chunks (LPW direct loads per wave each) may still be in flight
template <int MAXC, int LPW>
__device__ __forceinline__ void wait_chunks_in_flight(int chunks)
{
    if constexpr (MAXC == 0) {
        wait_vm<0>();
    } else {
        if (chunks >= MAXC) wait_vm<LPW * MAXC>();
        else wait_chunks_in_flight<MAXC - 1, LPW>(chunks);
    }
}

struct FwdArgs {
    const float *h, *W, *bias;  // BWD: h = delta of the layer above [M][K], W [N][K] (the layer's weights, read along their rows)
    float *out;
    const float *w_next;        // nullable
    float *dot_parts;           // [column tiles][M]: row part_base + (n0 - n_base) / BN of this launch's tile at column n0
    int n_base, n_cols, part_base;  // this launch covers columns [n_base, n_base + n_cols) of the N outputs
    const double *stats_ws;     // nullable: statistics workspace of the previous step kernel ...
    double *tsq_parts;          // ... whose sum(theta^2) records workgroups 0 .. 15 add up into 16 slices
    int M, N, K, ldh, ldw, ldo;
    // BWD only
    const float *act;           // [M][N] the layer's tanh outputs (pitch lda)
    float *colsum_parts;        // nullable: [M / 32][N] column sums of out over each row tile
    int lda;
    // BWD side job: the column sums an EARLIER launch left in parts are added up (row-tile order) into fin_colsum
    const float *fin_parts;     // nullable: [fin_rows][fin_n]
    const float *fin_bias;      // nullable with fin_beta == 0
    float *fin_colsum;          // [fin_n] = sum_r fin_parts[r][:] (+ fin_beta * fin_bias)
    float fin_beta;
    int fin_rows, fin_n;
};

template <int NS, int BN>
struct __attribute__((aligned(16))) FwdLds {
    static constexpr int KQ = 256 / BN, TP = BN + 4;        // K parts of a chunk = partial tiles the epilogue adds; their pitch
    union {
        struct {
            float A[NS][BM][BK];
            float B[NS][BK][BN];
        } ring;
        float T[KQ][BM][TP];
        double red[8];
        struct {
            float T_[KQ][BM][TP];
            float cs[8][BN];    // BWD: column sums of the waves' rows
        } epi;
    };
};

__device__ __forceinline__ float tanh_f32(float x) { return tanhf(x); }      // the activation launch's own tanh

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

// sum over the 16 lanes of a DPP row; the total arrives in the row's lane 15 (fixed order). LANES = 8: over each half of the
// row, totals in lanes 7 and 15
template <int LANES = 16>
__device__ __forceinline__ float row16_sum_lane15(float v)
{
    v += dpp_mov<0x111>(v);     // row_shr:1
    v += dpp_mov<0x112>(v);     // row_shr:2
    v += dpp_mov<0x114>(v);     // row_shr:4
    if constexpr (LANES == 16) v += dpp_mov<0x118>(v);     // row_shr:8
    return v;
}

// BWD = false: the forward layer above. BWD = true: the layer's backward step,
//   out[m][n] = ( sum_k delta[m][k] W[n][k] ) * (1 - act[m][n]^2),  colsum[n] = sum_m out[m][n] (+ beta bias[n])
// -- the same pipeline with B read along the rows of W (contraction-contiguous like A: same swizzled LDS image, two
// ds_read_b128 per chunk instead of eight ds_read_b32).
template <int NS, bool BWD, int BN>
__global__ void __launch_bounds__(512, 2) bnn_dense_tanh_kernel(const FwdArgs g)
{
    static_assert(NS >= 4 && NS % 2 == 0, "ring: one chunk being read, one landing, one free; unrolled by NS with two fragment sets");
    static_assert(BN == 64 || BN == 32, "a workgroup's 8 waves are 2 MFMA tiles x 4 K parts or 1 x 8");
    constexpr int D = NS - 1;                               // chunk kc + D is requested in iteration kc
    constexpr int KQ = 256 / BN, KW = BK / KQ;              // K parts of a chunk; k values per wave and chunk (16 or 8)
    constexpr int NM = KW / 2;                              // ... = MFMAs per wave and chunk x 2 (lane halves kl = 0 / 1 hold k pairs)
    constexpr int LPW = BN == 64 ? 3 : 2;                   // direct loads per wave and chunk
    __shared__ FwdLds<NS, BN> lds;                          // ONE shared object (a second one de-pipelines the direct loads)
    static_assert(sizeof(lds.ring) >= sizeof(lds.T), "the accumulator tiles reuse the ring");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = BN == 64 ? (wave & 1) : 0, kq = BN == 64 ? (wave >> 1) : wave;     // this wave's MFMA tile (columns 32 nt ...) and K part of every chunk
    // ---- workgroup -> tile, XCD-aware: XCD x owns tiles [x T/8, (x+1) T/8), consecutive tiles share the column tile
    const int tiles_m = g.M / BM, tiles = tiles_m * (g.n_cols / BN);
    int t = blockIdx.x;
    if (tiles % 8 == 0) t = (t & 7) * (tiles >> 3) + (t >> 3);
    const int n0 = g.n_base + (t / tiles_m) * BN, m0 = (t % tiles_m) * BM;
    const int nk = (g.K + BK - 1) / BK;                     // K % 16 == 0: the last chunk may hold 16, 32 or 48 valid k only
    // ---- direct loads: wave w requests 4 rows of A (256 B each) and, BN = 64, 2 x 4 rows of B per chunk (BN = 32: ONE 1 KiB
    // piece of B: forward 8 rows k of 128 B, backward 4 rows n of 256 B)
    const int ar = 4 * wave + (lane >> 4);                  // A row of this lane's 16 bytes
    const int aq = (lane & 15) ^ (ar & 15);                 // logical quad stored at physical slot lane & 15
    // B rows of this lane: BN = 64 br, br + 4 (two pieces); BN = 32 forward 8 w + lane / 8 (quad lane & 7), backward = A's pattern
    const int br = BN == 64 ? 8 * wave + (lane >> 4) : (BWD ? ar : 8 * wave + (lane >> 3));
    [[maybe_unused]] const unsigned a_lane = (unsigned)(ar * g.ldh + 4 * aq) * 4u;
    // B forward: rows k of W (n contiguous), lane & 15 = quad of the 64 tile columns. B backward: tile column n = row n0 + br of W
    // (k contiguous): quads swizzled like A's (rows br and br + 4 differ in bit 2 of the swizzle: two lane offsets)
    const int bq0 = (lane & 15) ^ (br & 15), bq1 = (lane & 15) ^ ((br + 4) & 15);
    [[maybe_unused]] const unsigned b_lane = BWD ? (unsigned)(br * g.ldw + 4 * bq0) * 4u
                                                 : (unsigned)(br * g.ldw + 4 * (BN == 64 ? (lane & 15) : (lane & 7))) * 4u;
    [[maybe_unused]] const unsigned b_lane1 = (unsigned)((br + 4) * g.ldw + 4 * bq1) * 4u;
    [[maybe_unused]] constexpr int BR0 = BN == 64 ? 8 : (BWD ? 4 : 8);       // first B row of wave w's piece = BR0 w
    [[maybe_unused]] const unsigned b_chunk = (unsigned)BK * (unsigned)g.ldw * 4u, b_rows4 = 4u * (unsigned)g.ldw * 4u;
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(g.h + (size_t)m0 * g.ldh), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(BWD ? g.W + (size_t)n0 * g.ldw : g.W + n0), 0, 0x7fffffff, 0x00020000);
#endif
    // row r of the B image of stage st: forward [k][BN n], backward [n][BK k] (same bytes, BK * BN floats per stage)
    [[maybe_unused]] auto ldsB = [&](int st, int r) { return &lds.ring.B[st][0][0] + r * (BWD ? BK : BN); };
    auto issue = [&](int kc, int st) {                      // chunks that lie wholly below K: scalar offsets only
#if defined(__HIP_DEVICE_COMPILE__)
        const unsigned sa = (unsigned)kc * (BK * 4);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, &lds.ring.A[st][4 * wave][0], 16, a_lane, sa, 0, 0);
        if constexpr (BWD) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, ldsB(st, BR0 * wave), 16, b_lane, sa, 0, 0);
            if constexpr (BN == 64) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, ldsB(st, 8 * wave + 4), 16, b_lane1, sa, 0, 0);
        } else {
            const unsigned sb = (unsigned)kc * b_chunk;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, ldsB(st, 8 * wave), 16, b_lane, sb, 0, 0);
            if constexpr (BN == 64) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, ldsB(st, 8 * wave + 4), 16, b_lane, sb + b_rows4, 0, 0);
        }
#endif
    };
    auto issue_clamped = [&](int kc, int st) {              // prologue / last chunks: beyond K any valid address will do (never used)
#if defined(__HIP_DEVICE_COMPILE__)
        int ka = kc * BK + 4 * aq;
        if (ka + 4 > g.K) ka = g.K - 4;
        const unsigned va = (unsigned)(ar * g.ldh + ka) * 4u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, &lds.ring.A[st][4 * wave][0], 16, va, 0, 0, 0);
        if constexpr (BWD) {
            int k0 = kc * BK + 4 * bq0, k1 = kc * BK + 4 * bq1;
            if (k0 + 4 > g.K) k0 = g.K - 4;
            if (k1 + 4 > g.K) k1 = g.K - 4;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, ldsB(st, BR0 * wave), 16, (unsigned)(br * g.ldw + k0) * 4u, 0, 0, 0);
            if constexpr (BN == 64)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, ldsB(st, 8 * wave + 4), 16, (unsigned)((br + 4) * g.ldw + k1) * 4u, 0, 0, 0);
        } else {
            int kb0 = kc * BK + br, kb1 = kc * BK + br + 4;
            if (kb0 >= g.K) kb0 = g.K - 1;
            if (kb1 >= g.K) kb1 = g.K - 1;
            const unsigned c4 = 4u * (unsigned)(BN == 64 ? (lane & 15) : (lane & 7)) * 4u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, ldsB(st, 8 * wave), 16, (unsigned)kb0 * (unsigned)g.ldw * 4u + c4, 0, 0, 0);
            if constexpr (BN == 64)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, ldsB(st, 8 * wave + 4), 16, (unsigned)kb1 * (unsigned)g.ldw * 4u + c4, 0, 0, 0);
        }
#endif
    };
    // ---- fragment addresses
    const int fm = lane & 31, kl = lane >> 5;
    const int sw = fm & 15;
    // a lane holds NM = 8 (4) consecutive k of its wave's K part: quads 4 kq + 2 kl, + 1 (BN = 32: the one quad 2 kq + kl)
    constexpr int QW = KW / 4;                              // quads per wave and chunk
    const int aoff0 = fm * BK + 4 * ((QW * kq + (QW / 2) * kl) ^ sw), aoff1 = fm * BK + 4 * ((QW * kq + (QW / 2) * kl + 1) ^ sw);
    const int boff = BWD ? 32 * nt * BK + aoff0 : (KW * kq + NM * kl) * BN + 32 * nt + fm;     // BWD: row 32 nt + fm, same quads as A
    const int boff1 = 32 * nt * BK + aoff1;
    struct Frag {
        f32x4_t a0, a1;
        float b[NM];
    };
    auto read_frags = [&](int st, Frag &f) {
        const float *A = &lds.ring.A[st][0][0];
        const float *B = &lds.ring.B[st][0][0];
        f.a0 = *reinterpret_cast<const f32x4_t *>(A + aoff0);
        if constexpr (NM == 8) f.a1 = *reinterpret_cast<const f32x4_t *>(A + aoff1);
        if constexpr (BWD) {
            const f32x4_t b0 = *reinterpret_cast<const f32x4_t *>(B + boff);
#pragma unroll
            for (int j = 0; j < 4; ++j) f.b[j] = b0[j];
            if constexpr (NM == 8) {
                const f32x4_t b1 = *reinterpret_cast<const f32x4_t *>(B + boff1);
#pragma unroll
                for (int j = 0; j < 4; ++j) f.b[4 + j] = b1[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < NM; ++j) f.b[j] = B[boff + j * BN];
        }
    };
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    auto mfmas = [&](const Frag &f, int j0, int j1) {
#pragma unroll
        for (int j = 0; j < NM; ++j)
            if (j >= j0 && j < j1) {
                const float a = j < 4 ? f.a0[j & 3] : f.a1[j & 3];
                if (!(j & 1)) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, f.b[j], acc0, 0, 0, 0);
                else acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, f.b[j], acc1, 0, 0, 0);
            }
    };
    // ---- prologue: (BWD) this lane's quad of the activations for the epilogue, then chunks 0 .. D - 1 requested ...
    f32x4_t actv = {0.f, 0.f, 0.f, 0.f};
    constexpr int QR = BN / 4;                              // quads (= epilogue lanes) per tile row; lanes >= 8 BN idle in the epilogue
    const int erow = tid / QR, ec4 = (tid % QR) * 4;
    const bool eactive = tid < BM * QR;
    if constexpr (BWD) {
        if (eactive) actv = *reinterpret_cast<const f32x4_t *>(g.act + (size_t)(m0 + erow) * g.lda + n0 + ec4);
    }
    // (BWD) side job of wave 0: the bias gradient of the layer ABOVE -- the per-row-tile column sums the previous launch left are
    // added up in row-tile order (deterministic; the launch boundary is what makes them visible). Loads first, the sums after
    // the chunk requests: older vector-memory operations only make the counted waits below stronger.
    constexpr int FIN_PRE = 8;                              // row tiles requested ahead (batch 256); any further ones afterwards
    float finv[BWD ? FIN_PRE : 1];
    const int fin_c = (int)blockIdx.x * BN + tid;
    const bool finisher = BWD && g.fin_parts != nullptr && tid < BN && fin_c < g.fin_n;
    if constexpr (BWD) {
        if (finisher) {
#pragma unroll
            for (int r = 0; r < FIN_PRE; ++r)
                if (r < g.fin_rows) finv[r] = g.fin_parts[(size_t)r * g.fin_n + fin_c];
        }
    }
#pragma unroll
    for (int c = 0; c < D; ++c)
        if (c < nk) issue_clamped(c, c);
    if constexpr (BWD) {
        if (finisher) {
            float tot = finv[0];
#pragma unroll
            for (int r = 1; r < FIN_PRE; ++r)
                if (r < g.fin_rows) tot += finv[r];
            for (int r = FIN_PRE; r < g.fin_rows; ++r) tot += g.fin_parts[(size_t)r * g.fin_n + fin_c];
            g.fin_colsum[fin_c] = (g.fin_beta != 0.f) ? tot + g.fin_beta * g.fin_bias[fin_c] : tot;
            for (int c = fin_c + (int)gridDim.x * BN; c < g.fin_n; c += (int)gridDim.x * BN) {      // more columns than 64 per workgroup
                float t2 = g.fin_parts[c];
                for (int r = 1; r < g.fin_rows; ++r) t2 += g.fin_parts[(size_t)r * g.fin_n + c];
                g.fin_colsum[c] = (g.fin_beta != 0.f) ? t2 + g.fin_beta * g.fin_bias[c] : t2;
            }
        }
    }
    // ... and while they fly, the side job of the last hidden layer's launch: workgroups 0 .. 15 add up one contiguous slice
    // each of the sum(theta^2) records the previous step kernel left in its statistics workspace (the loss head adds the
    // slices in order: same arithmetic as sgmcmc_bias_tanh_rowdot_*'s side job)
    double tsq = 0.0;
    const unsigned n_slices = gridDim.x < (unsigned)TSQ_SLICES ? gridDim.x : (unsigned)TSQ_SLICES;
    const bool slicer = !BWD && g.stats_ws != nullptr && blockIdx.x < n_slices;
    if (slicer) {
        const unsigned nparts = (unsigned)reinterpret_cast<const unsigned long long *>(g.stats_ws)[0];
        const double *__restrict__ p = g.stats_ws + 4;
        const unsigned len = (nparts + n_slices - 1) / n_slices;
        const unsigned lo = blockIdx.x * len, hi = (lo + len < nparts) ? lo + len : nparts;
        for (unsigned i = lo + tid; i < hi; i += 512) tsq += p[4 * (size_t)i];      // statistic 0 of record i
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) tsq += __shfl_down(tsq, off, 64);
    }
    wait_chunks_in_flight<D - 1, LPW>(nk - 1);
    __builtin_amdgcn_s_barrier();
    Frag fa, fb;
    read_frags(0, fa);
    // One iteration: chunk kc + 1 is waited for and read from LDS (into the other fragment set) while the MFMAs of chunk kc
    // issue. The steady state is unrolled by NS: ring stages are compile-time constants and the two fragment sets swap roles
    // without register copies.
    auto steady = [&](int kc, const Frag &cur, Frag &nxt, auto stage) {
        constexpr int I = decltype(stage)::value;           // kc % NS
        wait_vm<LPW * (D - 2)>();                           // chunk kc + 1 landed: chunks kc + 2 .. kc + D - 1 may be in flight
        __builtin_amdgcn_s_barrier();                       // ... for every wave; and the stage of chunk kc - 1 is free
        // order inside an iteration (10 M-parameter chain, us per step): the fragment reads right
        // after the second MFMA and the loads after the fourth 186.0; loads + reads bunched after the second 187.0; one load per
        // MFMA gap 186.0; s_setprio 1 for the later-dispatched wave of each SIMD 187.7
        mfmas(cur, 0, NM / 4);
        __builtin_amdgcn_sched_barrier(0);
        read_frags((I + 1) % NS, nxt);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(cur, NM / 4, NM / 2);
        __builtin_amdgcn_sched_barrier(0);
        issue(kc + D, (I + D) % NS);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(cur, NM / 2, NM);
    };
    int kc = 0;
    for (; kc + D + NS < nk; kc += NS) {                    // every chunk requested here (up to kc + NS - 1 + D) lies wholly below K
        steady(kc, fa, fb, std::integral_constant<int, 0>());
        steady(kc + 1, fb, fa, std::integral_constant<int, 1>());
        steady(kc + 2, fa, fb, std::integral_constant<int, 2>());
        steady(kc + 3, fb, fa, std::integral_constant<int, 3>());
    }
    int st_read = 1, st_issue = D % NS;                     // kc % NS == 0 here: stage of chunk kc + 1, stage of chunk kc + D
    for (; kc + 1 < nk; ++kc) {                             // the last chunks: counted waits, the (possibly short) last chunk requested
        wait_chunks_in_flight<D - 2, LPW>(nk - 2 - kc);
        __builtin_amdgcn_s_barrier();
        mfmas(fa, 0, NM / 4);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(st_read, fb);
        if (kc + D < nk) issue_clamped(kc + D, st_issue);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(fa, NM / 4, NM);
        fa = fb;
        st_read = st_read + 1 == NS ? 0 : st_read + 1;
        st_issue = st_issue + 1 == NS ? 0 : st_issue + 1;
    }
    // last chunk: only the K parts that lie below K exist (K % 16 == 0: a half tile's eighths exist in pairs)
    if ((nk - 1) * BK + KW * kq < g.K) mfmas(fa, 0, NM);
    // ---- epilogue: the KQ partial tiles meet in LDS (fixed order), bias + tanh on row-major quads, 16-byte stores
    __syncthreads();                                        // every fragment read is done: the ring is free
#pragma unroll
    for (int r = 0; r < 16; ++r) lds.T[kq][(r & 3) + 8 * (r >> 2) + 4 * kl][32 * nt + fm] = acc0[r] + acc1[r];
    __syncthreads();
    {
        const int row = erow, c4 = ec4;                     // lanes tid < 8 BN own one quad of the tile each
        f32x4_t s = {0.f, 0.f, 0.f, 0.f};
        if (eactive) {
            s = *reinterpret_cast<const f32x4_t *>(&lds.T[0][row][c4]);
#pragma unroll
            for (int p = 1; p < KQ; ++p) {
                const f32x4_t sp = *reinterpret_cast<const f32x4_t *>(&lds.T[p][row][c4]);
#pragma unroll
                for (int j = 0; j < 4; ++j) s[j] += sp[j];
            }
        }
        if constexpr (BWD) {
            // tanh' of the layer below as the epilogue (sgmcmc_tanh_backward_colsum_*'s arithmetic) ...
            f32x4_t v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = s[j] * (1.f - actv[j] * actv[j]);
            if (eactive) *reinterpret_cast<f32x4_t *>(g.out + (size_t)(m0 + row) * g.ldo + n0 + c4) = v;
            // ... and the bias gradient: column sums over the tile's 32 rows (the rows of a wave by shuffles, the waves in order).
            // The row tiles of a column are NOT added here: that takes a second pass over what other workgroups wrote, and in
            // one launch it costs more than it saves (arrival counter + agent-scope fences: +12 us; relaxed write-through
            // atomics: +2.5 us, the step got slower -- profiles/r04_bwd_epilogue_probe.txt). The partial rows are left
            // for the NEXT launch to add up on the side (fin_* above), or for sgmcmc_colsum_finish_f32.
            if (g.colsum_parts == nullptr) return;          // (the first layer's bias gradient comes from the [x | 1]^T delta product)
#pragma unroll
            for (int off = QR; off < 64; off <<= 1)
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += __shfl_xor(v[j], off, 64);
            if (eactive && lane < QR) *reinterpret_cast<f32x4_t *>(&lds.epi.cs[wave][c4]) = v;
            __syncthreads();
            if (tid < BN) {
                constexpr int AW = BM * QR / 64;            // waves that hold tile rows
                float tot = lds.epi.cs[0][tid];
#pragma unroll
                for (int w = 1; w < AW; ++w) tot += lds.epi.cs[w][tid];
                g.colsum_parts[(size_t)(m0 / BM) * g.N + n0 + tid] = tot;
            }
            return;
        }
        if (eactive) {
            const f32x4_t b = *reinterpret_cast<const f32x4_t *>(g.bias + n0 + c4);
            f32x4_t v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = tanh_f32(s[j] + b[j]);
            *reinterpret_cast<f32x4_t *>(g.out + (size_t)(m0 + row) * g.ldo + n0 + c4) = v;
            if (g.w_next != nullptr) {
                const f32x4_t w = *reinterpret_cast<const f32x4_t *>(g.w_next + n0 + c4);
                float d = ((v[0] * w[0] + v[1] * w[1]) + v[2] * w[2]) + v[3] * w[3];
                d = row16_sum_lane15<QR>(d);                // the QR lanes of a tile row are (half of) one DPP row
                if ((lane & (QR - 1)) == QR - 1) g.dot_parts[(size_t)(g.part_base + (n0 - g.n_base) / BN) * g.M + m0 + row] = d;
            }
        }
    }
    if (slicer) {                                           // uniform per workgroup
        __syncthreads();                                    // T has been consumed: red (same union) may be written
        if (lane == 0) lds.red[wave] = tsq;
        __syncthreads();
        if (tid == 0)
            g.tsq_parts[blockIdx.x] = ((((((lds.red[0] + lds.red[1]) + lds.red[2]) + lds.red[3]) + lds.red[4]) + lds.red[5]) + lds.red[6]) + lds.red[7];
    }
}

// colsum[c] = sum_r parts[r][c] (+ beta bias[c]) in row order: what a following backward launch does on the side, as a launch
__global__ void __launch_bounds__(256) colsum_finish_kernel(const float *__restrict__ parts, int rows, int n, const float *__restrict__ bias,
                                                            float beta, float *__restrict__ colsum)
{
    const int c = (int)(blockIdx.x * 256u + threadIdx.x);
    if (c >= n) return;
    float tot = parts[c];
    for (int r = 1; r < rows; ++r) tot += parts[(size_t)r * n + c];
    colsum[c] = (beta != 0.f) ? tot + beta * bias[c] : tot;
}

// Which columns of the N outputs run as 32 x 64 tiles and which as 32 x 32 half tiles. A launch of T tiles on C compute units
// takes ceil(T / C) rounds of workgroups (one per CU: the ring fills LDS); when the last round is less than half full, the columns
// it would cover are cut into HALF tiles instead -- twice as many workgroups of half the work, all running at once -- so that
// e.g. 256 x 4864 (608 tiles on 256 CUs) takes 2.5 rounds instead of 3. Full tiles come first (columns [0, n_full)).
struct ColumnPlan {
    int n_full;                 // columns on full tiles (a multiple of 64)
    int parts;                  // column tiles in all = rows of dot_parts: n_full / 64 + (N - n_full) / 32
};

// Compute units of the current device, queried once per device and kept (the plan a caller sized dot_parts with and the plan a
// launch uses must be the same split: both read this). 0: no device / the query failed.
int current_device_cus()
{
    constexpr int MAX_DEV = 64;
    static std::atomic<int> cached[MAX_DEV];                // zero-initialised: 0 = not asked yet
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return 0;
    int cus = cached[dev].load(std::memory_order_relaxed);
    if (cus > 0) return cus;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 0;
    cached[dev].store(cus, std::memory_order_relaxed);
    return cus;
}

ColumnPlan plan_columns(int M, int N, int cus)
{
    const int tiles_m = M / BM, col_tiles = N / 64, tiles = tiles_m * col_tiles;
    ColumnPlan p{N, col_tiles};
    if (tiles <= cus || tiles % cus == 0) return p;
    const int full_cols = (tiles / cus) * cus / tiles_m;     // column tiles that fill whole rounds
    const int left = col_tiles - full_cols;
    if (full_cols >= 1 && 2 * left * tiles_m <= cus) {       // the half tiles run as ONE round
        p.n_full = full_cols * 64;
        p.parts = full_cols + 2 * left;
    }
    return p;
}

template <bool BWD>
int launch_dense(FwdArgs g, const ColumnPlan &plan, hipStream_t stream, const char *what)
{
    g.n_base = 0;
    g.n_cols = plan.n_full;
    g.part_base = 0;
    hipLaunchKernelGGL((bnn_dense_tanh_kernel<4, BWD, 64>), dim3((g.M / BM) * (plan.n_full / 64)), dim3(512), 0, stream, g);
    if (plan.n_full < g.N) {
        // the rest of the columns as half tiles, behind the full ones on the stream; the side jobs (column-sum finish, sum(theta^2)
        // slices) belong to the first launch
        g.n_base = plan.n_full;
        g.n_cols = g.N - plan.n_full;
        g.part_base = plan.n_full / 64;
        g.fin_parts = nullptr;
        g.stats_ws = nullptr;
        g.tsq_parts = nullptr;
        hipLaunchKernelGGL((bnn_dense_tanh_kernel<4, BWD, 32>), dim3((g.M / BM) * (g.n_cols / 32)), dim3(512), 0, stream, g);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, what);
}

}  // namespace

extern "C" {

/* see include/sgmcmc_hip.h */
int sgmcmc_bnn_dense_tanh_dot_parts(int M, int N)
{
    if (M <= 0 || N <= 0 || M % BM || N % 64) return 0;
    const int cus = current_device_cus();                   // (without a device this is host arithmetic for MI355X's 256 CUs;
    return plan_columns(M, N, cus > 0 ? cus : 256).parts;   //  a launch cannot follow it then)
}

/* see include/sgmcmc_hip.h */
int sgmcmc_bnn_dense_tanh_f32(const float *h, const float *W, const float *bias, float *out, int M, int N, int K, int ldh,
                              int ldw, int ldo, const float *w_next, float *dot_parts, const void *stats_ws, double *tsq_parts,
                              sgmcmc_stream_t stream)
{
    if (!h || !W || !bias || !out || ((w_next != nullptr) != (dot_parts != nullptr)) || ((stats_ws != nullptr) != (tsq_parts != nullptr)))
        return fail(SGMCMC_EINVAL, "bnn_dense_tanh: NULL argument (w_next / dot_parts and stats_ws / tsq_parts go together)");
    if (M <= 0 || N <= 0 || K < 64 || M % BM || N % 64 || K % 16 || ldh < K || ldw < N || ldo < N || ldh % 4 || ldw % 4 || ldo % 4 ||
        ((reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(bias) |
          reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(w_next)) & 15u))
        return fail(SGMCMC_EINVAL, "bnn_dense_tanh: needs M %% 32 == 0, N %% 64 == 0, K %% 16 == 0, K >= 64, 16-byte aligned rows");
    if ((double)K * ldw * 4.0 >= 2147483648.0 || (double)M * ldh * 4.0 >= 2147483648.0)
        return fail(SGMCMC_EINVAL, "bnn_dense_tanh: an operand spans more than 2 GiB (32-bit buffer offsets)");
    const int cus = current_device_cus();
    if (cus <= 0) return fail(SGMCMC_ENODEV, "bnn_dense_tanh: no HIP device (compute-unit count unavailable)");
    const ColumnPlan plan = plan_columns(M, N, cus);
    if (stats_ws != nullptr && (M / BM) * (plan.n_full / 64) < TSQ_SLICES)
        return fail(SGMCMC_EINVAL, "bnn_dense_tanh: the sum(theta^2) side job needs at least 16 output tiles (the loss head adds 16 slices)");
    FwdArgs g{};
    g.h = h; g.W = W; g.bias = bias; g.out = out; g.w_next = w_next; g.dot_parts = dot_parts;
    g.stats_ws = static_cast<const double *>(stats_ws); g.tsq_parts = tsq_parts;
    g.M = M; g.N = N; g.K = K; g.ldh = ldh; g.ldw = ldw; g.ldo = ldo;
    return launch_dense<false>(g, plan, static_cast<hipStream_t>(stream), "launch bnn_dense_tanh");
}

/* see include/sgmcmc_hip.h */
int sgmcmc_bnn_dense_tanh_backward_f32(const float *delta, const float *W, const float *act, float *out, float *colsum_parts,
                                       int M, int N, int K, int ldd, int ldw, int lda, int ldo, const float *fin_parts,
                                       int fin_rows, int fin_n, const float *fin_bias, float fin_beta, float *fin_colsum,
                                       sgmcmc_stream_t stream)
{
    if (!delta || !W || !act || !out)
        return fail(SGMCMC_EINVAL, "bnn_dense_tanh_backward: NULL argument");
    if (M <= 0 || N <= 0 || K < 64 || M % BM || N % 64 || K % 16 || ldd < K || ldw < K || lda < N || ldo < N || ldd % 4 || ldw % 4 ||
        lda % 4 || ldo % 4 ||
        ((reinterpret_cast<uintptr_t>(delta) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(act) |
          reinterpret_cast<uintptr_t>(out)) & 15u))
        return fail(SGMCMC_EINVAL, "bnn_dense_tanh_backward: needs M %% 32 == 0, N %% 64 == 0, K %% 16 == 0, K >= 64, 16-byte aligned rows");
    if ((double)N * ldw * 4.0 >= 2147483648.0 || (double)M * ldd * 4.0 >= 2147483648.0)
        return fail(SGMCMC_EINVAL, "bnn_dense_tanh_backward: an operand spans more than 2 GiB (32-bit buffer offsets)");
    if (fin_parts != nullptr &&
        (!fin_colsum || fin_rows <= 0 || fin_n <= 0 || (fin_beta != 0.f && !fin_bias) || fin_parts == colsum_parts))
        return fail(SGMCMC_EINVAL, "bnn_dense_tanh_backward: the side job needs fin_colsum, fin_rows > 0, fin_n > 0, fin_bias with "
                                   "fin_beta != 0 and partial sums other than the ones this launch writes");
    FwdArgs g{};
    g.h = delta; g.W = W; g.out = out; g.M = M; g.N = N; g.K = K; g.ldh = ldd; g.ldw = ldw; g.ldo = ldo;
    g.act = act; g.colsum_parts = colsum_parts; g.lda = lda;
    g.fin_parts = fin_parts; g.fin_bias = fin_bias; g.fin_colsum = fin_colsum; g.fin_beta = fin_beta; g.fin_rows = fin_rows; g.fin_n = fin_n;
    const int cus = current_device_cus();
    if (cus <= 0) return fail(SGMCMC_ENODEV, "bnn_dense_tanh_backward: no HIP device (compute-unit count unavailable)");
    return launch_dense<true>(g, plan_columns(M, N, cus), static_cast<hipStream_t>(stream), "launch bnn_dense_tanh_backward");
}

/* see include/sgmcmc_hip.h */
int sgmcmc_colsum_finish_f32(const float *parts, int rows, int n, const float *bias, float beta, float *colsum, sgmcmc_stream_t stream)
{
    if (n <= 0) return 0;
    if (!parts || !colsum || rows <= 0 || (beta != 0.f && !bias)) return fail(SGMCMC_EINVAL, "colsum_finish: NULL argument or no rows");
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), parts,
                       rows, n, bias, beta, colsum);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "launch colsum_finish");
}

}  // extern "C"
