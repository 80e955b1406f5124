// bnn_gw_bf16x3.hpp -- round 6 gate experiment (NOT part of libsgmcmc_hip.so): the weight gradients of the BNN's dense layers, gW = h^T delta (what tf.gradients builds for
// pysgmcmc/models/bayesian_neural_network.py:30-56, reached from pysgmcmc/samplers/sghmc.py:121-122), at fp32 accuracy on the
// bf16 matrix pipe.
//
// Both operands of this product are ACTIVATIONS of the step ([batch][features], the batch is the contraction index), so the
// launches that produce them can write them a second time as three exact bf16 planes
//     x = x0 + x1 + x2,  x0 = top 16 bits of x, x1 = top 16 bits of (x - x0), x2 = x - x0 - x1   (8 + 8 + 8 significant bits, no rounding)
// and the product loop is direct-to-LDS loads + six v_mfma_f32_32x32x16_bf16 per 16 batch rows -- the partial products of order
// <= 2^-16 (a0 b0 | a0 b1, a1 b0 | a0 b2, a1 b1, a2 b0) -- with NO vector-ALU work (round 5 showed that splitting operands inside
// the product loop is what sinks this idea on gfx950: profiles/r05_bf16x3_gate.txt).
//
// Plane layout (chosen for this kernel; the producers' epilogues transpose through LDS anyway): plane p of X [M][N] is
//     P[p][m / 8][n][m % 8]  (bf16)     -- 16 bytes = the 8 batch rows one lane feeds to one MFMA for feature n
// so a wave's fragment read is one conflict-free ds_read_b128 per lane and a direct load of 64 lanes x 16 bytes is 1 KiB contiguous
// both in memory and in LDS.
//
// Decomposition: 128 x 128 output tile per workgroup of 4 waves (one per SIMD; 2 workgroups per CU), wave = 64 x 64 = 2 x 2 MFMA
// tiles; a chunk = 16 batch rows = ONE MFMA k-step: 24 KiB of planes per stage (2 operands x 3 planes x 2 x 128 x 16 B),
// ring of NS stages. Accumulation order over the batch is fixed: results are bit-reproducible.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace sgmcmc_gw {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int TILE = 128, KC = 16, STAGE_BYTES = 24 * 1024;
constexpr int MAX_BATCH = 4;

struct GwArgs {
    const void *A[MAX_BATCH];   // planes of the layer inputs  h   [M][nA]: rows of the gradient
    const void *B[MAX_BATCH];   // planes of the deltas            [M][nB]: columns of the gradient
    float *C[MAX_BATCH];        // gradient [nA][nB], pitch ldc (a slice of the gradient arena)
    int nA, nB, M, ldc;
    unsigned plane_a_bytes, plane_b_bytes;      // distance between planes (>= M * n * 2)
    int tiles_i, tiles_j;       // ceil(nA / 128), ceil(nB / 128)
    int batch;                  // products per launch (persistent kernel)
};

constexpr int vmcnt_imm(int n) { return (n & 15) | (7 << 4) | (15 << 8) | ((n >> 4) << 14); }
template <int N>
__device__ __forceinline__ void wait_vm()
{
    __builtin_amdgcn_s_waitcnt(vmcnt_imm(N));
}

// exact three-way split of a pair of consecutive batch rows (x = row m, y = row m + 1), packed low half = x
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

// NS ring stages; TWO_ACC: the five small partial products get accumulators of their own (added to a0 b0's at the end)
template <int NS, bool TWO_ACC, int PROBE = 0, bool STORE16 = false, bool NTS = false>       // NTS: nontemporal stores of the gradient; PROBE (timing experiments only): 1 no MFMA, 2 no loads in the loop, 4 no stores, 8 no fragment reads in the loop
__global__ void __launch_bounds__(256, 2) gw_bf16x3_kernel(const GwArgs g)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NS][STAGE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 1, wj = wave & 1;
    // ---- workgroup -> (product z, tile): XCD-aware when the tile grid is made of 8 x 8 blocks (workgroup b runs on XCD b % 8:
    // every XCD works on whole 8 x 8 blocks of tiles, 16 operand slices of 64 KiB x 3 planes per block in its L2)
    const int per = g.tiles_i * g.tiles_j;
    int id = blockIdx.x, z, ti, tj;
    {
        // workgroup b runs on XCD b % 8: XCD x gets the CONTIGUOUS range of tile numbers [start_x, start_x + count_x) ...
        const int T = (int)gridDim.x, q8 = T >> 3, r8 = T & 7, x = id & 7;
        id = x * q8 + (x < r8 ? x : r8) + (id >> 3);
        z = id / per;
        // ... and tile numbers walk the gradient in panels of 8 tile rows, rows fastest: 64 consecutive numbers = an 8 x 8 block of
        // tiles = 16 operand slices of 64 KiB x 3 planes in the XCD's L2
        const int r = id - z * per, panel = 8 * g.tiles_j, gidx = r / panel, rows = (g.tiles_i - 8 * gidx) < 8 ? (g.tiles_i - 8 * gidx) : 8;
        const int w = r - gidx * panel;
        ti = 8 * gidx + w % rows;
        tj = w / rows;
    }
    const int i0 = ti * TILE, j0 = tj * TILE;
    const int nk = g.M / KC;
    // ---- direct loads: a chunk is 24 pieces of 1 KiB (operand, plane, m8 of the chunk, half of the tile's 128 features); wave w
    // requests pieces 3 w .. 3 w + 2 of A and of B. Features beyond nA / nB: the buffer range check returns zeros.
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void *>(g.A[z]), 0, (int)(2u * g.plane_a_bytes + (unsigned)(g.M / 8) * (unsigned)g.nA * 16u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void *>(g.B[z]), 0, (int)(2u * g.plane_b_bytes + (unsigned)(g.M / 8) * (unsigned)g.nB * 16u), 0x00020000);
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
    f32x16 accm[2][2], accs[2][2];
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
    ACC[I][J] = STORE16 ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[J][PB]), __builtin_bit_cast(bf16x8, fa[I][PA]), \
                                                                  ACC[I][J], 0, 0, 0)                                                \
                        : __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[I][PA]), __builtin_bit_cast(bf16x8, fb[J][PB]), \
                                                                  ACC[I][J], 0, 0, 0)
#define GW_ALL(PA, PB, ACC) GW_MF(0, 0, PA, PB, ACC); GW_MF(0, 1, PA, PB, ACC); GW_MF(1, 0, PA, PB, ACC); GW_MF(1, 1, PA, PB, ACC)
        if constexpr (PROBE & 1) {
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y)
#pragma unroll
                    for (int p = 0; p < 3; ++p)
#pragma unroll
                        for (int q = 0; q < 4; ++q) accm[x][y][4 * p + q] += __uint_as_float(fa[x][p][q] ^ fb[y][p][q]);
        } else if constexpr (TWO_ACC) {
            GW_ALL(0, 0, accm);
            GW_ALL(0, 2, accs); GW_ALL(2, 0, accs); GW_ALL(1, 1, accs); GW_ALL(0, 1, accs); GW_ALL(1, 0, accs);
        } else {
            GW_ALL(0, 2, accm); GW_ALL(2, 0, accm); GW_ALL(1, 1, accm); GW_ALL(0, 1, accm); GW_ALL(1, 0, accm);
            GW_ALL(0, 0, accm);
        }
#undef GW_ALL
#undef GW_MF
    };
    // ---- ring: chunks c + 1 .. c + NS - 2 in flight while chunk c is multiplied
#pragma unroll
    for (int c = 0; c < NS - 1; ++c)
        if (c < nk) issue(c, c);
    int st = 0, st_issue = NS - 1;
    for (int c = 0; c < nk; ++c) {
        // this wave's pieces of chunk c have landed: at most the NS - 2 younger chunks (6 pieces each) are still in flight
        const int younger = nk - 1 - c;
        if (younger >= NS - 2) wait_vm<6 * (NS - 2)>();
        else if (NS > 3 && younger == 1) wait_vm<6>();
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();                       // ... and every wave's; the stage of chunk c - 1 is free
        if (!(PROBE & 2) && c + NS - 1 < nk) issue(c + NS - 1, st_issue);
        chunk_mfmas(st);
        st = st + 1 == NS ? 0 : st + 1;
        st_issue = st_issue + 1 == NS ? 0 : st_issue + 1;
    }
    // ---- epilogue: straight from the accumulators. MFMA tile (x, y): lane (fm, kh) register r holds row 32 x + (r & 3) +
    // 8 (r >> 2) + 4 kh, column 32 y + fm of the wave's 64 x 64: 32 consecutive floats per half wave and store
    float *__restrict__ C = g.C[z];
    if constexpr (STORE16) {
        // the deltas are the MFMA's row operand: a lane holds 4 consecutive COLUMNS of the gradient per register quad -> 16-byte stores
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const int row = i0 + wi * 64 + 32 * x + fm;
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int col = j0 + wj * 64 + 32 * y + 8 * q + 4 * kh;
                    f32x4_t v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = TWO_ACC ? accm[x][y][4 * q + e] + accs[x][y][4 * q + e] : accm[x][y][4 * q + e];
                    if ((PROBE & 4) ? (v[0] == 123.456f) : (row < g.nA && col < g.nB)) *reinterpret_cast<f32x4_t *>(C + (size_t)row * g.ldc + col) = v;
                }
        }
    } else {
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y) {
                const int col = j0 + wj * 64 + 32 * y + fm;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = i0 + wi * 64 + 32 * x + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    const float v = TWO_ACC ? accm[x][y][r] + accs[x][y][r] : accm[x][y][r];
                    if ((PROBE & 4) ? (v == 123.456f) : (row < g.nA && col < g.nB)) {
                        if constexpr (NTS) __builtin_nontemporal_store(v, &C[(size_t)row * g.ldc + col]);
                        else C[(size_t)row * g.ldc + col] = v;
                    }
                }
            }
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The persistent form: ONE workgroup of 8 waves per CU (two per SIMD) walks its list of 128 x 128 tiles with the load ring running
// across tile boundaries, so that a tile's stores drain while the next tile is multiplied (the stores of 32 MB of gradients at the
// end of a launch whose workgroups all finish together cost 4.5 us of 27: profiles/r06_gw_gate.txt).
//   * wave (wi, wj) owns 64 x 32 of the tile: 2 MFMA tiles, 2 x 3 + 1 x 3 fragments per chunk, 12 MFMAs per chunk
//   * a stage = 2 chunks = 32 batch rows = 48 KiB; ring of 3 stages; wave w requests 6 of a stage's 48 pieces (one operand only)
//   * fragments of the NEXT chunk are read while the MFMAs of the current one issue; one barrier per stage, placed between the two
//     chunks' MFMA groups: it publishes stage q + 1 and frees stage q's slot for stage q + 3
//   * vmcnt counts loads AND stores on gfx9: the waits after a tile's epilogue allow its 32 stores per lane to stay in flight
template <bool TWO_ACC, int PROBE = 0>
__global__ void __launch_bounds__(512, 1) gw_bf16x3_persistent_kernel(const GwArgs g)
{
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NS = 3, STAGE = 2 * STAGE_BYTES;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NS][STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 2, wj = wave & 3;
    const int per = g.tiles_i * g.tiles_j, T = per * g.batch;
    const int nst = g.M / 32;                               // stages per tile
    // ---- this workgroup's tiles: id(k) = first + k * stride, k < count. XCD-aware when it divides (workgroup b runs on XCD b % 8:
    // XCD x owns tiles [x T / 8, (x + 1) T / 8))
    int first, stride, count;
    if ((T & 7) == 0 && (gridDim.x & 7) == 0) {
        const int x = blockIdx.x & 7, l = blockIdx.x >> 3, gl = gridDim.x >> 3, tx = T >> 3;
        first = x * tx + l; stride = gl; count = l < tx ? (tx - l + gl - 1) / gl : 0;
    } else {
        first = blockIdx.x; stride = gridDim.x; count = first < T ? (T - first + stride - 1) / stride : 0;
    }
    const bool blocked = ((g.tiles_i | g.tiles_j) & 7) == 0;  // tile ids run through 8 x 8 blocks of tiles (operand slices shared in L2)
    auto decode = [&](int id, int &z, int &i0, int &j0) {
        z = id / per;
        const int r = id - z * per;
        int ti, tj;
        if (blocked) {
            const int blk = r >> 6, l = r & 63, bj = g.tiles_j >> 3;
            ti = (blk / bj) * 8 + (l >> 3);
            tj = (blk % bj) * 8 + (l & 7);
        } else {
            ti = r / g.tiles_j;
            tj = r - ti * g.tiles_j;
        }
        i0 = ti * TILE; j0 = tj * TILE;
    };
    // ---- the issuing side. Piece q of a stage: chunk s = q / 24, then (operand, plane, m8, half) as in the kernel above; wave w
    // requests q = 6 w .. 6 w + 5: chunk w / 4, ONE operand (w % 4 < 2: A)
    const int l_s = wave >> 2, l_op = (wave & 3) >> 1;
    const unsigned plane_bytes = l_op ? g.plane_b_bytes : g.plane_a_bytes;
    const int n_op = l_op ? g.nB : g.nA;
    const unsigned big = 0x7ffffff0u;
    unsigned l_soff[6], l_lds[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) {
        const int r = 6 * (wave & 1) + u, p = r >> 2, m8 = (r >> 1) & 1, hf = r & 1;     // r = piece within the operand's 12
        l_soff[u] = (unsigned)p * plane_bytes + (unsigned)((2 * l_s + m8) * n_op + 64 * hf) * 16u;
        l_lds[u] = (unsigned)(l_s * STAGE_BYTES + (l_op * 12 + r) * 1024);
    }
    const unsigned stage_bytes_g = 4u * (unsigned)n_op * 16u;      // a stage = 4 m8 rows of the plane
    int is_k = 0, is_st = 0;                                       // next stage to request: tile k of the list, stage is_st of it
    __amdgpu_buffer_rsrc_t l_rsrc;
    unsigned l_vo0 = big, l_vo1 = big, l_tile_off = 0;
    auto issue_setup = [&]() {                                     // per tile of the issuing side
        int z, i0, j0;
        decode(first + is_k * stride, z, i0, j0);
        const int o0 = l_op ? j0 : i0;
        l_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(l_op ? g.B[z] : g.A[z]), 0,
                                                   (int)(2u * plane_bytes + (unsigned)(g.M / 8) * (unsigned)n_op * 16u), 0x00020000);
        l_vo0 = (o0 + lane < n_op) ? (unsigned)lane * 16u : big;
        l_vo1 = (o0 + 64 + lane < n_op) ? (unsigned)lane * 16u : big;
        l_tile_off = (unsigned)o0 * 16u;
    };
    auto issue = [&](int slot) {
        if (is_k >= count) return;
        if constexpr (!(PROBE & 2)) {
            const unsigned so = l_tile_off + (unsigned)is_st * stage_bytes_g;
#pragma unroll
            for (int u = 0; u < 6; ++u)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(l_rsrc, &lds[slot][l_lds[u]], 16, (u & 1) ? l_vo1 : l_vo0, l_soff[u] + so, 0, 0);
        }
        if (++is_st == nst) {
            is_st = 0;
            if (++is_k < count) issue_setup();
        }
    };
    // ---- fragments
    const int fm = lane & 31, kh = lane >> 5;
    const int offA = (kh * 128 + wi * 64 + fm) * 16, offB = 12 * 1024 + (kh * 128 + wj * 32 + fm) * 16;
    struct Frags {
        u32x4 a[2][3], b[3];
    };
    auto read_frags = [&](int slot, int chunk, Frags &f) {
        const unsigned char *S = &lds[slot][chunk * STAGE_BYTES];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            f.a[0][p] = *reinterpret_cast<const u32x4 *>(S + offA + p * 4096);
            f.b[p] = *reinterpret_cast<const u32x4 *>(S + offB + p * 4096);
            f.a[1][p] = *reinterpret_cast<const u32x4 *>(S + offA + p * 4096 + 512);
        }
    };
    f32x16 accm[2], accs[2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int r = 0; r < 16; ++r) accm[x][r] = accs[x][r] = 0.f;
    };
    auto mfmas = [&](const Frags &f) {
#define GW_MF(X, PA, PB, ACC)                                                                                                       \
    ACC[X] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.a[X][PA]), __builtin_bit_cast(bf16x8, f.b[PB]), ACC[X], 0, 0, 0)
#define GW_BOTH(PA, PB, ACC) GW_MF(0, PA, PB, ACC); GW_MF(1, PA, PB, ACC)
        if constexpr (PROBE & 1) {
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int q = 0; q < 4; ++q) accm[x][4 * p + q] += __uint_as_float(f.a[x][p][q] ^ f.b[p][q]);
        } else if constexpr (TWO_ACC) {
            GW_BOTH(0, 0, accm);
            GW_BOTH(0, 2, accs); GW_BOTH(2, 0, accs); GW_BOTH(1, 1, accs); GW_BOTH(0, 1, accs); GW_BOTH(1, 0, accs);
        } else {
            GW_BOTH(0, 2, accm); GW_BOTH(2, 0, accm); GW_BOTH(1, 1, accm); GW_BOTH(0, 1, accm); GW_BOTH(1, 0, accm);
            GW_BOTH(0, 0, accm);
        }
#undef GW_BOTH
#undef GW_MF
    };
    // MFMA tile x: lane (fm, kh) register r holds row 32 x + (r & 3) + 8 (r >> 2) + 4 kh, column fm of the wave's 64 x 32
    auto store_tile = [&](int k) -> bool {                   // true: a whole tile = exactly 32 store instructions per wave
        int z, i0, j0;
        decode(first + k * stride, z, i0, j0);
        float *__restrict__ C = g.C[z];
        const int col = j0 + wj * 32 + fm;
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wi * 64 + 32 * x + (r & 3) + 8 * (r >> 2) + 4 * kh;
                const float v = TWO_ACC ? accm[x][r] + accs[x][r] : accm[x][r];
                if ((PROBE & 4) ? (v == 123.456f) : (row < g.nA && col < g.nB)) C[(size_t)row * g.ldc + col] = v;
            }
        return !(PROBE & 4) && i0 + TILE <= g.nA && j0 + TILE <= g.nB;
    };
    if (count == 0) return;
    const int total = count * nst;
    zero_acc();
    issue_setup();
    issue(0); issue(1); issue(2);
    if (total >= 3) wait_vm<12>();
    else if (total == 2) wait_vm<6>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    Frags F0, F1;
    read_frags(0, 0, F0);
    int slot = 0, tile_st = 0, k = 0, grace = 0;            // grace: iterations during which the last epilogue's stores may still fly
    for (int sq = 0; sq < total; ++sq) {
        const int slot1 = slot + 1 == NS ? 0 : slot + 1;
        if (!(PROBE & 8) || sq == 0) read_frags(slot, 1, F1);
        mfmas(F0);
        __builtin_amdgcn_s_waitcnt(0xc07f | (3 << 14));     // lgkmcnt(0): every fragment read of this stage has returned
        if (sq + 2 < total) {
            if (grace > 0) wait_vm<38>();                   // stage sq + 2's pieces and the 32 stores behind / between them
            else wait_vm<6>();
        } else {
            wait_vm<0>();
        }
        if (grace > 0) --grace;
        __builtin_amdgcn_s_barrier();                       // stage sq + 1 is complete for every wave; slot of stage sq is free
        issue(slot);                                        // stage sq + 3
        if (!(PROBE & 8) && sq + 1 < total) read_frags(slot1, 0, F0);
        mfmas(F1);
        if (++tile_st == nst) {
            // a cut tile skips some of its stores: their number is unknown to the counted waits, which then wait for all of them
            grace = store_tile(k) ? 2 : 0;
            zero_acc();
            tile_st = 0; ++k;
        }
        slot = slot1;
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Wider tiles (round 6, after the gate): WM x 64 rows by 128 columns per workgroup of 2 WM waves (64 x 64 each), ring of two stages.
// Operand delivery per output falls with the tile: (64 WM + 128) / (64 WM x 128) -- 128 x 128: 1, 256 x 128: 0.75, 384 x 128: 0.67.
// Pieces of a chunk (1 KiB each: plane, m8, 64 features) are dealt round-robin to the waves; with two stages every wait is vmcnt(0).
template <int WM, bool TWO_ACC>
__global__ void __launch_bounds__(128 * WM, 1) gw_bf16x3_wide_kernel(const GwArgs g)
{
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NW = 2 * WM, ROWS = 64 * WM, A_PIECES = 6 * WM, PIECES = A_PIECES + 12, STAGE = PIECES * 1024;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2][STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 1, wj = wave & 1;
    const int tiles_i = (g.nA + ROWS - 1) / ROWS, tiles_j = g.tiles_j, per = tiles_i * tiles_j;
    int id = blockIdx.x, z, ti, tj;
    {
        const int T = (int)gridDim.x, q8 = T >> 3, r8 = T & 7, x = id & 7;
        id = x * q8 + (x < r8 ? x : r8) + (id >> 3);
        z = id / per;
        const int r = id - z * per, panel = 8 * tiles_j, gidx = r / panel, rows = (tiles_i - 8 * gidx) < 8 ? (tiles_i - 8 * gidx) : 8;
        const int w = r - gidx * panel;
        ti = 8 * gidx + w % rows;
        tj = w / rows;
    }
    const int i0 = ti * ROWS, j0 = tj * TILE;
    const int nk = g.M / KC;
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void *>(g.A[z]), 0, (int)(2u * g.plane_a_bytes + (unsigned)(g.M / 8) * (unsigned)g.nA * 16u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void *>(g.B[z]), 0, (int)(2u * g.plane_b_bytes + (unsigned)(g.M / 8) * (unsigned)g.nB * 16u), 0x00020000);
    const unsigned big = 0x7ffffff0u;
    const unsigned chunk_a = 2u * (unsigned)g.nA * 16u, chunk_b = 2u * (unsigned)g.nB * 16u;
    // piece q: A pieces first -- q = (p * 2 + m8) * WM + seg --, then B -- q - A_PIECES = (p * 2 + m8) * 2 + seg
    constexpr int PER_WAVE = (PIECES + NW - 1) / NW;
    auto issue = [&](int c, int st) {
#pragma unroll
        for (int u = 0; u < PER_WAVE; ++u) {
            const int q = wave + u * NW;
            if (q < A_PIECES) {
                const int pm = q / WM, seg = q - pm * WM, p = pm >> 1, m8 = pm & 1;
                const unsigned so = (unsigned)p * g.plane_a_bytes + (unsigned)(m8 * g.nA + i0 + 64 * seg) * 16u + (unsigned)c * chunk_a;
                const unsigned vo = (i0 + 64 * seg + lane < g.nA) ? (unsigned)lane * 16u : big;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, &lds[st][q * 1024], 16, vo, so, 0, 0);
            } else if (q < PIECES) {
                const int qb = q - A_PIECES, pm = qb >> 1, seg = qb & 1, p = pm >> 1, m8 = pm & 1;
                const unsigned so = (unsigned)p * g.plane_b_bytes + (unsigned)(m8 * g.nB + j0 + 64 * seg) * 16u + (unsigned)c * chunk_b;
                const unsigned vo = (j0 + 64 * seg + lane < g.nB) ? (unsigned)lane * 16u : big;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, &lds[st][q * 1024], 16, vo, so, 0, 0);
            }
        }
    };
    const int fm = lane & 31, kh = lane >> 5;
    const int offA = (kh * ROWS + wi * 64 + fm) * 16, offB = A_PIECES * 1024 + (kh * 128 + wj * 64 + fm) * 16;
    f32x16 accm[2][2], accs[2][2];
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
                fa[x][p] = *reinterpret_cast<const u32x4 *>(S + offA + p * (2 * ROWS * 16) + x * 512);
                fb[x][p] = *reinterpret_cast<const u32x4 *>(S + offB + p * 4096 + x * 512);
            }
#define GW_MF(I, J, PA, PB, ACC)                                                                                                    \
    ACC[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[I][PA]), __builtin_bit_cast(bf16x8, fb[J][PB]), \
                                                        ACC[I][J], 0, 0, 0)
#define GW_ALL(PA, PB, ACC) GW_MF(0, 0, PA, PB, ACC); GW_MF(0, 1, PA, PB, ACC); GW_MF(1, 0, PA, PB, ACC); GW_MF(1, 1, PA, PB, ACC)
        if constexpr (TWO_ACC) {
            GW_ALL(0, 0, accm);
            GW_ALL(0, 2, accs); GW_ALL(2, 0, accs); GW_ALL(1, 1, accs); GW_ALL(0, 1, accs); GW_ALL(1, 0, accs);
        } else {
            GW_ALL(0, 2, accm); GW_ALL(2, 0, accm); GW_ALL(1, 1, accm); GW_ALL(0, 1, accm); GW_ALL(1, 0, accm);
            GW_ALL(0, 0, accm);
        }
#undef GW_ALL
#undef GW_MF
    };
    issue(0, 0);
    int st = 0;
    for (int c = 0; c < nk; ++c) {
        wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if (c + 1 < nk) issue(c + 1, st ^ 1);
        chunk_mfmas(st);
        st ^= 1;
    }
    float *__restrict__ C = g.C[z];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const int col = j0 + wj * 64 + 32 * y + fm;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wi * 64 + 32 * x + (r & 3) + 8 * (r >> 2) + 4 * kh;
                const float v = TWO_ACC ? accm[x][y][r] + accs[x][y][r] : accm[x][y][r];
                if (row < g.nA && col < g.nB) C[(size_t)row * g.ldc + col] = v;
            }
        }
#endif
}

// X [M][N] (pitch ldx) -> three bf16 planes in the layout above; the producers' epilogues do this themselves, this launch serves
// the inputs that have no producer kernel of ours and the tests. One lane = 8 batch rows of one feature.
__global__ void __launch_bounds__(256) split_planes_kernel(const float *__restrict__ X, int M, int N, int ldx, void *__restrict__ planes,
                                                           unsigned plane_bytes)
{
    const int n = (int)(blockIdx.x * 256u + threadIdx.x), m8 = (int)blockIdx.y;
    if (n >= N) return;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = X[(size_t)(8 * m8 + j) * ldx + n];
    u32x4 p0, p1, p2;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const Pair3 s = split_pair(v[2 * q], v[2 * q + 1]);
        p0[q] = s.p0; p1[q] = s.p1; p2[q] = s.p2;
    }
    unsigned char *base = static_cast<unsigned char *>(planes) + ((size_t)m8 * N + n) * 16u;
    *reinterpret_cast<u32x4 *>(base) = p0;
    *reinterpret_cast<u32x4 *>(base + plane_bytes) = p1;
    *reinterpret_cast<u32x4 *>(base + 2 * (size_t)plane_bytes) = p2;
}

}  // namespace sgmcmc_gw
