// bnn_dense_bf16x3.hip -- round 5 gate experiment, stand-alone harness.
//
// The forward hidden layer of the BNN (pysgmcmc/models/bayesian_neural_network.py:30-52)
//   out[m][n] = tanh( sum_k h[m][k] W[k][n] + b[n] )          h [M][K], W [K][N] row-major, fp32
// with the product formed on the bf16 matrix pipe at fp32 accuracy: every fp32 operand is split IN REGISTERS, exactly, into three
// bf16 planes (x = x0 + x1 + x2, 8 + 8 + 8 significant bits, truncation splits), and the six partial products of order <= 2^-16
// (a0 b0, a0 b1, a1 b0, a0 b2, a1 b1, a2 b0) are accumulated in fp32 by v_mfma_f32_32x32x16_bf16 -- 6 x 1/16 of the fp32 MFMA time.
//
// Pipeline (built on csrc/sgmcmc_bnn_gemm.hip's): one workgroup of 8 waves per 32 x 64 output tile (one per CU at batch 256 x 2048
// columns); fp32 operands go from global memory straight into LDS (buffer_load_dwordx4 ... lds) in 128-deep K chunks, ring of 3 stages
// x 48 KB; wave w owns the 16-deep slice w of every chunk for BOTH 32 x 32 MFMA tiles (1 A fragment, 2 B fragments: 24 fp32 per lane
// per chunk), splits them (11 VALU per pair of elements) while the 12 MFMAs of the previous chunk issue, and the 8 partial tiles
// meet in LDS in a fixed order before the bias + tanh epilogue.
//
// Build + run (GPU box): make -C tools/gpu bnn_dense_bf16x3 && tools/gpu/bnn_dense_bf16x3 [path/to/libsgmcmc_hip.so]
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#pragma clang fp contract(off)

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 32, BN = 64, BK = 128, NW = 8, NS = 3;
constexpr int TP = BN + 4;

constexpr int vmcnt_imm(int n) { return (n & 15) | (7 << 4) | (15 << 8) | ((n >> 4) << 14); }
template <int N>
__device__ __forceinline__ void wait_vm()
{
    __builtin_amdgcn_s_waitcnt(vmcnt_imm(N));
}
__device__ __forceinline__ void wait_lgkm0() { __builtin_amdgcn_s_waitcnt(0xc07f | (3 << 14)); }     // lgkmcnt(0), vmcnt / expcnt untouched

struct Args {
    const float *h, *W, *bias;
    float *out;
    int M, N, K, ldh, ldw, ldo;
};

struct __attribute__((aligned(16))) Lds {
    union {
        struct {
            float A[NS][BM][BK];
            float B[NS][BK][BN];
        } ring;
        float T[NW][BM][TP];
    };
};

struct Planes {                 // three bf16 planes of one A fragment and two B fragments (8 k values per lane each)
    u32x4 a[3], b[2][3];
};
struct Raw {
    f32x4_t a0, a1;
    float b[2][8];
};

// (x, y) = two consecutive k: exact split into three packed bf16 pairs, low half = x
struct Pair3 {
    unsigned p0, p1, p2;
};
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ Pair3 split_pair(float x, float y)
{
    // 9 vector instructions per pair: v_perm (plane pair), 2 x v_and (the plane as fp32), v_pk_add_f32 (both exact remainders), twice; v_perm
    // measured on gfx950 (tools/gpu/valu_rate.hip): v_and / v_sub 2 SIMD cycles per wave-instruction, v_perm / v_pk_add_f32 /
    // v_pack_b32_f16 / v_cvt_pk_bf16_f32 / v_dot2c_f32_bf16 4 -> 28 cycles per pair either way; v_dot2c (x - x0 from the packed plane) is not exact
    const unsigned xb = __float_as_uint(x), yb = __float_as_uint(y);
    const unsigned p0 = __builtin_amdgcn_perm(yb, xb, 0x07060302u);
    const float xr = x - __uint_as_float(xb & 0xffff0000u), yr = y - __uint_as_float(yb & 0xffff0000u);
    const unsigned xrb = __float_as_uint(xr), yrb = __float_as_uint(yr);
    const unsigned p1 = __builtin_amdgcn_perm(yrb, xrb, 0x07060302u);
    const float xl = xr - __uint_as_float(xrb & 0xffff0000u), yl = yr - __uint_as_float(yrb & 0xffff0000u);
    const unsigned p2 = __builtin_amdgcn_perm(__float_as_uint(yl), __float_as_uint(xl), 0x07060302u);
    return {p0, p1, p2};
}

// PROBE bits: 1 = no split (raw bits as planes: wrong numbers, timing only), 2 = no MFMA, 4 = no loads in the loop
template <int PROBE>
__global__ void __launch_bounds__(512, 2) dense_tanh_bf16x3_kernel(const Args g)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ Lds lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_m = g.M / BM, tiles = tiles_m * (g.N / BN);
    int t = blockIdx.x;
    if (tiles % 8 == 0) t = (t & 7) * (tiles >> 3) + (t >> 3);
    const int n0 = (t / tiles_m) * BN, m0 = (t % tiles_m) * BM;
    const int nk = (g.K + BK - 1) / BK;
    // ---- direct loads. A stage [32 m][128 k]: wave w requests rows 4 w .. 4 w + 3 (two 1 KiB pieces of two rows), quads XOR-swizzled
    // with the row & 15 on the GLOBAL side. B stage [128 k][64 n]: wave w requests rows 16 w .. 16 w + 15 (four pieces of four rows).
    const int ar0 = 4 * wave + (lane >> 5), ar1 = ar0 + 2;
    const unsigned a_lane0 = (unsigned)(ar0 * g.ldh + 4 * ((lane & 31) ^ (ar0 & 15))) * 4u;
    const unsigned a_lane1 = (unsigned)(ar1 * g.ldh + 4 * ((lane & 31) ^ (ar1 & 15))) * 4u;
    const unsigned b_lane = (unsigned)((16 * wave + (lane >> 4)) * g.ldw + 4 * (lane & 15)) * 4u;
    const unsigned b_chunk = (unsigned)BK * (unsigned)g.ldw * 4u, b_rows4 = 4u * (unsigned)g.ldw * 4u;
    // exact extents: lanes beyond them (K tails) are dropped by the buffer range check
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(g.h + (size_t)m0 * g.ldh), 0, (int)(((BM - 1) * g.ldh + g.K) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(g.W + n0), 0, (int)((((size_t)g.K - 1) * g.ldw + BN) * 4), 0x00020000);
    auto issue = [&](int kc, int st) {
        const unsigned sa = (unsigned)kc * (BK * 4), sb = (unsigned)kc * b_chunk;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, &lds.ring.A[st][4 * wave][0], 16, a_lane0, sa, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, &lds.ring.A[st][4 * wave + 2][0], 16, a_lane1, sa, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, &lds.ring.B[st][16 * wave + 4 * u][0], 16, b_lane, sb + u * b_rows4, 0, 0);
    };
    // ---- fragment addresses: lane (fm, kh) holds k = 16 w + 8 kh + j, j = 0 .. 7, of row fm (A) / of columns fm, 32 + fm (B)
    const int fm = lane & 31, kh = lane >> 5;
    const int aoff0 = fm * BK + 4 * ((4 * wave + 2 * kh) ^ (fm & 15)), aoff1 = fm * BK + 4 * ((4 * wave + 2 * kh + 1) ^ (fm & 15));
    // two base registers the compiler cannot relate: its ds_read2_b32 pairs are then (k, k + 1) of ONE column -- the register pair
    // v_pk_add_f32 wants -- instead of (column, column + 32) of one k
    int boff = (16 * wave + 8 * kh) * BN + fm, boff1 = boff + 32;
    asm volatile("" : "+v"(boff1));
    auto read_raw = [&](int st, Raw &r) {
        const float *A = &lds.ring.A[st][0][0];
        const float *B = &lds.ring.B[st][0][0];
        r.a0 = *reinterpret_cast<const f32x4_t *>(A + aoff0);
        r.a1 = *reinterpret_cast<const f32x4_t *>(A + aoff1);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            r.b[0][j] = B[boff + j * BN];
            r.b[1][j] = B[boff1 + j * BN];
        }
    };
    auto split = [&](const Raw &r, Planes &p) {
        if constexpr (PROBE & 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float x = q < 2 ? r.a0[2 * q] : r.a1[2 * q - 4];
                p.a[0][q] = p.a[1][q] = p.a[2][q] = __float_as_uint(x);
                p.b[0][0][q] = p.b[0][1][q] = p.b[0][2][q] = __float_as_uint(r.b[0][2 * q]) ^ __float_as_uint(r.b[0][2 * q + 1]);
                p.b[1][0][q] = p.b[1][1][q] = p.b[1][2][q] = __float_as_uint(r.b[1][2 * q]) ^ __float_as_uint(r.b[1][2 * q + 1]);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float x = q < 2 ? r.a0[2 * q] : r.a1[2 * q - 4], y = q < 2 ? r.a0[2 * q + 1] : r.a1[2 * q - 3];
                const Pair3 sa = (PROBE & 8) ? Pair3{__float_as_uint(x), __float_as_uint(y), __float_as_uint(x) ^ __float_as_uint(y)} : split_pair(x, y), s0 = split_pair(r.b[0][2 * q], r.b[0][2 * q + 1]), s1 = split_pair(r.b[1][2 * q], r.b[1][2 * q + 1]);
                p.a[0][q] = sa.p0; p.a[1][q] = sa.p1; p.a[2][q] = sa.p2;
                p.b[0][0][q] = s0.p0; p.b[0][1][q] = s0.p1; p.b[0][2][q] = s0.p2;
                p.b[1][0][q] = s1.p0; p.b[1][1][q] = s1.p1; p.b[1][2][q] = s1.p2;
            }
        }
    };
    f32x16 accm[2], accs[2];                                // per MFMA tile: a0 b0 sums, and the five small partial products
#pragma unroll
    for (int r = 0; r < 16; ++r) accm[0][r] = accm[1][r] = accs[0][r] = accs[1][r] = 0.f;
#define MF(T, I, J, ACC)                                                                                                            \
    ACC[T] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, p.a[I]), __builtin_bit_cast(bf16x8, p.b[T][J]), ACC[T], 0, 0, 0)
    auto mfmas = [&](const Planes &p) {
        if constexpr (!(PROBE & 2)) {
            MF(0, 0, 0, accm); MF(1, 0, 0, accm);
            MF(0, 0, 2, accs); MF(1, 0, 2, accs);
            MF(0, 2, 0, accs); MF(1, 2, 0, accs);
            MF(0, 1, 1, accs); MF(1, 1, 1, accs);
            MF(0, 0, 1, accs); MF(1, 0, 1, accs);
            MF(0, 1, 0, accs); MF(1, 1, 0, accs);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                accs[0][q] += __uint_as_float(p.a[0][q] ^ p.a[1][q] ^ p.a[2][q] ^ p.b[0][0][q] ^ p.b[0][1][q] ^ p.b[0][2][q]);
                accs[1][q] += __uint_as_float(p.b[1][0][q] ^ p.b[1][1][q] ^ p.b[1][2][q]);
            }
        }
    };
    // ---- prologue: chunks 0 .. 2 requested, chunk 0 read and split
    constexpr bool LOADS = !(PROBE & 4);
#pragma unroll
    for (int c = 0; c < 3; ++c)
        if (c < nk) issue(c, c);
    if (nk >= 3) wait_vm<12>();
    else if (nk == 2) wait_vm<6>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    Planes P0, P1;
    Raw raw;
    read_raw(0, raw);
    wait_lgkm0();
    split(raw, P0);
    // iteration c: chunk c + 1 has landed -> its fragments are read and split into `nxt` while the MFMAs of chunk c issue from `cur`;
    // chunk c + 3 is requested into the stage chunk c left (every wave read it during iteration c - 1).
    // STEADY: chunks c + 2 and c + 3 exist -- no branches, so the MFMAs and the split share one scheduling region and interleave
    auto step = [&](int c, const Planes &cur, Planes &nxt, auto steady) {
        constexpr bool STEADY = decltype(steady)::value;
        if constexpr (STEADY) {
            wait_vm<6>();                                   // chunk c + 2 may be in flight
        } else {
            if (c + 2 < nk) wait_vm<6>();
            else wait_vm<0>();
        }
        __builtin_amdgcn_s_barrier();
        read_raw((c + 1) % NS, raw);
        if constexpr (STEADY) {
            if (LOADS) issue(c + 3, c % NS);
        } else {
            if (LOADS && c + 3 < nk) issue(c + 3, c % NS);
        }
        mfmas(cur);
        split(raw, nxt);
        if constexpr (STEADY && (PROBE & 3) == 0) {
            constexpr int PER = (PROBE & 8) ? 10 : 15;     // vector instructions between MFMAs: 16 (24) pairs x 11 / 9 gaps
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x002, PER, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
        }
    };
    int c = 0;
    for (; c + 4 < nk; c += 2) {
        step(c, P0, P1, std::true_type());
        step(c + 1, P1, P0, std::true_type());
    }
    for (; c + 2 <= nk - 1; c += 2) {
        step(c, P0, P1, std::false_type());
        step(c + 1, P1, P0, std::false_type());
    }
    const bool last_valid = (nk - 1) * BK + 16 * wave < g.K;    // the last chunk may hold fewer than 8 slices (K % 16 == 0)
    if (c + 1 <= nk - 1) {
        step(c, P0, P1, std::false_type());
        if (last_valid) mfmas(P1);
    } else if (last_valid) {
        mfmas(P0);
    }
#undef MF
    // ---- epilogue: the 8 partial tiles meet in LDS (fixed order), bias + tanh on row-major quads, 16-byte stores
    __syncthreads();
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) lds.T[wave][(r & 3) + 8 * (r >> 2) + 4 * kh][32 * tt + fm] = accm[tt][r] + accs[tt][r];
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
        const f32x4_t b = *reinterpret_cast<const f32x4_t *>(g.bias + n0 + c4);
        f32x4_t v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = tanhf(s[j] + b[j]);
        *reinterpret_cast<f32x4_t *>(g.out + (size_t)(m0 + row) * g.ldo + n0 + c4) = v;
    }
#endif
}

__global__ void fill(float *x, size_t n, unsigned seed, float scale, int squash)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed;
        float s = 0.f;
        for (int r = 0; r < 4; ++r) {
            h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
            s += (float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f;
        }
        x[i] = squash ? tanhf(3.f * s) : s * scale;
    }
}

__global__ void ref_f64(const float *__restrict__ h, const float *__restrict__ W, const float *__restrict__ b, double *__restrict__ out,
                        int M, int N, int K)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N) return;
    double acc = 0;
    for (int k = 0; k < K; ++k) acc = fma((double)h[(size_t)m * K + k], (double)W[(size_t)k * N + n], acc);
    out[(size_t)m * N + n] = tanh(acc + (double)b[n]);
}

template <typename F>
float time_us(F f, int reps = 50)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) f();
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1e3f;
}

void errors(const char *what, const std::vector<float> &c, const std::vector<double> &ref)
{
    double maxabs = 0, sumsq = 0, refsq = 0;
    for (size_t i = 0; i < c.size(); ++i) {
        const double e = (double)c[i] - ref[i];
        maxabs = fmax(maxabs, fabs(e)); sumsq += e * e; refsq += ref[i] * ref[i];
    }
    printf("    %-44s max|err| %.3e   rms err / rms value %.3e\n", what, maxabs, sqrt(sumsq / refsq));
}

template <int PROBE>
void launch(const Args &g)
{
    hipLaunchKernelGGL((dense_tanh_bf16x3_kernel<PROBE>), dim3((g.M / BM) * (g.N / BN)), dim3(512), 0, 0, g);
}

typedef int (*fp32_fn)(const float *, const float *, const float *, float *, int, int, int, int, int, int, const float *, float *,
                       const void *, double *, void *);

}  // namespace

int main(int argc, char **argv)
{
    fp32_fn fp32 = nullptr;
    if (void *lib = dlopen(argc > 1 ? argv[1] : "pysgmcmc_amd/csrc/libsgmcmc_hip.so", RTLD_NOW))
        fp32 = reinterpret_cast<fp32_fn>(dlsym(lib, "sgmcmc_bnn_dense_tanh_f32"));
    if (!fp32) printf("(libsgmcmc_hip.so not loaded: no fp32 MFMA column)\n");
    const int M = 256, N = 2048;
    for (int K : {2048, 784, 64, 400}) {
        float *h, *W, *b, *out;
        double *D;
        hipMalloc(&h, (size_t)M * K * 4); hipMalloc(&W, (size_t)K * N * 4); hipMalloc(&b, N * 4); hipMalloc(&out, (size_t)M * N * 4);
        hipMalloc(&D, (size_t)M * N * 8);
        for (int data = 0; data < (K == 2048 ? 2 : 1); ++data) {
            hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, h, (size_t)M * K, 3u, 2.0f, data == 0);          // tanh outputs / wide
            hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, W, (size_t)K * N, 4u, 3.5f / sqrtf((float)K), 0);
            hipLaunchKernelGGL(fill, dim3(8), dim3(256), 0, 0, b, (size_t)N, 5u, 0.2f, 0);
            std::vector<float> c((size_t)M * N);
            std::vector<double> ref((size_t)M * N);
            hipLaunchKernelGGL(ref_f64, dim3((N + 255) / 256, M), dim3(256), 0, 0, h, W, b, D, M, N, K);
            hipMemcpy(ref.data(), D, ref.size() * 8, hipMemcpyDeviceToHost);
            printf("M=%d N=%d K=%d, h = %s\n", M, N, K, data == 0 ? "tanh outputs" : "bell-shaped, |h| < 4");
            const Args g{h, W, b, out, M, N, K, K, N, N};
            if (fp32) {
                hipMemset(out, 0, c.size() * 4);
                int rc = fp32(h, W, b, out, M, N, K, K, N, N, nullptr, nullptr, nullptr, nullptr, nullptr);
                hipMemcpy(c.data(), out, c.size() * 4, hipMemcpyDeviceToHost);
                if (rc) printf("    fp32 kernel refused (%d)\n", rc);
                else {
                    errors("fp32 MFMA kernel (libsgmcmc_hip.so)", c, ref);
                    printf("        %.2f us per launch\n", time_us([&] { fp32(h, W, b, out, M, N, K, K, N, N, nullptr, nullptr, nullptr, nullptr, nullptr); }));
                }
            }
            hipMemset(out, 0, c.size() * 4);
            launch<0>(g);
            hipError_t e = hipDeviceSynchronize();
            if (e != hipSuccess) { printf("    bf16x3 kernel failed: %s\n", hipGetErrorString(e)); return 1; }
            hipMemcpy(c.data(), out, c.size() * 4, hipMemcpyDeviceToHost);
            errors("bf16 x 3 planes, 6 partial products", c, ref);
            printf("        %.2f us per launch", time_us([&] { launch<0>(g); }));
            if (K == 2048)
                printf("   | no split %.2f | no MFMA %.2f | no loads in the loop %.2f | loads only %.2f | A not split %.2f | A not split, no loads %.2f",
                       time_us([&] { launch<1>(g); }), time_us([&] { launch<2>(g); }), time_us([&] { launch<4>(g); }), time_us([&] { launch<3>(g); }),
                       time_us([&] { launch<8>(g); }), time_us([&] { launch<12>(g); }));
            printf("\n");
        }
        hipFree(h); hipFree(W); hipFree(b); hipFree(out); hipFree(D);
    }
    return 0;
}
