// Prototype (stand-alone, not part of the library): an fp32-accurate product on the bf16 matrix pipe.
//
//   C[M][N] = A^T B,  A [K][M], B [K][N] fp32 (the weight-gradient shape: K = batch)
//
// Every fp32 operand is split EXACTLY into three bf16 parts (8 + 8 + 8 significant bits, truncation splits, so
// x = hi + mid + lo with no rounding), stored as planes in the layout the matrix instruction wants
// (P[part][k / 8][m][8 consecutive k]: a lane's 16-byte operand of v_mfma_f32_32x32x16_bf16), and the product is formed from the
// 9 (or the 6 largest) partial products part_i(A) x part_j(B), each exact in fp32, accumulated in the fp32 accumulators.
// The planes go from global memory straight into LDS (global_load_lds_dwordx4), ring of NS stages, like the fp32 kernel of
// pysgmcmc_amd/csrc/sgmcmc_gemm.hip. Reported: error against an fp64 product next to the error of a k-ordered fp32 fmaf chain
// (= what v_mfma_f32_32x32x2_f32 and the library's fp32 GEMMs compute, up to summation order), and microseconds.
//
// Build + run: hipcc --offload-arch=gfx950 -O3 -o /tmp/gemm_bf16x9 tools/gpu/gemm_bf16x9.hip && /tmp/gemm_bf16x9
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

__global__ void fill(float *x, size_t n, unsigned seed, float scale)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed;
        float s = 0.f;
        for (int r = 0; r < 4; ++r) {                      // sum of 4 uniforms: bell-shaped, varying exponents
            h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
            s += (float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f;
        }
        x[i] = s * scale;
    }
}

// X [K][M] -> P [3][K / 8][M][8]
__global__ void split_planes(const float *__restrict__ X, u16 *__restrict__ P, int K, int M)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)(K / 8) * M;
    if (idx >= total) return;
    const int m = (int)(idx % M), ko = (int)(idx / M);
    u16 hi[8], mid[8], lo[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = X[(size_t)(8 * ko + j) * M + m];
        const unsigned xb = __float_as_uint(x) & 0xffff0000u;
        const float r = x - __uint_as_float(xb);           // exact: the low 16 bits of the significand
        const unsigned rb = __float_as_uint(r) & 0xffff0000u;
        const float l = r - __uint_as_float(rb);           // exact: at most 8 significant bits left
        hi[j] = (u16)(xb >> 16); mid[j] = (u16)(rb >> 16); lo[j] = (u16)(__float_as_uint(l) >> 16);
    }
    const size_t plane = (size_t)(K / 8) * M * 8;
    u16 *p = P + ((size_t)ko * M + m) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) { p[j] = hi[j]; p[plane + j] = mid[j]; p[2 * plane + j] = lo[j]; }
}

template <int BK, int NS, int TNW>
struct Ring {
    u16 A[NS][3][BK / 8][64][8];
    u16 B[NS][3][BK / 8][64 * TNW][8];
};

// TERMS = 9: every partial product; 6: without mid x lo, lo x mid, lo x lo (each <= 2^-24 of the product)
// TNW: 32 x 32 output tiles per wave along n (workgroup tile 64 x 64 TNW): the A fragments are reused TNW times
template <int BK, int NS, int TERMS, bool TWO_ACC = false, bool STORE = true, int TNW = 1>
__global__ void __launch_bounds__(256) gemm_tn_bf16x(const u16 *__restrict__ PA, const u16 *__restrict__ PB, float *__restrict__ C,
                                                     int M, int N, int K)
{
#if defined(__HIP_DEVICE_COMPILE__)                        /* the host pass has no declaration of the gfx950 builtins */
    __shared__ Ring<BK, NS, TNW> lds;
    constexpr int KO = BK / 8;                             // k-octets per chunk
    constexpr int ITEMS = (1 + TNW) * 3 * KO, PER_WAVE = (ITEMS + 3) / 4;
    static_assert(PER_WAVE <= 15, "wait counts");
    const bool full_share = ITEMS % 4 == 0 || __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) < ITEMS % 4;      // this wave issues PER_WAVE loads per chunk (else one less)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform: scalar branches below
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64 * TNW;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32 * TNW;
    const size_t planeA = (size_t)(K / 8) * M * 8, planeB = (size_t)(K / 8) * N * 8;
    // split-K: slice blockIdx.z of gridDim.z works on k in [z K / Z, (z + 1) K / Z) and writes its own partial C + z M N
    const int kc0 = blockIdx.z * (K / (int)gridDim.z / BK);
    C += (size_t)blockIdx.z * M * N;
    auto issue = [&](int kc, int st) {
#pragma unroll
        for (int u = 0; u < PER_WAVE; ++u) {
            const int item = wave + 4 * u;                 // items 0 .. 3 KO - 1: A; then B in 64-column slabs
            if (item >= ITEMS) break;
            const int op = item / (3 * KO), part = (item / KO) % 3, ko = item % KO;
            const int kog = (kc0 + kc) * KO + ko;
            if (op == 0)
                __builtin_amdgcn_global_load_lds(PA + part * planeA + ((size_t)kog * M + m0) * 8 + lane * 8, &lds.A[st][part][ko][0][0], 16, 0, 0);
            else
                __builtin_amdgcn_global_load_lds(PB + part * planeB + ((size_t)kog * N + n0 + 64 * (op - 1)) * 8 + lane * 8,
                                                 &lds.B[st][part][ko][64 * (op - 1)][0], 16, 0, 0);
        }
    };
    f32x16 accs[TNW], acc2s[TNW];                          // TWO_ACC: the partial products alternate between two accumulators
#pragma unroll
    for (int t = 0; t < TNW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { accs[t][r] = 0.f; acc2s[t][r] = 0.f; }
    const int nk = K / (int)gridDim.z / BK;
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    const int kh = lane >> 5, cl = lane & 31;
    for (int kc = 0; kc < nk; ++kc) {
        if (kc + 1 >= nk) __builtin_amdgcn_s_waitcnt(0x0F70);
        else if (full_share) __builtin_amdgcn_s_waitcnt(0x0F70 | PER_WAVE);
        else __builtin_amdgcn_s_waitcnt(0x0F70 | (PER_WAVE - 1));
        __builtin_amdgcn_s_barrier();
        const int st = kc % NS;
        bf16x8 a[BK / 16][3], bb[BK / 16][TNW][3];
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                a[ks][p] = *reinterpret_cast<const bf16x8 *>(&lds.A[st][p][2 * ks + kh][wm + cl][0]);
#pragma unroll
                for (int t = 0; t < TNW; ++t)
                    bb[ks][t][p] = *reinterpret_cast<const bf16x8 *>(&lds.B[st][p][2 * ks + kh][wn + 32 * t + cl][0]);
            }
        if (kc + 2 < nk) issue(kc + 2, (kc + 2) % NS);
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks)
#pragma unroll
        for (int t = 0; t < TNW; ++t) {
            f32x16 &acc = accs[t], &acc2 = acc2s[t];
            // small terms first; with TWO_ACC consecutive instructions never depend on each other
#define MF(I, J, ACC) ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][I], bb[ks][t][J], ACC, 0, 0, 0)
            if (TWO_ACC) {
                if (TERMS == 9) { MF(2, 2, acc2); MF(1, 2, acc); MF(2, 1, acc2); }
                MF(0, 2, acc); MF(2, 0, acc2); MF(1, 1, acc); MF(0, 1, acc2); MF(1, 0, acc); MF(0, 0, acc2);
            } else {
                if (TERMS == 9) { MF(2, 2, acc); MF(1, 2, acc); MF(2, 1, acc); }
                MF(0, 2, acc); MF(2, 0, acc); MF(1, 1, acc); MF(0, 1, acc); MF(1, 0, acc); MF(0, 0, acc);
            }
#undef MF
        }
    }
#pragma unroll
    for (int t = 0; t < TNW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const float v = accs[t][r] + acc2s[t][r];
            if (STORE || v == 1.2345e33f) C[(size_t)row * N + n0 + wn + 32 * t + (lane & 31)] = v;
        }
#endif
}

// C[i] = sum_z P[z][i] (in the product this sum would ride in the consumer: the bias + tanh launch)
__global__ void sum_partials(const float *__restrict__ P, float *__restrict__ C, size_t n, int Z)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int z = 0; z < Z; ++z) s += P[(size_t)z * n + i];
    C[i] = s;
}

// references: one thread per output element, k-ordered chain
template <typename T>
__global__ void ref_tn(const float *__restrict__ A, const float *__restrict__ B, T *__restrict__ C, int M, int N, int K)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N) return;
    T acc = T(0);
    for (int k = 0; k < K; ++k) acc = fma((T)A[(size_t)k * M + m], (T)B[(size_t)k * N + n], acc);
    C[(size_t)m * N + n] = acc;
}

template <int BK, int NS, int TERMS, bool TWO_ACC = false, bool STORE = true, int TNW = 1>
constexpr int tnw_of() { return TNW; }

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
    double maxabs = 0, sumsq = 0, refsq = 0, bias = 0;
    for (size_t i = 0; i < c.size(); ++i) {
        const double e = (double)c[i] - ref[i];
        maxabs = fmax(maxabs, fabs(e)); sumsq += e * e; refsq += ref[i] * ref[i]; bias += e;
    }
    printf("    %-42s max|err| %.3e   rms err / rms value %.3e   mean signed err / rms value %+.3e\n", what, maxabs,
           sqrt(sumsq / refsq), bias / c.size() / sqrt(refsq / c.size()));
}

int main()
{
    const int K = 256;
    for (int size : {2048, 768}) {
        const int M = size, N = 2048;
        float *A, *B, *C;
        double *D;
        u16 *PA, *PB;
        hipMalloc(&A, (size_t)K * M * 4); hipMalloc(&B, (size_t)K * N * 4); hipMalloc(&C, (size_t)M * N * 4);
        hipMalloc(&D, (size_t)M * N * 8);
        hipMalloc(&PA, (size_t)3 * K * M * 2); hipMalloc(&PB, (size_t)3 * K * N * 2);
        hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, A, (size_t)K * M, 1u, 2.0f);       // activations ~ O(1)
        hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, B, (size_t)K * N, 2u, 0.02f);      // deltas ~ O(1e-2)
        std::vector<float> c((size_t)M * N);
        std::vector<double> ref((size_t)M * N);
        hipLaunchKernelGGL(ref_tn<double>, dim3((N + 255) / 256, M), dim3(256), 0, 0, A, B, D, M, N, K);
        hipMemcpy(ref.data(), D, ref.size() * 8, hipMemcpyDeviceToHost);
        printf("M=%d N=%d K=%d\n", M, N, K);
        hipLaunchKernelGGL(ref_tn<float>, dim3((N + 255) / 256, M), dim3(256), 0, 0, A, B, C, M, N, K);
        hipMemcpy(c.data(), C, c.size() * 4, hipMemcpyDeviceToHost);
        errors("fp32 fmaf chain in k order (= fp32 MFMA)", c, ref);
        auto split = [&] {
            hipLaunchKernelGGL(split_planes, dim3((unsigned)(((size_t)(K / 8) * M + 255) / 256)), dim3(256), 0, 0, A, PA, K, M);
            hipLaunchKernelGGL(split_planes, dim3((unsigned)(((size_t)(K / 8) * N + 255) / 256)), dim3(256), 0, 0, B, PB, K, N);
        };
        split();
#define RUN(BK, NS, TERMS, ...)                                                                                                    \
        {                                                                                                                     \
            hipMemset(C, 0, (size_t)M * N * 4);                                                                               \
            const dim3 grid(N / 64 / tnw_of<BK, NS, TERMS, ##__VA_ARGS__>(), M / 64);                                               \
            auto f = [&] { hipLaunchKernelGGL((gemm_tn_bf16x<BK, NS, TERMS, ##__VA_ARGS__>), grid, dim3(256), 0, 0, PA, PB, C, M, N, K); };  \
            f();                                                                                                              \
            hipMemcpy(c.data(), C, c.size() * 4, hipMemcpyDeviceToHost);                                                      \
            char what[96];                                                                                                    \
            snprintf(what, sizeof what, "bf16 x %d partial products, BK %d, ring %d %s", TERMS, BK, NS, #__VA_ARGS__);                        \
            errors(what, c, ref);                                                                                             \
            printf("        %.1f us per launch (planes already split)\n", time_us(f));                                      \
        }
        RUN(32, 3, 9) RUN(32, 3, 9, true) RUN(32, 3, 6, true) RUN(16, 3, 9, true, true, 2) RUN(16, 3, 6, true, true, 2) RUN(16, 3, 9, true, false, 2) RUN(16, 4, 9, true, true, 2)
        printf("    splitting both operands into planes: %.1f us\n", time_us(split));
        hipFree(A); hipFree(B); hipFree(C); hipFree(D); hipFree(PA); hipFree(PB);
    }
    {
        // the long-K shape of the forward / delta products: 256 x 2048 outputs, K = 2048 -> 128 tiles of 64 x 64, so the K range
        // is split over blockIdx.z and the partial outputs are summed by a second launch (the consumer's job in a pipeline)
        const int M = 256, N = 2048, KL = 2048;
        float *A, *B, *C, *P;
        double *D;
        u16 *PA, *PB;
        hipMalloc(&A, (size_t)KL * M * 4); hipMalloc(&B, (size_t)KL * N * 4); hipMalloc(&C, (size_t)M * N * 4);
        hipMalloc(&P, (size_t)16 * M * N * 4); hipMalloc(&D, (size_t)M * N * 8);
        hipMalloc(&PA, (size_t)3 * KL * M * 2); hipMalloc(&PB, (size_t)3 * KL * N * 2);
        hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, A, (size_t)KL * M, 3u, 2.0f);      // activations
        hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, B, (size_t)KL * N, 4u, 0.05f);     // weights
        std::vector<float> c((size_t)M * N);
        std::vector<double> ref((size_t)M * N);
        hipLaunchKernelGGL(ref_tn<double>, dim3((N + 255) / 256, M), dim3(256), 0, 0, A, B, D, M, N, KL);
        hipMemcpy(ref.data(), D, ref.size() * 8, hipMemcpyDeviceToHost);
        printf("M=%d N=%d K=%d (operands stored k-major like above; the planes do not care)\n", M, N, KL);
        hipLaunchKernelGGL(ref_tn<float>, dim3((N + 255) / 256, M), dim3(256), 0, 0, A, B, C, M, N, KL);
        hipMemcpy(c.data(), C, c.size() * 4, hipMemcpyDeviceToHost);
        errors("fp32 fmaf chain in k order (= fp32 MFMA)", c, ref);
        hipLaunchKernelGGL(split_planes, dim3((unsigned)(((size_t)(KL / 8) * M + 255) / 256)), dim3(256), 0, 0, A, PA, KL, M);
        hipLaunchKernelGGL(split_planes, dim3((unsigned)(((size_t)(KL / 8) * N + 255) / 256)), dim3(256), 0, 0, B, PB, KL, N);
#define RUNZ(BK, NS, TERMS, Z, TNWV)                                                                                          \
        {                                                                                                                     \
            const dim3 grid(N / 64 / TNWV, M / 64, Z);                                                                        \
            auto g = [&] { hipLaunchKernelGGL((gemm_tn_bf16x<BK, NS, TERMS, true, true, TNWV>), grid, dim3(256), 0, 0, PA, PB, P, M, N, KL); }; \
            auto r = [&] { hipLaunchKernelGGL(sum_partials, dim3((unsigned)(((size_t)M * N + 255) / 256)), dim3(256), 0, 0, P, C, (size_t)M * N, Z); }; \
            g(); r();                                                                                                         \
            hipMemcpy(c.data(), C, c.size() * 4, hipMemcpyDeviceToHost);                                                      \
            char what[96];                                                                                                    \
            snprintf(what, sizeof what, "bf16 x %d, BK %d, ring %d, split-K %d, %d tiles/wave", TERMS, BK, NS, Z, TNWV);      \
            errors(what, c, ref);                                                                                             \
            printf("        product %.1f us + sum of the %d partials %.1f us\n", time_us(g), Z, time_us(r));                  \
        }
        RUNZ(32, 3, 9, 8, 1) RUNZ(32, 3, 6, 8, 1) RUNZ(32, 3, 6, 4, 1) RUNZ(16, 3, 6, 8, 2) RUNZ(16, 3, 6, 16, 2) RUNZ(16, 3, 9, 8, 2) RUNZ(32, 3, 6, 2, 1)
        hipFree(A); hipFree(B); hipFree(C); hipFree(P); hipFree(D); hipFree(PA); hipFree(PB);
    }
    return 0;
}
