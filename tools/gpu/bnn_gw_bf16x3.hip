// bnn_gw_bf16x3.hip -- round 6 gate experiment, stand-alone harness around tools/gpu/bnn_gw_bf16x3.hpp.
//
// gate: the two batched 2048 x 256 x 2048 weight-gradient products gW = h^T delta of the 10 M-parameter BNN <= 20 us isolated
// (library: 32.2 us in the step) with max and rms error against fp64 <= the library's on the same inputs; the bf16 planes are
// written beforehand (in the product they come from the epilogues of the launches that produce h and delta).
//
// Build + run (GPU box): make -C tools/gpu bnn_gw_bf16x3 && tools/gpu/bnn_gw_bf16x3
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#pragma clang fp contract(off)

#include "bnn_gw_bf16x3.hpp"

using namespace sgmcmc_gw;

namespace {

__global__ void fill(float *x, size_t n, unsigned seed, float scale, int mode)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed;
        float s = 0.f;
        for (int r = 0; r < 4; ++r) {
            h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
            s += (float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f;
        }
        // mode 0: bell-shaped * scale; 1: tanh outputs; 2: deltas with a per-row scale spread over 6 decades
        if (mode == 1) x[i] = tanhf(3.f * s);
        else if (mode == 2) x[i] = s * scale * exp10f(-6.f * (float)((h >> 3) & 1023) / 1023.f);
        else x[i] = s * scale;
    }
}

// C[i][j] = sum_m A[m][i] B[m][j] in fp64
__global__ void ref_f64(const float *__restrict__ A, const float *__restrict__ B, double *__restrict__ C, int M, int nA, int nB)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= nB) return;
    double acc = 0;
    for (int m = 0; m < M; ++m) acc = fma((double)A[(size_t)m * nA + i], (double)B[(size_t)m * nB + j], acc);
    C[(size_t)i * nB + j] = acc;
}

template <typename F>
float time_us(F f, int reps = 100)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 10; ++i) f();
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
    printf("    %-52s max|err| %.3e   rms err / rms value %.3e\n", what, maxabs, sqrt(sumsq / refsq));
}

template <int NS, bool TWO, int PROBE = 0, bool S16 = false>
void launch(const GwArgs &g, int batch)
{
    hipLaunchKernelGGL((gw_bf16x3_kernel<NS, TWO, PROBE, S16>), dim3(batch * g.tiles_i * g.tiles_j), dim3(256), 0, 0, g);
}

template <bool TWO, int PROBE = 0>
void launch_p(const GwArgs &g, int grid)
{
    hipLaunchKernelGGL((gw_bf16x3_persistent_kernel<TWO, PROBE>), dim3(grid), dim3(512), 0, 0, g);
}

template <int WM, bool TWO>
void launch_w(const GwArgs &g, int batch)
{
    const int rows = 64 * WM, tiles_i = (g.nA + rows - 1) / rows;
    hipLaunchKernelGGL((gw_bf16x3_wide_kernel<WM, TWO>), dim3(batch * tiles_i * g.tiles_j), dim3(128 * WM), 0, 0, g);
}

}  // namespace

int main(int argc, char **argv)
{
    rocblas_handle hb;
    rocblas_create_handle(&hb);
    const int M = 256;
    struct Shape { int nA, nB, batch; } shapes[] = {{2048, 2048, 2}, {2048, 2048, 1}, {785, 2048, 1}, {4864, 4864, 2}, {200, 72, 1}};
    const int only = argc > 1 ? atoi(argv[1]) : -1;                 // one shape only (under rocprofv3: kernel stats per shape)
    for (const Shape &s : shapes) {
        if (only >= 0 && &s != &shapes[only]) continue;
        const int nA = s.nA, nB = s.nB, batch = s.batch;
        const size_t ea = (size_t)M * nA, eb = (size_t)M * nB, ec = (size_t)nA * nB;
        float *A, *B, *C, *Cl;
        double *D;
        void *PA, *PB;
        hipMalloc(&A, ea * 4 * batch); hipMalloc(&B, eb * 4 * batch); hipMalloc(&C, ec * 4 * batch); hipMalloc(&Cl, ec * 4 * batch);
        hipMalloc(&D, ec * 8);
        const unsigned pa = (unsigned)(ea * 2), pb = (unsigned)(eb * 2);
        hipMalloc(&PA, (size_t)pa * 3 * batch); hipMalloc(&PB, (size_t)pb * 3 * batch);
        printf("gW [%d x %d] = h^T delta over a batch of %d, %d product(s) per launch\n", nA, nB, M, batch);
        GwArgs g{};
        for (int z = 0; z < batch; ++z) {
            hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, A + z * ea, ea, 3u + z, 2.0f, 1);
            hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, B + z * eb, eb, 7u + z, 0.05f, 2);
            hipLaunchKernelGGL(split_planes_kernel, dim3((nA + 255) / 256, M / 8), dim3(256), 0, 0, A + z * ea, M, nA, nA,
                               static_cast<unsigned char *>(PA) + (size_t)z * 3 * pa, pa);
            hipLaunchKernelGGL(split_planes_kernel, dim3((nB + 255) / 256, M / 8), dim3(256), 0, 0, B + z * eb, M, nB, nB,
                               static_cast<unsigned char *>(PB) + (size_t)z * 3 * pb, pb);
            g.A[z] = static_cast<unsigned char *>(PA) + (size_t)z * 3 * pa;
            g.B[z] = static_cast<unsigned char *>(PB) + (size_t)z * 3 * pb;
            g.C[z] = C + z * ec;
        }
        g.nA = nA; g.nB = nB; g.M = M; g.ldc = nB; g.plane_a_bytes = pa; g.plane_b_bytes = pb;
        g.tiles_i = (nA + TILE - 1) / TILE; g.tiles_j = (nB + TILE - 1) / TILE; g.batch = batch;
        const int T = batch * g.tiles_i * g.tiles_j, grid = T < 256 ? T : 256;
        const float one = 1.f, zero = 0.f;
        auto lib = [&] {
            // row-major C = A^T B  <=>  column-major C(j, i) = B_cm(j, m) A_cm(i, m)^T
            rocblas_sgemm_strided_batched(hb, rocblas_operation_none, rocblas_operation_transpose, nB, nA, M, &one, B, nB, (rocblas_stride)eb,
                                          A, nA, (rocblas_stride)ea, &zero, Cl, nB, (rocblas_stride)ec, batch);
        };
        lib();
        hipMemset(C, 0xff, ec * 4 * batch);
        launch<3, true>(g, batch);
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { printf("    kernel failed: %s\n", hipGetErrorString(e)); return 1; }
        std::vector<float> c(ec), cl(ec), c1(ec);
        std::vector<double> ref(ec);
        for (int z = 0; z < batch; ++z) {
            hipLaunchKernelGGL(ref_f64, dim3((nB + 255) / 256, nA), dim3(256), 0, 0, A + z * ea, B + z * eb, D, M, nA, nB);
            hipMemcpy(ref.data(), D, ec * 8, hipMemcpyDeviceToHost);
            hipMemcpy(c.data(), C + z * ec, ec * 4, hipMemcpyDeviceToHost);
            hipMemcpy(cl.data(), Cl + z * ec, ec * 4, hipMemcpyDeviceToHost);
            printf("  product %d\n", z);
            errors("library (rocblas_sgemm_strided_batched)", cl, ref);
            errors("bf16 x 3 planes, 6 products, two accumulator sets", c, ref);
        }
        hipMemset(C, 0xff, ec * 4 * batch);
        launch<3, false>(g, batch);
        hipDeviceSynchronize();
        hipMemcpy(c1.data(), C + (batch - 1) * ec, ec * 4, hipMemcpyDeviceToHost);
        errors("bf16 x 3 planes, 6 products, one accumulator set", c1, ref);
        if (nB % 4 == 0) {
            hipMemset(C, 0xff, ec * 4 * batch);
            launch<3, true, 0, true>(g, batch);
            hipDeviceSynchronize();
            hipMemcpy(c1.data(), C + (batch - 1) * ec, ec * 4, hipMemcpyDeviceToHost);
            errors("... two sets, deltas as the row operand (16-byte stores)", c1, ref);
            printf("    16-byte stores: NS=3 two acc %.2f us | NS=2 two acc %.2f | NS=3 one acc %.2f\n", time_us([&] { launch<3, true, 0, true>(g, batch); }),
                   time_us([&] { launch<2, true, 0, true>(g, batch); }), time_us([&] { launch<3, false, 0, true>(g, batch); }));
        }
        {
            hipMemset(C, 0xff, ec * 4 * batch);
            launch_p<true>(g, grid);
            hipError_t e2 = hipDeviceSynchronize();
            if (e2 != hipSuccess) { printf("    persistent kernel failed: %s\n", hipGetErrorString(e2)); return 1; }
            for (int z = 0; z < batch; ++z) {
                if (z != batch - 1) {
                    hipLaunchKernelGGL(ref_f64, dim3((nB + 255) / 256, nA), dim3(256), 0, 0, A + z * ea, B + z * eb, D, M, nA, nB);
                    hipMemcpy(ref.data(), D, ec * 8, hipMemcpyDeviceToHost);
                }
            }
            hipLaunchKernelGGL(ref_f64, dim3((nB + 255) / 256, nA), dim3(256), 0, 0, A + (batch - 1) * ea, B + (batch - 1) * eb, D, M, nA, nB);
            hipMemcpy(ref.data(), D, ec * 8, hipMemcpyDeviceToHost);
            hipMemcpy(c1.data(), C + (batch - 1) * ec, ec * 4, hipMemcpyDeviceToHost);
            errors("PERSISTENT, two sets (last product)", c1, ref);
            size_t d2 = 0;
            for (size_t i = 0; i < ec; ++i) d2 += (c1[i] != c[i]);
            hipMemcpy(c1.data(), C, ec * 4, hipMemcpyDeviceToHost);
            size_t nan0 = 0;
            for (size_t i = 0; i < ec; ++i) nan0 += (c1[i] != c1[i]);
            printf("    persistent vs one-tile-per-workgroup kernel: %zu elements differ; product 0 holds %zu NaN\n", d2, nan0);
            printf("    PERSISTENT (grid %d): two acc %.2f us | one acc %.2f | grid/2 %.2f | probes: no MFMA %.2f | no loads in the loop %.2f | no stores %.2f | no loads, no stores %.2f | ring only %.2f | MFMAs only %.2f | MFMAs + loads %.2f\n", grid,
                   time_us([&] { launch_p<true>(g, grid); }), time_us([&] { launch_p<false>(g, grid); }), time_us([&] { launch_p<true>(g, grid / 2 ? grid / 2 : 1); }),
                   time_us([&] { launch_p<true, 1>(g, grid); }), time_us([&] { launch_p<true, 2>(g, grid); }), time_us([&] { launch_p<true, 4>(g, grid); }),
                   time_us([&] { launch_p<true, 6>(g, grid); }), time_us([&] { launch_p<true, 7>(g, grid); }),
                   time_us([&] { launch_p<true, 14>(g, grid); }), time_us([&] { launch_p<true, 12>(g, grid); }));
        }
        {
            hipMemset(C, 0xff, ec * 4 * batch);
            launch_w<4, true>(g, batch);
            hipError_t e3 = hipDeviceSynchronize();
            if (e3 != hipSuccess) { printf("    wide kernel failed: %s\n", hipGetErrorString(e3)); return 1; }
            hipMemcpy(c1.data(), C + (batch - 1) * ec, ec * 4, hipMemcpyDeviceToHost);
            size_t d3 = 0;
            for (size_t i = 0; i < ec; ++i) d3 += (c1[i] != c[i]);
            hipMemset(C, 0xff, ec * 4 * batch);
            launch_w<6, true>(g, batch);
            hipDeviceSynchronize();
            hipMemcpy(c1.data(), C + (batch - 1) * ec, ec * 4, hipMemcpyDeviceToHost);
            size_t d4 = 0;
            for (size_t i = 0; i < ec; ++i) d4 += (c1[i] != c[i]);
            printf("    WIDE tiles: 256 x 128 differs from 128 x 128 in %zu elements, 384 x 128 in %zu\n", d3, d4);
            printf("    WIDE tiles, us per launch: 128x128 two acc %.2f | 256x128 two acc %.2f | 256x128 one acc %.2f | 384x128 two acc %.2f | 384x128 one acc %.2f\n",
                   time_us([&] { launch_w<2, true>(g, batch); }), time_us([&] { launch_w<4, true>(g, batch); }), time_us([&] { launch_w<4, false>(g, batch); }),
                   time_us([&] { launch_w<6, true>(g, batch); }), time_us([&] { launch_w<6, false>(g, batch); }));
        }
        printf("    nontemporal stores of the gradient: NS=2 two acc %.2f us (plain %.2f) | NS=3 two acc %.2f (plain %.2f)\n",
               time_us([&] { hipLaunchKernelGGL((gw_bf16x3_kernel<2, true, 0, false, true>), dim3(batch * g.tiles_i * g.tiles_j), dim3(256), 0, 0, g); }),
               time_us([&] { launch<2, true>(g, batch); }),
               time_us([&] { hipLaunchKernelGGL((gw_bf16x3_kernel<3, true, 0, false, true>), dim3(batch * g.tiles_i * g.tiles_j), dim3(256), 0, 0, g); }),
               time_us([&] { launch<3, true>(g, batch); }));
        // bit-reproducible?
        launch<3, true>(g, batch);
        hipDeviceSynchronize();
        hipMemcpy(c1.data(), C + (batch - 1) * ec, ec * 4, hipMemcpyDeviceToHost);
        size_t diff = 0;
        for (size_t i = 0; i < ec; ++i) diff += (c1[i] != c[i]);
        printf("    relaunch differs in %zu elements\n", diff);
        printf("    time per launch: library %.2f us | NS=3 two acc %.2f | NS=3 one acc %.2f | NS=2 two acc %.2f | NS=2 one acc %.2f\n",
               time_us(lib), time_us([&] { launch<3, true>(g, batch); }), time_us([&] { launch<3, false>(g, batch); }),
               time_us([&] { launch<2, true>(g, batch); }), time_us([&] { launch<2, false>(g, batch); }));
        if (nA == 2048 || nA == 4864)
            printf("    probes (NS=3, two acc): no MFMA %.2f | no loads in the loop %.2f | no stores %.2f | no MFMA, no stores %.2f | no loads, no stores %.2f | nothing but the ring %.2f\n",
                   time_us([&] { launch<3, true, 1>(g, batch); }), time_us([&] { launch<3, true, 2>(g, batch); }), time_us([&] { launch<3, true, 4>(g, batch); }),
                   time_us([&] { launch<3, true, 5>(g, batch); }), time_us([&] { launch<3, true, 6>(g, batch); }), time_us([&] { launch<3, true, 7>(g, batch); }));
        printf("    split launches (one per operand, %d x %d and %d x %d): %.2f us | %.2f us\n", M, nA, M, nB,
               time_us([&] { hipLaunchKernelGGL(split_planes_kernel, dim3((nA + 255) / 256, M / 8), dim3(256), 0, 0, A, M, nA, nA, PA, pa); }),
               time_us([&] { hipLaunchKernelGGL(split_planes_kernel, dim3((nB + 255) / 256, M / 8), dim3(256), 0, 0, B, M, nB, nB, PB, pb); }));
        hipFree(A); hipFree(B); hipFree(C); hipFree(Cl); hipFree(D); hipFree(PA); hipFree(PB);
    }
    (void)argc; (void)argv;
    return 0;
}
