// build: /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o build/mfma_f32_probe tools/mfma_f32_probe.hip ; run: gpurun -- ./build/mfma_f32_probe
// Throughput probe: v_mfma_f32_32x32x2_f32 with 4 independent accumulator chains per wave, 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void probe(float *out, int iters, float a0, float b0) {
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float a = a0 + threadIdx.x, b = b0;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float *out;
    hipMalloc(&out, 1 << 24);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wpb : {256, 512, 1024}) {
        const int iters = 20000, blocks = 256 * 2;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(probe, dim3(blocks), dim3(wpb), 0, 0, out, iters, 1.0f, 1e-3f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double mfmas = (double)blocks * (wpb / 64) * iters * 4;
        printf("threads/block %4d: %.3f ms, %.1f TFLOP/s, %.1f cycles per MFMA per SIMD at 2.4 GHz\n", wpb, ms,
               mfmas * 4096 / ms * 1e-9, ms * 1e-3 * 2.4e9 / (mfmas / 1024));
    }
    return 0;
}
