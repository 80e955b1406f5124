// Test probe: the first rocrand4() of rocRAND's own Philox4x32-10 engine (rocrand_kernel.h, ROCm) after
// rocrand_init(seed, subsequence, offset), evaluated on the host and on the device. tests/test_rocrand_layout.py compares these
// words with the library's stream to check the claim in include/sgmcmc_hip.h ("identical to rocRAND's philox4x32_10 stream
// with subsequence = quad, offset = 4 * step").
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_kernel.h>

__global__ void rocrand_words_kernel(unsigned long long seed, const unsigned long long *subseq, const unsigned long long *offset,
                                     int n, unsigned int *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    rocrand_state_philox4x32_10 st;
    rocrand_init(seed, subseq[i], offset[i], &st);
    const uint4 v = rocrand4(&st);
    out[4 * i + 0] = v.x; out[4 * i + 1] = v.y; out[4 * i + 2] = v.z; out[4 * i + 3] = v.w;
}

extern "C" {

// host evaluation of the same header (its engine is __host__ __device__): no GPU needed
void rocrand_words_host(unsigned long long seed, const unsigned long long *subseq, const unsigned long long *offset, int n,
                        unsigned int *out)
{
    for (int i = 0; i < n; ++i) {
        rocrand_state_philox4x32_10 st;
        rocrand_init(seed, subseq[i], offset[i], &st);
        const uint4 v = rocrand4(&st);
        out[4 * i + 0] = v.x; out[4 * i + 1] = v.y; out[4 * i + 2] = v.z; out[4 * i + 3] = v.w;
    }
}

// device evaluation; subseq / offset / out are HOST arrays. Returns 0 or a hipError_t.
int rocrand_words_device(unsigned long long seed, const unsigned long long *subseq, const unsigned long long *offset, int n,
                         unsigned int *out)
{
    unsigned long long *d_s = nullptr, *d_o = nullptr;
    unsigned int *d_out = nullptr;
    hipError_t e;
    if ((e = hipMalloc(&d_s, n * sizeof(*d_s))) != hipSuccess) return (int)e;
    if ((e = hipMalloc(&d_o, n * sizeof(*d_o))) != hipSuccess) return (int)e;
    if ((e = hipMalloc(&d_out, 4 * n * sizeof(*d_out))) != hipSuccess) return (int)e;
    hipMemcpy(d_s, subseq, n * sizeof(*d_s), hipMemcpyHostToDevice);
    hipMemcpy(d_o, offset, n * sizeof(*d_o), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(rocrand_words_kernel, dim3((n + 63) / 64), dim3(64), 0, 0, seed, d_s, d_o, n, d_out);
    e = hipMemcpy(out, d_out, 4 * n * sizeof(*d_out), hipMemcpyDeviceToHost);
    hipFree(d_s); hipFree(d_o); hipFree(d_out);
    return (int)e;
}

}  // extern "C"
