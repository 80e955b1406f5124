// valu_rate.hip -- issue rates of the vector instructions the bf16-split uses, alone and next to v_mfma_f32_32x32x16_bf16 (gfx950).
// One workgroup of W waves per CU (W = 4: one per SIMD, 8: two per SIMD), every wave runs REPS x 64 independent instructions of
// one kind; reported: SIMD cycles per wave-instruction at the clock the launch sustained (s_memtime ticks are shader cycles).
// Build: make -C tools/gpu valu_rate && tools/gpu/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int REPS = 256;

// KIND 0 v_and_b32, 1 v_perm_b32, 2 v_pk_add_f32, 3 v_sub_f32, 4 v_pk_mul_f32, 5 v_cvt_pk_bf16_f32, 6 v_lshlrev_b32, 7 v_fma_f32
// MF: number of bf16 MFMAs interleaved per 16 vector instructions (0 = none)
template <int KIND, int MF>
__global__ void __launch_bounds__(512) rate(unsigned *out, unsigned long long *ticks, unsigned seed)
{
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned x[16];
    float f[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { x[i] = threadIdx.x * 2654435761u + i + seed; f[i] = (float)(x[i] & 1023) * 1e-3f + 1.f; }
    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;
    u32x4 pa = {x[0], x[1], x[2], x[3]}, pb = {x[4], x[5], x[6], x[7]};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < REPS; ++rep) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (KIND == 0) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(x[i]));
                if (KIND == 1) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(x[(i + 1) & 15]), "s"(0x07060302u));
                if (KIND == 3) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 15]));
                if (KIND == 6) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(x[i]));
                if (KIND == 7) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(f[(i + 1) & 15]));
                if (KIND == 8) asm volatile("v_pack_b32_f16 %0, %0, %1 op_sel:[1,1,0]" : "+v"(x[i]) : "v"(x[(i + 1) & 15]));
                if (KIND == 9) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(f[i]) : "v"(x[(i + 1) & 15]), "v"(x[(i + 2) & 15]));
                if (KIND == 10) asm volatile("v_bfi_b32 %0, %2, %0, %1" : "+v"(x[i]) : "v"(x[(i + 1) & 15]), "s"(0xffff0000u));
                if (KIND == 11) asm volatile("v_and_or_b32 %0, %0, %2, %1" : "+v"(x[i]) : "v"(x[(i + 1) & 15]), "s"(0xffff0000u));
                if (KIND == 12) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(x[i]) : "v"(x[(i + 1) & 15]));
                if (KIND == 13) asm volatile("v_lshrrev_b32 %0, 16, %0" : "+v"(x[i]));
                if (KIND == 14) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "+v"(x[i]) : "v"(f[i]), "v"(f[(i + 1) & 15]));
            }
            if (KIND == 2 || KIND == 4 || KIND == 5) {
#pragma unroll
                for (int i = 0; i < 16; i += 2) {           // 16 packed instructions on 8 register pairs
                    f32x2 v = {f[i], f[i + 1]}, w = {f[(i + 2) & 15], f[(i + 3) & 15]};
                    if (KIND == 2) { asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(v) : "v"(w)); asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(v) : "v"(w)); }
                    if (KIND == 4) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v) : "v"(w)); asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v) : "v"(w)); }
                    if (KIND == 5) { unsigned d; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(v.x), "v"(v.y)); x[i] ^= d; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(w.x), "v"(w.y)); x[i + 1] ^= d; }
                    f[i] = v.x; f[i + 1] = v.y;
                }
            }
            if (MF > 0 && u < MF) {
                acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pa), __builtin_bit_cast(bf16x8, pb), acc[u & 1], 0, 0, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s ^= x[i] ^ __float_as_uint(f[i]);
#pragma unroll
    for (int r = 0; r < 16; ++r) s ^= __float_as_uint(acc[0][r] + acc[1][r]);
    if (s == 0x12345678u) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
#endif
}

template <int KIND, int MF>
void run(const char *what, int waves, unsigned *out, unsigned long long *ticks)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((rate<KIND, MF>), dim3(256), dim3(64 * waves), 0, 0, out, ticks, 1u);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((rate<KIND, MF>), dim3(256), dim3(64 * waves), 0, 0, out, ticks, 1u);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t;
    hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    const double n_instr = (double)REPS * 64.0, per_simd = waves / 4.0;
    // s_memtime counts at a fixed 100 MHz on this part: use wall time x an assumed 2.4 GHz as a second view
    printf("%-34s %d waves/SIMD, %d MFMA per 16: %7.1f us/launch  | %6.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz)  memtime ticks %llu\n",
           what, waves / 4, MF, ms / 10 * 1e3, ms / 10 * 1e6 / (n_instr * per_simd), ms / 10 * 1e6 / (n_instr * per_simd) * 2.4, t);
}

// exactness of the cheap split: planes by v_pack_b32_f16 (high halves) + remainders by v_dot2c_f32_bf16 against v_perm / v_and / v_sub
__global__ void split_check(const unsigned *in, unsigned *bad, unsigned n)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const unsigned xb = in[2 * i], yb = in[2 * i + 1];
    const float x = __uint_as_float(xb), y = __uint_as_float(yb);
    // reference
    const unsigned p0 = __builtin_amdgcn_perm(yb, xb, 0x07060302u);
    const float xr = x - __uint_as_float(xb & 0xffff0000u), yr = y - __uint_as_float(yb & 0xffff0000u);
    const unsigned p1 = __builtin_amdgcn_perm(__float_as_uint(yr), __float_as_uint(xr), 0x07060302u);
    const float xl = xr - __uint_as_float(__float_as_uint(xr) & 0xffff0000u), yl = yr - __uint_as_float(__float_as_uint(yr) & 0xffff0000u);
    const unsigned p2 = __builtin_amdgcn_perm(__float_as_uint(yl), __float_as_uint(xl), 0x07060302u);
    // cheap
    unsigned q0, q1, q2;
    float ax = x, ay = y;
    const unsigned mx = 0x0000bf80u, my = 0xbf800000u;     // (-1, 0), (0, -1) as bf16 pairs
    asm volatile("v_pack_b32_f16 %0, %1, %2 op_sel:[1,1,0]" : "=v"(q0) : "v"(xb), "v"(yb));
    asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(ax) : "v"(q0), "v"(mx));
    asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(ay) : "v"(q0), "v"(my));
    asm volatile("v_pack_b32_f16 %0, %1, %2 op_sel:[1,1,0]" : "=v"(q1) : "v"(ax), "v"(ay));
    float bx = ax, by = ay;
    asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(bx) : "v"(q1), "v"(mx));
    asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(by) : "v"(q1), "v"(my));
    asm volatile("v_pack_b32_f16 %0, %1, %2 op_sel:[1,1,0]" : "=v"(q2) : "v"(bx), "v"(by));
    if (q0 != p0) atomicAdd(&bad[0], 1u);
    if (q1 != p1) atomicAdd(&bad[1], 1u);
    if (q2 != p2) atomicAdd(&bad[2], 1u);
    if ((q0 != p0 || q1 != p1 || q2 != p2) && atomicAdd(&bad[3], 1u) < 8) {
        bad[4 + 8 * (bad[3] - 1 < 7 ? bad[3] - 1 : 7)] = xb;
    }
#endif
}

int main()
{
    {
        const unsigned n = 1u << 22;
        unsigned *in, *bad;
        hipMalloc(&in, n * 4); hipMalloc(&bad, 4 * 80);
        unsigned *h = new unsigned[n];
        unsigned s = 12345u;
        for (int mode = 0; mode < 3; ++mode) {
            for (unsigned i = 0; i < n; ++i) {
                s = s * 1664525u + 1013904223u;
                unsigned v = s ^ (s >> 13);
                if (mode == 1) v = (v & 0x807fffffu) | ((100u + (v >> 23) % 56u) << 23);       // moderate exponents 2^-27 .. 2^28
                if (mode == 2) v = (v & 0x807fffffu) | (((v >> 23) % 40u) << 23);             // tiny values and fp32 denormals
                h[i] = v;
            }
            hipMemcpy(in, h, n * 4, hipMemcpyHostToDevice);
            hipMemset(bad, 0, 4 * 80);
            hipLaunchKernelGGL(split_check, dim3(n / 2 / 256), dim3(256), 0, 0, in, bad, n);
            unsigned r[12];
            hipMemcpy(r, bad, 48, hipMemcpyDeviceToHost);
            printf("split check, %s: plane 0 / 1 / 2 mismatches of %u pairs: %u %u %u   (first inputs 0x%08x 0x%08x)\n",
                   mode == 0 ? "random bit patterns (NaN / Inf included)" : mode == 1 ? "moderate exponents" : "tiny values / denormals", n / 2, r[0], r[1], r[2], r[4], r[5]);
        }
        delete[] h;
        hipFree(in); hipFree(bad);
    }
    unsigned *out;
    unsigned long long *ticks;
    hipMalloc(&out, 4); hipMalloc(&ticks, 8);
    for (int waves : {4, 8}) {
        run<0, 0>("v_and_b32", waves, out, ticks);
        run<1, 0>("v_perm_b32", waves, out, ticks);
        run<2, 0>("v_pk_add_f32", waves, out, ticks);
        run<3, 0>("v_sub_f32", waves, out, ticks);
        run<4, 0>("v_pk_mul_f32", waves, out, ticks);
        run<5, 0>("v_cvt_pk_bf16_f32 (+ v_xor)", waves, out, ticks);
        run<6, 0>("v_lshlrev_b32", waves, out, ticks);
        run<7, 0>("v_fma_f32", waves, out, ticks);
        run<8, 0>("v_pack_b32_f16 op_sel hi,hi", waves, out, ticks);
        run<9, 0>("v_dot2c_f32_bf16", waves, out, ticks);
        run<10, 0>("v_bfi_b32", waves, out, ticks);
        run<11, 0>("v_and_or_b32", waves, out, ticks);
        run<12, 0>("v_alignbit_b32", waves, out, ticks);
        run<13, 0>("v_lshrrev_b32", waves, out, ticks);
        run<14, 0>("v_cvt_pk_bf16_f32 alone", waves, out, ticks);
        run<8, 2>("v_pack_b32_f16 op_sel hi,hi", waves, out, ticks);
        run<9, 2>("v_dot2c_f32_bf16", waves, out, ticks);
        run<0, 1>("v_and_b32", waves, out, ticks);
        run<0, 2>("v_and_b32", waves, out, ticks);
        run<0, 4>("v_and_b32", waves, out, ticks);
        run<1, 2>("v_perm_b32", waves, out, ticks);
        run<2, 2>("v_pk_add_f32", waves, out, ticks);
        run<3, 2>("v_sub_f32", waves, out, ticks);
    }
    // MFMA alone: 4 per 16 "instructions" of nothing -> KIND 99
    for (int waves : {4, 8}) run<99, 4>("(no vector work) MFMA only", waves, out, ticks);
    return 0;
}
