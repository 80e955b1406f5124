/* sgmcmc_gemm_experiments.h -- entry points of the round-3 GEMM experiments (NOT part of the product ABI,
 * include/sgmcmc_hip.h): a hand-written fp32 matrix-core weight-gradient product with tile-shape / timing-probe variants and
 * its fusion with the frozen SGHMC update. Built by tools/experiments/Makefile into libsgmcmc_hip_experiments.so (the product
 * objects + csrc/sgmcmc_gemm.hip), loaded through PYSGMCMC_AMD_LIB by tools/experiments/gemm_kernels.py.
 * Measurements: DESIGN.md section 3, profiles/r03_gemm_fusion_probe.txt. */
#ifndef SGMCMC_GEMM_EXPERIMENTS_H
#define SGMCMC_GEMM_EXPERIMENTS_H
#include "sgmcmc_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Weight-gradient product of a dense layer on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact f32):
 * C[M][N] = A^T B, A = the layer's input activations [K = batch][lda >= M], B = its deltas [K][ldb >= N]
 * (replaces the tf.gradients matmul behind pysgmcmc/samplers/sghmc.py:121-122 for one kernel matrix).
 * N % 128 == 0, K % 16 == 0, M % 4 == 0, operands and C 16-byte aligned, lda/ldb/ldc % 4 == 0.
 * variant 0: 64 x 64 tiles, operands loaded from global memory DIRECTLY into LDS (global_load_lds_dwordx4), 3-deep ring;
 * variants 1-11: other tile shapes / register-staged operand loads (tools/gemm_probe2.py); low byte = variant, bits 8-11
 * = timing probes (skip the store, a quarter of K, ...).                                                         */
int sgmcmc_gemm_tn_f32(const float *A, const float *B, float *C, int M, int N, int K, int lda, int ldb, int ldc,
                       int variant /* tile shape, 0 = default */, int *phase_counters /* NULL, or 2048 zeroed ints */,
                       int phase_sleep, sgmcmc_stream_t stream);

/* The same product with the frozen SGHMC update of the layer as its epilogue (kernel K1's arithmetic and Philox stream,
 * sghmc.py:211-251 with fed minv): the tile of gW a workgroup accumulated never goes to HBM; the workgroup updates the
 * same tile of theta / V (the layer's weights W = theta[0 .. M N), row-major [M][N]) and the launch also updates the
 * n_tail parameters that follow W in the arena (bias, ...) from their already computed gradient grad_tail.
 * 20 B of HBM traffic per weight instead of 4 (GEMM output) + 24 (K1), hidden under the matrix-core work.
 *   first_element: index of W[0][0] in the chain's parameter vector (Philox counter of element i = (first_element + i) / 4)
 *   grad_out:      NULL, or [M][N]: also write gW (tests: K1 on this gradient gives the same theta', V' bit for bit)
 *   stats_ws / stats_record_*: as sgmcmc_step_opts_t; one {sum theta'^2, 0, 0, 0} record per workgroup,
 *                  sgmcmc_gemm_tn_sghmc_blocks() = the launch's workgroup count.
 *   gemm_blocks:   bits 0-15: persistent workgroups of the product, 0 = default (1024 = 4 per CU); bits 16+: probe, when
 *                  the tile's theta/V/minv are requested (0 = after the K loop, the default; 1, 2 = partly / wholly before it:
 *                  no faster)
 *   K % 16 == 0, N % 128 == 0, M % 4 == 0.                                                                        */
int sgmcmc_gemm_tn_sghmc_f32(const float *A, const float *B, int M, int N, int K, int lda, int ldb, float *theta, float *V,
                             const float *minv, const float *grad_tail, size_t n_tail, float *grad_out, float eps,
                             float scale_grad, float mdecay, float grad_decay, uint64_t seed, uint64_t step,
                             const uint64_t *step_dev, uint64_t first_element, void *stats_ws, uint32_t stats_record_base,
                             uint32_t stats_record_total, int gemm_blocks, int *phase_counters /* NULL, or 2048 zeroed ints */,
                             int phase_sleeps, sgmcmc_stream_t stream);
int sgmcmc_gemm_tn_sghmc_blocks(int M, int N, size_t n_tail, int gemm_blocks);

#ifdef __cplusplus
}
#endif
#endif
