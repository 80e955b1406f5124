/* TEST INFRASTRUCTURE ONLY: exercises every entry point of the C oracle under AddressSanitizer + UndefinedBehaviorSanitizer
 * ("race detection / sanitizers": the GPU sanitizers are not available on this pool, the host
 * restatement is checked instead). Built and run by `make -C oracle sanitize` (tests/test_oracle_pins.py).          */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

int oracle_sghmc_step_f32(float *, float *, const float *, float *, float *, float *, float *, float *, size_t, float, float,
                          float, float, int, const float *, uint64_t, uint64_t);
int oracle_sghmc_step_f64(double *, double *, const double *, double *, double *, double *, double *, double *, size_t, double,
                          double, double, double, int, const double *, uint64_t, uint64_t);
int oracle_sgld_step_f32(float *, const float *, float *, float *, float *, float *, float *, size_t, float, float, float, float,
                         int, const float *, uint64_t, uint64_t);
int oracle_rsghmc_step_f32(float *, float *, const float *, size_t, float, float, float, float, float, float, const float *,
                           uint64_t, uint64_t);
int oracle_moments_update_f32(const float *, float *, float *, size_t, uint64_t);
int oracle_rhat_pack_f32(const float *, const float *, size_t, uint64_t, size_t, size_t, float *);
int oracle_rhat_finish_f32(const float *, size_t, size_t, int, uint64_t, float *);
int oracle_rsghmc_toy_chain_f32(int, const float *, int, float *, float *, int, float, float, float, float, float, uint64_t,
                                uint64_t, uint64_t, uint64_t, float *);
void oracle_philox_normal_f32(uint64_t, uint64_t, size_t, float *);
void oracle_philox_uniform_bits(uint64_t, uint64_t, size_t, uint32_t *);

#define N 1003   /* ragged on purpose */

int main(void)
{
    float *th = malloc(N * sizeof(float)), *V = calloc(N, sizeof(float)), *gr = malloc(N * sizeof(float));
    float *tau = malloc(N * sizeof(float)), *g = malloc(N * sizeof(float)), *vh = malloc(N * sizeof(float));
    float *mi = malloc(N * sizeof(float)), *r = malloc(N * sizeof(float)), *xi = malloc(N * sizeof(float));
    float *mean = calloc(N, sizeof(float)), *m2 = calloc(N, sizeof(float));
    double *thd = malloc(N * sizeof(double)), *Vd = calloc(N, sizeof(double)), *grd = malloc(N * sizeof(double));
    double *taud = malloc(N * sizeof(double)), *gd = malloc(N * sizeof(double)), *vhd = malloc(N * sizeof(double));
    double *mid = malloc(N * sizeof(double));
    uint32_t *bits = malloc(N * sizeof(uint32_t));
    int i, t, bad = 0;
    for (i = 0; i < N; ++i) {
        th[i] = 0.01f * (float)(i % 17 - 8); gr[i] = 0.3f * (float)(i % 5 - 2); tau[i] = g[i] = vh[i] = mi[i] = 1.0f; r[i] = 0.5f;
        thd[i] = th[i]; grd[i] = gr[i]; taud[i] = gd[i] = vhd[i] = mid[i] = 1.0;
    }
    oracle_philox_uniform_bits(7, 3, N, bits);
    for (t = 0; t < 6; ++t) {
        oracle_philox_normal_f32(11, (uint64_t)t, N, xi);
        bad |= oracle_sghmc_step_f32(th, V, gr, tau, g, vh, mi, r, N, 0.01f, 100.0f, 0.05f, 1e-4f, t < 3, t & 1 ? xi : NULL, 5, (uint64_t)t);
        bad |= oracle_sghmc_step_f64(thd, Vd, grd, taud, gd, vhd, mid, NULL, N, 0.01, 100.0, 0.05, 0.0, t < 3, NULL, 5, (uint64_t)t);
        bad |= oracle_sgld_step_f32(th, gr, tau, g, vh, mi, NULL, N, 0.01f, 1.0f, 100.0f, 0.0f, t < 3, xi, 5, (uint64_t)t);
        bad |= oracle_rsghmc_step_f32(th, V, gr, N, 0.001f, 1.0f, 1.0f, 1.0f, 0.0f, 0.0f, NULL, 5, (uint64_t)t);
        bad |= oracle_moments_update_f32(th, mean, m2, N, (uint64_t)t + 1);
    }
    {
        size_t L = 252, S = 4;                                   /* 4 shards x 252 >= 1003, none empty */
        float *pack = malloc(3 * S * L * sizeof(float)), *rhat = malloc(L * sizeof(float));
        bad |= oracle_rhat_pack_f32(mean, m2, N, 6, S, L, pack);
        for (i = 0; i < (int)(3 * S * L); ++i) pack[i] *= 2.0f;  /* "two identical chains" */
        bad |= oracle_rhat_finish_f32(pack + 3 * 3 * L, N - 3 * L, L, 2, 6, rhat);   /* the last, ragged shard */
        free(pack); free(rhat);
    }
    {
        float tp[9] = {-5, 0, 5, 2, 0.5f, 2, 1.0f / 3, 1.0f / 3, 1.0f / 3}, x[1] = {0}, p[1] = {0.3f}, kept[100];
        float xy[2] = {0, 6}, pp[2] = {0.1f, -0.2f}, kept2[200];
        bad |= oracle_rsghmc_toy_chain_f32(0, tp, 3, x, p, 1, 0.5f, 1, 1, 1, 0, 3, 0, 991, 10, kept);
        bad |= oracle_rsghmc_toy_chain_f32(1, tp, 0, xy, pp, 2, 0.5f, 1, 1, 1, 0, 3, 0, 991, 10, kept2);
        if (!isfinite(kept[99]) || !isfinite(kept2[199])) bad |= 2;
    }
    for (i = 0; i < N; ++i) if (!isfinite(th[i]) || !isfinite(thd[i])) bad |= 4;
    free(th); free(V); free(gr); free(tau); free(g); free(vh); free(mi); free(r); free(xi); free(mean); free(m2);
    free(thd); free(Vd); free(grd); free(taud); free(gd); free(vhd); free(mid); free(bits);
    printf(bad ? "sanitize: FAILED (%d)\n" : "sanitize: ok\n", bad);
    return bad;
}
