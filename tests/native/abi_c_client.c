/* A C client of the C ABI, no Python and no torch anywhere: what a maintainer binding include/vitlens_hip.h from another
 * host language links against.  Allocates device memory with the HIP runtime, runs the persistent bf16 GEMM + GELU
 * (vl_gemm_bf16, replaces mlp.c_fc + nn.GELU of ResidualAttentionBlock, open_clip/transformer.py:226-234), a LayerNorm
 * (vl_layernorm_fwd, transformer.py:17-34) and the fp32 GEMM (vl_gemm_f32), checks them against plain C loops on the host,
 * and provokes one argument error to show the status / vl_last_error protocol.  Built and run by tests/test_hip_abi_c_client.py
 * (gcc, on the GPU box).  Exit code 0 = all checks passed. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vitlens_hip.h"

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "hip error %s at line %d\n", hipGetErrorString(_e), __LINE__); return 2; } } while (0)

static uint16_t f2bf(float f) {                 /* round to nearest even */
  uint32_t u; memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static float frand(uint32_t* s) { *s = *s * 1664525u + 1013904223u; return ((*s >> 8) & 0xffff) / 32768.0f - 1.0f; }

int main(void) {
  if (vl_version() != VL_ABI_VERSION) {            /* built against another revision of the header than the library */
    printf("FAIL abi version: library %d, header %d\n", vl_version(), VL_ABI_VERSION);
    return 2;
  }
  const int M = 512, N = 768, K = 512;
  uint32_t seed = 12345u;
  uint16_t* a = malloc(sizeof(uint16_t) * M * K); uint16_t* w = malloc(sizeof(uint16_t) * N * K);
  float* bias = malloc(sizeof(float) * N); uint16_t* out = malloc(sizeof(uint16_t) * M * N);
  for (int i = 0; i < M * K; ++i) a[i] = f2bf(frand(&seed));
  for (int i = 0; i < N * K; ++i) w[i] = f2bf(frand(&seed) * 0.05f);
  for (int i = 0; i < N; ++i) bias[i] = frand(&seed);
  void *da, *dw, *db, *dout;
  CK(hipMalloc(&da, sizeof(uint16_t) * M * K)); CK(hipMalloc(&dw, sizeof(uint16_t) * N * K));
  CK(hipMalloc(&db, sizeof(float) * N)); CK(hipMalloc(&dout, sizeof(uint16_t) * M * N));
  CK(hipMemcpy(da, a, sizeof(uint16_t) * M * K, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, w, sizeof(uint16_t) * N * K, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, bias, sizeof(float) * N, hipMemcpyHostToDevice));
  hipStream_t s; CK(hipStreamCreate(&s));
  int fails = 0;

  /* 1. out = gelu(A W^T + bias), bf16, persistent 256x256 kernel by explicit request and through the auto dispatch */
  for (int pass = 0; pass < 2; ++pass) {
    const int cfg = pass ? VL_GEMM_AUTO : VL_GEMM_PERSIST;
    if (vl_gemm_bf16(da, dw, db, dout, NULL, M, N, K, K, K, N, 1.0f, VL_EPI_BF16, VL_ACT_GELU, cfg, s)) { fprintf(stderr, "vl_gemm_bf16: %s\n", vl_last_error()); return 3; }
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(out, dout, sizeof(uint16_t) * M * N, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int m = 0; m < M; m += 37)
      for (int n = 0; n < N; n += 11) {
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc += bf2f(a[m * K + k]) * bf2f(w[n * K + k]);
        const float pre = acc + bias[n];
        const float ref = 0.5f * pre * (1.0f + erff(pre * 0.70710678f));
        const double e = fabs(bf2f(out[m * N + n]) - ref) / (fabs(ref) * 0.0079 + 2e-3);      /* one bf16 ulp + an absolute floor */
        if (e > worst) worst = e;
      }
    printf("vl_gemm_bf16 (cfg %d) + GELU vs host loop: worst error %.3f of the bound\n", cfg, worst);
    fails += worst > 1.0;
  }

  /* 2. LayerNorm f32 -> f32 over rows of 768 */
  {
    const int rows = 64, D = 768;
    float* x = malloc(sizeof(float) * rows * D); float* g = malloc(sizeof(float) * D); float* bt = malloc(sizeof(float) * D); float* y = malloc(sizeof(float) * rows * D);
    for (int i = 0; i < rows * D; ++i) x[i] = 3.0f * frand(&seed) + 0.5f;
    for (int i = 0; i < D; ++i) { g[i] = 1.0f + 0.2f * frand(&seed); bt[i] = 0.1f * frand(&seed); }
    void *dx, *dg, *dbt, *dy;
    CK(hipMalloc(&dx, sizeof(float) * rows * D)); CK(hipMalloc(&dg, sizeof(float) * D)); CK(hipMalloc(&dbt, sizeof(float) * D)); CK(hipMalloc(&dy, sizeof(float) * rows * D));
    CK(hipMemcpy(dx, x, sizeof(float) * rows * D, hipMemcpyHostToDevice)); CK(hipMemcpy(dg, g, sizeof(float) * D, hipMemcpyHostToDevice)); CK(hipMemcpy(dbt, bt, sizeof(float) * D, hipMemcpyHostToDevice));
    if (vl_layernorm_fwd(dx, VL_F32, D, NULL, 0, dg, dbt, dy, VL_F32, D, NULL, NULL, rows, D, 1e-5f, s)) { fprintf(stderr, "vl_layernorm_fwd: %s\n", vl_last_error()); return 3; }
    CK(hipStreamSynchronize(s)); CK(hipMemcpy(y, dy, sizeof(float) * rows * D, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int r = 0; r < rows; ++r) {
      double mu = 0, var = 0;
      for (int d = 0; d < D; ++d) mu += x[r * D + d];
      mu /= D;
      for (int d = 0; d < D; ++d) var += (x[r * D + d] - mu) * (x[r * D + d] - mu);
      var /= D;
      for (int d = 0; d < D; ++d) {
        const double ref = (x[r * D + d] - mu) / sqrt(var + 1e-5) * g[d] + bt[d];
        const double e = fabs(y[r * D + d] - ref);
        if (e > worst) worst = e;
      }
    }
    printf("vl_layernorm_fwd vs host loop: worst abs error %.2e\n", worst);
    fails += worst > 2e-5;
  }

  /* 3. fp32 GEMM on the fp32-input MFMA, ragged shape */
  {
    const int m2 = 70, n2 = 36, k2 = 128;
    float* fa = malloc(sizeof(float) * m2 * k2); float* fw = malloc(sizeof(float) * n2 * k2); float* fo = malloc(sizeof(float) * m2 * n2);
    for (int i = 0; i < m2 * k2; ++i) fa[i] = frand(&seed);
    for (int i = 0; i < n2 * k2; ++i) fw[i] = frand(&seed);
    void *dfa, *dfw, *dfo;
    CK(hipMalloc(&dfa, sizeof(float) * m2 * k2)); CK(hipMalloc(&dfw, sizeof(float) * n2 * k2)); CK(hipMalloc(&dfo, sizeof(float) * m2 * n2));
    CK(hipMemcpy(dfa, fa, sizeof(float) * m2 * k2, hipMemcpyHostToDevice)); CK(hipMemcpy(dfw, fw, sizeof(float) * n2 * k2, hipMemcpyHostToDevice));
    if (vl_gemm_f32(dfa, dfw, NULL, dfo, NULL, m2, n2, k2, k2, k2, n2, 1.0f, VL_ACT_NONE, s)) { fprintf(stderr, "vl_gemm_f32: %s\n", vl_last_error()); return 3; }
    CK(hipStreamSynchronize(s)); CK(hipMemcpy(fo, dfo, sizeof(float) * m2 * n2, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int m = 0; m < m2; ++m)
      for (int n = 0; n < n2; ++n) {
        double acc = 0;
        for (int k = 0; k < k2; ++k) acc += (double)fa[m * k2 + k] * fw[n * k2 + k];
        const double e = fabs(fo[m * n2 + n] - acc);
        if (e > worst) worst = e;
      }
    printf("vl_gemm_f32 vs host double loop: worst abs error %.2e\n", worst);
    fails += worst > 2e-5;
  }

  /* 4. errors are status codes + a message, never exceptions: K not a multiple of 64 */
  {
    const int rc = vl_gemm_bf16(da, dw, db, dout, NULL, M, N, 100, K, K, N, 1.0f, VL_EPI_BF16, VL_ACT_NONE, VL_GEMM_AUTO, s);
    printf("bad argument -> status %d, message: %s\n", rc, vl_last_error());
    fails += rc == 0;
  }
  printf(fails ? "FAILED (%d)\n" : "ALL OK\n", fails);
  return fails ? 1 : 0;
}
