// Round 4 probe: does v_dot2c_f32_bf16 (__builtin_amdgcn_fdot2_f32_bf16) give the plain fp32 sum of products?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
typedef __attribute__((ext_vector_type(2))) __bf16 v2bf;
__global__ void k(const unsigned* in, float* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned w = in[i];
  v2bf a = __builtin_bit_cast(v2bf, w);
  v2bf one = __builtin_bit_cast(v2bf, 0x3f803f80u);
  float s1 = __builtin_amdgcn_fdot2_f32_bf16(a, one, 0.5f, false);
  float s2 = __builtin_amdgcn_fdot2_f32_bf16(a, a, 0.25f, false);
  out[2 * i] = s1; out[2 * i + 1] = s2;
}
int main() {
  const int n = 1 << 16;
  unsigned* h = (unsigned*)malloc(n * 4); float* o = (float*)malloc(n * 8);
  srand(1);
  for (int i = 0; i < n; ++i) {
    float a = (rand() / (float)RAND_MAX - 0.5f) * 80.f, b = (rand() / (float)RAND_MAX - 0.5f) * 3.f;
    unsigned ua, ub; memcpy(&ua, &a, 4); memcpy(&ub, &b, 4);
    h[i] = (ua >> 16) | (ub & 0xffff0000u);
  }
  unsigned* d; float* dd; hipMalloc(&d, n * 4); hipMalloc(&dd, n * 8);
  hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, dd, n);
  hipMemcpy(o, dd, n * 8, hipMemcpyDeviceToHost);
  double e1 = 0, e2 = 0;
  for (int i = 0; i < n; ++i) {
    unsigned lo = h[i] << 16, hi = h[i] & 0xffff0000u; float a, b; memcpy(&a, &lo, 4); memcpy(&b, &hi, 4);
    e1 = fmax(e1, fabs((double)o[2 * i] - ((double)a + b + 0.5)) / (fabs(a) + fabs(b) + 1));
    e2 = fmax(e2, fabs((double)o[2 * i + 1] - ((double)a * a + (double)b * b + 0.25)) / (a * a + b * b + 1));
  }
  printf("dot2 bf16: max rel err sum %.3e, sum of squares %.3e (first: %g %g)\n", e1, e2, o[0], o[1]);
  return 0;
}
