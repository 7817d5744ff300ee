// Semantics of gfx950's LDS transpose read (ds_read_b64_tr_b16) as the TN weight-gradient GEMM uses it: LDS holds the element
// index; every lane supplies the address of 4 consecutive elements; printed: what each lane receives.
// Build: hipcc --offload-arch=gfx950 -O3 tools/tr_probe.hip -o tools/bin/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(short* out, int mode) {
  __shared__ short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  // mode 0: lane-linear addresses (lane l -> elements 4l..4l+3).
  // mode 1: a [token][256 m] image: group g reads tokens g*8 + (i>>2), columns (i&3)*4..+3 of m-block 3 (row = 256 elements)
  int e = mode == 0 ? l * 4 : (g * 8 + (i >> 2)) * 256 + 3 * 16 + (i & 3) * 4;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + e));
  *(s16x4*)(out + l * 4) = v;
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  short h[256];
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) {
      if (mode == 0) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
      else printf("lane %2d: (t%2d,m%3d) (t%2d,m%3d) (t%2d,m%3d) (t%2d,m%3d)\n", l, h[l*4] / 256, h[l*4] % 256, h[l*4+1] / 256, h[l*4+1] % 256,
                  h[l*4+2] / 256, h[l*4+2] % 256, h[l*4+3] / 256, h[l*4+3] % 256);
    }
  }
  return 0;
}
