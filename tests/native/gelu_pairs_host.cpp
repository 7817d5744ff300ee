// Host build of the GELU helpers of vit-lens_amd/csrc/vl_common.h (tests/test_gelu_pairs_host.py): the packed-pair forms
// (gelu_erf2, gelu_erf_grad2, gelu_and_grad_pk) against the scalar forms they restate, on every finite bf16 value in both
// lanes.  The device builtins are replaced by their host meanings; what is compared is the structure of the arithmetic
// (which element goes where, signs, the step of gelu'), bit for bit.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#define __device__
#define __forceinline__ inline
static inline float vl_host_rcp(float x) { return 1.0f / x; }
static inline float vl_host_med3(float a, float b, float c) { return fminf(fmaxf(a, b), c); }
static inline float __shfl_xor(float v, int, int) { return v; }
#define __builtin_amdgcn_rcpf vl_host_rcp
#define __builtin_amdgcn_exp2f exp2f
#define __builtin_amdgcn_fmed3f vl_host_med3
#include VL_COMMON_HOST_H

static inline uint32_t bits(float f) { return __builtin_bit_cast(uint32_t, f); }

int main() {
  long bad = 0, n = 0;
  for (unsigned lo = 0; lo < 65536; ++lo) {
    const unsigned hi = (lo * 40503u + 12345u) & 0xffff;           // a different value, usually of the other sign, in the high lane
    if (((lo >> 7) & 0xff) == 0xff || ((hi >> 7) & 0xff) == 0xff) continue;     // infinities / NaNs
    const unsigned w = lo | (hi << 16);
    const float a = bf2f((bf16_t)lo), b = bf2f((bf16_t)hi);
    const vl_f32x2 x = {a, b};
    const vl_f32x2 y2 = gelu_erf2(x), g2 = gelu_erf_grad2(x);
    unsigned y, d;
    gelu_and_grad_pk(w, y, d);
    const int ok = bits(y2[0]) == bits(gelu_erf(a)) && bits(y2[1]) == bits(gelu_erf(b)) &&
                   bits(g2[0]) == bits(gelu_erf_grad(a)) && bits(g2[1]) == bits(gelu_erf_grad(b)) &&
                   y == pack2bf(gelu_erf(a), gelu_erf(b)) && d == pack2bf(gelu_erf_grad(a), gelu_erf_grad(b)) &&
                   pack2bf(x) == w && bits(unpack2bf(w)[0]) == bits(a) && bits(unpack2bf(w)[1]) == bits(b);
    if (!ok && bad++ < 5) printf("mismatch at w=%08x\n", w);
    ++n;
  }
  printf("checked=%ld bad=%ld\n", n, bad);
  return bad != 0;
}
