// Common device helpers for the ViT-Lens gfx950 kernels (CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef uint16_t bf16_t;  // storage type for bf16 in global memory

#define VL_WAVE 64

// round-to-nearest-even fp32 -> bf16 on the gfx950 conversion unit (v_cvt_pk_bf16_f32: one instruction per PAIR,
// IEEE NaN handling) instead of ~5 integer ops per element.
typedef __attribute__((ext_vector_type(2))) float vl_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 vl_bf16x2;
__device__ __forceinline__ bf16_t f2bf(float f) {
  return __builtin_bit_cast(bf16_t, (__bf16)f);
}
__device__ __forceinline__ float bf2f(bf16_t h) {
  return __builtin_bit_cast(float, ((unsigned int)h) << 16);
}
__device__ __forceinline__ unsigned int pack2bf(float lo, float hi) {
  const vl_f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, vl_bf16x2));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7): far below the bf16 output
// resolution of the GELU epilogue and ~3x cheaper than erff().
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  p *= t;
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
  const float r = fmaf(-p, e, 1.0f);
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752f));
}
// d/dx gelu(x) = Phi(x) + x*phi(x)
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.0f + erf_as(x * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
  return fmaf(x, pdf, cdf);
}
