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
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

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

// IEEE half (the frozen text tower's operands): round-to-nearest-even, SATURATING at the largest finite half - fp16 has five
// exponent bits and an activation outlier must not become an infinity (bf16 shares fp32's range and needs no clamp)
typedef __attribute__((ext_vector_type(2))) _Float16 vl_f16x2;
__device__ __forceinline__ unsigned int pack2h(float lo, float hi) {
  const vl_f32x2 v = {__builtin_amdgcn_fmed3f(lo, -65504.0f, 65504.0f), __builtin_amdgcn_fmed3f(hi, -65504.0f, 65504.0f)};
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, vl_f16x2));
}
__device__ __forceinline__ uint16_t f2h(float f) {
  return __builtin_bit_cast(uint16_t, (_Float16)__builtin_amdgcn_fmed3f(f, -65504.0f, 65504.0f));
}
__device__ __forceinline__ float h2f(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }

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

// GELU(x) = x*Phi(x) and its derivative from ONE shared evaluation of q(t)*exp(-x^2/2) (A&S 7.1.26 with the 1/2 folded
// into the coefficients, |abs err| <= 4e-7): 13 / 15 VALU instructions per element instead of 17 / 24 - the activation
// arithmetic is 15-30 % of a K = 1024 GEMM's epilogue-inclusive time.
//   x >= 0: Phi = 1 - h, x < 0: Phi = h, with h = q(t) e^{-x^2/2}, t = 1 / (1 + p|x|/sqrt 2)
//   gelu  = max(x, 0) - |x| h                gelu' = [x >= 0] + e^{-x^2/2} (x / sqrt(2 pi) - sign(x) q(t))
struct GeluParts { float q, e; };
__device__ __forceinline__ GeluParts gelu_parts(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.23164190f, ax, 1.0f));          // 0.3275911 / sqrt(2)
  float p = fmaf(0.5307027145f, t, -0.7265760135f);                            // 0.5 * (a5, a4, a3, a2, a1)
  p = fmaf(p, t, 0.7107068705f);
  p = fmaf(p, t, -0.142248368f);
  p = fmaf(p, t, 0.127414796f);
  const float zs = x * 0.84932180f;                                            // sqrt(0.5 * log2 e)
  GeluParts r;
  r.q = p * t;
  r.e = __builtin_amdgcn_exp2f(-(zs * zs));
  return r;
}
__device__ __forceinline__ float gelu_from_parts(float x, const GeluParts g) {
  return fmaf(-(fabsf(x) * g.q), g.e, fmaxf(x, 0.0f));
}
// d/dx gelu(x) = Phi(x) + x*phi(x)
__device__ __forceinline__ float gelu_grad_from_parts(float x, const GeluParts g) {
  const float step = (__builtin_bit_cast(int, x) >= 0) ? 1.0f : 0.0f;           // by the sign BIT: consistent with copysign at +-0
  return fmaf(g.e, fmaf(0.3989422804014327f, x, -copysignf(g.q, x)), step);
}
__device__ __forceinline__ float gelu_erf(float x) { return gelu_from_parts(x, gelu_parts(x)); }
__device__ __forceinline__ float gelu_erf_grad(float x) { return gelu_grad_from_parts(x, gelu_parts(x)); }
// The same arithmetic on PAIRS of values in packed-fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32: two IEEE operations per
// lane per issue, the same roundings as the scalar forms above - results are bit-identical, tests/test_hip_gemm_park.py): the
// epilogues that evaluate GELU on whole tiles spend 20-35 % of a launch in this arithmetic with the matrix pipes idle.
struct GeluParts2 { vl_f32x2 q, e; };
__device__ __forceinline__ GeluParts2 gelu_parts2(vl_f32x2 x) {
  const vl_f32x2 ax = __builtin_elementwise_abs(x);
  const vl_f32x2 d = __builtin_elementwise_fma((vl_f32x2)(0.23164190f), ax, (vl_f32x2)(1.0f));
  const vl_f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  vl_f32x2 p = __builtin_elementwise_fma((vl_f32x2)(0.5307027145f), t, (vl_f32x2)(-0.7265760135f));
  p = __builtin_elementwise_fma(p, t, (vl_f32x2)(0.7107068705f));
  p = __builtin_elementwise_fma(p, t, (vl_f32x2)(-0.142248368f));
  p = __builtin_elementwise_fma(p, t, (vl_f32x2)(0.127414796f));
  const vl_f32x2 zs = x * 0.84932180f;
  const vl_f32x2 z2 = zs * zs;
  GeluParts2 r;
  r.q = p * t;
  r.e = vl_f32x2{__builtin_amdgcn_exp2f(-z2[0]), __builtin_amdgcn_exp2f(-z2[1])};
  return r;
}
__device__ __forceinline__ vl_f32x2 gelu_from_parts2(vl_f32x2 x, const GeluParts2 g) {
  const vl_f32x2 m = {fmaxf(x[0], 0.0f), fmaxf(x[1], 0.0f)};
  return __builtin_elementwise_fma(-(__builtin_elementwise_abs(x) * g.q), g.e, m);
}
__device__ __forceinline__ vl_f32x2 gelu_grad_from_parts2(vl_f32x2 x, const GeluParts2 g) {
  const float x0 = x[0], x1 = x[1];      // (scalars first: __builtin_bit_cast applied to a vector ELEMENT reads element 0 both times)
  const vl_f32x2 step = {(__builtin_bit_cast(int, x0) >= 0) ? 1.0f : 0.0f, (__builtin_bit_cast(int, x1) >= 0) ? 1.0f : 0.0f};
  const vl_f32x2 sq = {copysignf(g.q[0], x0), copysignf(g.q[1], x1)};
  return __builtin_elementwise_fma(g.e, __builtin_elementwise_fma((vl_f32x2)(0.3989422804014327f), x, -sq), step);
}
__device__ __forceinline__ vl_f32x2 gelu_erf2(vl_f32x2 x) { return gelu_from_parts2(x, gelu_parts2(x)); }
__device__ __forceinline__ vl_f32x2 gelu_erf_grad2(vl_f32x2 x) { return gelu_grad_from_parts2(x, gelu_parts2(x)); }
__device__ __forceinline__ vl_f32x2 unpack2bf(unsigned int w) { return vl_f32x2{bf2f((bf16_t)(w & 0xffff)), bf2f((bf16_t)(w >> 16))}; }
__device__ __forceinline__ unsigned int pack2bf(vl_f32x2 v) {
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, vl_bf16x2));
}
// One packed pair of bf16 pre-activations -> packed bf16 (gelu, gelu'): the pair the forward of a TRAINED MLP leaves behind
// (VL_ACT_GELU_DSAVE), so that the dX GEMM's epilogue is a multiplication instead of a second erf evaluation.
__device__ __forceinline__ void gelu_and_grad_pk(unsigned int w, unsigned int& y, unsigned int& d) {
  const vl_f32x2 x = unpack2bf(w);
  const GeluParts2 g = gelu_parts2(x);
  y = pack2bf(gelu_from_parts2(x, g));
  d = pack2bf(gelu_grad_from_parts2(x, g));
}
__device__ __forceinline__ unsigned int mul_pk_bf16(unsigned int a, unsigned int b) {
  return pack2bf(bf2f((bf16_t)(a & 0xffff)) * bf2f((bf16_t)(b & 0xffff)), bf2f((bf16_t)(a >> 16)) * bf2f((bf16_t)(b >> 16)));
}
