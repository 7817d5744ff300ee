// Evaluation-time image / depth preprocessing on the GPU: the separable bicubic resamplers the
// reference's torchvision transforms end in, fused with crop and normalisation.
//
//   8-bit path  (open_clip/transform.py:138-155: Resize(BICUBIC) -> CenterCrop -> ToTensor -> Normalize on a PIL image)
//     = Pillow's two-pass resampler: 22-bit fixed-point coefficients, int32 accumulation from 1 << 21, arithmetic shift,
//     clip to [0, 255], with the HORIZONTAL result rounded to 8 bits before the vertical pass.  Both kernels reproduce
//     that integer arithmetic exactly (byte-exact vs Image.resize); the coefficient tables are built on the host in
//     double precision, as Pillow builds them (vitlens_hip/preproc.py).  The vertical kernel finishes with
//     (u8 / 255 - mean) / std in IEEE fp32 (correctly rounded divisions, no contraction): bit-exact vs torch.
//   float path  (modal_depth/processors/vt_processor.py:292-337: DepthNorm -> Resize(bicubic) -> CenterCrop -> Normalize
//     on a tensor) = ATen's separable bicubic (antialiased: normalised a = -0.5 taps over the scaled support; plain: the
//     four a = -0.75 taps with border clamp).  The clamp-and-scale of DepthNorm is applied as the source is read.
//
// Only the crop window is computed: the horizontal pass produces the columns of the crop for the source rows the
// vertical taps of the crop's rows touch.  HBM-bound byte work: one thread per output pixel, taps through L1/L2
// (adjacent outputs share all but one tap), outputs coalesced along x.
#include "vl_common.h"
#include "vitlens_hip.h"

extern "C" int vl_set_error(const char* msg);
#define VL_HIP_OK(e) do { hipError_t _e = (e); if (_e != hipSuccess) return vl_set_error(hipGetErrorString(_e)); } while (0)

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ unsigned char clip8(int acc) {
  const int v = acc >> PRECISION_BITS;
  return (unsigned char)min(max(v, 0), 255);
}

// Tap loop of one output pixel: acc[c] += k[j] * s[j * step + c] for the C channels.  With 3 or 4 channels the bytes of a
// pixel come from ONE unaligned 4-byte load (gfx950 runs with unaligned access enabled; hipcc itself lowers an align-1
// 4-byte copy to global_load_dword) instead of C byte loads - the loop is bound by load instructions, not by bytes.  The
// fourth byte of a 3-channel pixel belongs to the next pixel; `end` is one past the last readable byte, and a thread whose
// last tap would read beyond it (the final pixel of a buffer only) takes the byte loop.
template <int C>
__device__ __forceinline__ void tap_loop(const unsigned char* __restrict__ s, long step, const int* __restrict__ k, int n,
                                         const unsigned char* end, int (&acc)[C]) {
  if (C >= 3 && s + (long)(n - 1) * step + 4 <= end) {
    for (int j = 0; j < n; ++j) {
      unsigned v;
      __builtin_memcpy(&v, s + (long)j * step, 4);
      const int kj = k[j];
#pragma unroll
      for (int c = 0; c < C; ++c) acc[c] += (int)((v >> (8 * c)) & 0xffu) * kj;
    }
  } else {
    for (int j = 0; j < n; ++j) {
      const int kj = k[j];
#pragma unroll
      for (int c = 0; c < C; ++c) acc[c] += (int)s[(long)j * step + c] * kj;
    }
  }
}

template <int C>
__device__ __forceinline__ void h_pixel(const unsigned char* __restrict__ s, const int* __restrict__ k, int n, const unsigned char* end,
                                        unsigned char* __restrict__ dst) {
  int acc[C];
#pragma unroll
  for (int c = 0; c < C; ++c) acc[c] = 1 << (PRECISION_BITS - 1);
  tap_loop<C>(s, C, k, n, end, acc);
#pragma unroll
  for (int c = 0; c < C; ++c) dst[c] = clip8(acc[c]);
}

__device__ __forceinline__ void h_pixel_any(int C, const unsigned char* s, const int* k, int n, const unsigned char* end, unsigned char* dst) {
  switch (C) {
    case 1: h_pixel<1>(s, k, n, end, dst); break;
    case 2: h_pixel<2>(s, k, n, end, dst); break;
    case 3: h_pixel<3>(s, k, n, end, dst); break;
    default: h_pixel<4>(s, k, n, end, dst); break;
  }
}

// dst[r, i, c] = clip8(2^21 + sum_j kk[xout0+i, j] * src[row0+r, bounds[xout0+i].min + j, c])
__global__ void __launch_bounds__(256) resample_h_u8_kernel(const unsigned char* __restrict__ src, long row_stride, int W, int C, int row0,
                                                            int nrows, const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                                            int xout0, int nxout, unsigned char* __restrict__ dst) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)nrows * nxout) return;
  const int r = (int)(i / nxout), x = (int)(i - (long)r * nxout);
  const int xo = xout0 + x;
  const int xmin = bounds[2 * xo], n = bounds[2 * xo + 1];
  const unsigned char* end = src + (long)(row0 + nrows - 1) * row_stride + (long)W * C;
  h_pixel_any(C, src + (long)(row0 + r) * row_stride + (long)xmin * C, kk + (long)xo * ksize, n, end, dst + ((long)r * nxout + x) * C);
}

struct NormP { float mean[4], std[4]; };

// out[c, y, x] = (clip8(2^21 + sum_j kk[yout0+y, j] * src[bounds[yout0+y].min + j - row0, x, c]) / 255 - mean[c]) / std[c]
template <int C>
__device__ __forceinline__ void v_pixel(const unsigned char* __restrict__ s, long step, const int* __restrict__ k, int n,
                                        const unsigned char* end, const NormP& np, float* __restrict__ out, long plane,
                                        unsigned char* __restrict__ out_u8) {
#pragma clang fp contract(off)
  int acc[C];
#pragma unroll
  for (int c = 0; c < C; ++c) acc[c] = 1 << (PRECISION_BITS - 1);
  tap_loop<C>(s, step, k, n, end, acc);
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const unsigned char u = clip8(acc[c]);
    if (out_u8) out_u8[c] = u;
    if (out) {
      const float v = (float)u / 255.0f;
      out[c * plane] = (v - np.mean[c]) / np.std[c];
    }
  }
}

__device__ __forceinline__ void v_pixel_any(int C, const unsigned char* s, long step, const int* k, int n, const unsigned char* end,
                                            const NormP& np, float* out, long plane, unsigned char* out_u8) {
  switch (C) {
    case 1: v_pixel<1>(s, step, k, n, end, np, out, plane, out_u8); break;
    case 2: v_pixel<2>(s, step, k, n, end, np, out, plane, out_u8); break;
    case 3: v_pixel<3>(s, step, k, n, end, np, out, plane, out_u8); break;
    default: v_pixel<4>(s, step, k, n, end, np, out, plane, out_u8); break;
  }
}

__global__ void __launch_bounds__(256) resample_v_u8_norm_kernel(const unsigned char* __restrict__ src, int W, int C, int row0, int nrows,
                                                                 const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                                                 int yout0, int nyout, const NormP np, float* __restrict__ out,
                                                                 unsigned char* __restrict__ out_u8) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)nyout * W) return;
  const int y = (int)(i / W), x = (int)(i - (long)y * W);
  const int yo = yout0 + y;
  const int ymin = bounds[2 * yo], n = bounds[2 * yo + 1];
  v_pixel_any(C, src + ((long)(ymin - row0) * W + x) * C, (long)W * C, kk + (long)yo * ksize, n, src + (long)nrows * W * C, np,
              out ? out + (long)y * W + x : nullptr, (long)nyout * W, out_u8 ? out_u8 + ((long)y * W + x) * C : nullptr);
}

// ---- batched 8-bit path: one launch per pass for a list of images of DIFFERENT sizes (blockIdx.z = image) ----
// desc [n][16] int64 per image: 0 src pointer, 1 row stride (bytes), 2 source width, 3 row0, 4 nrows (source rows the
// vertical taps of the crop touch), 5 xout0, 6 yout0 (crop origin in the resized image), 7/8/9 horizontal bounds offset,
// coefficient offset (in ints, into `tables`) and ksize, 10/11/12 the same for the vertical axis, 13 offset of this
// image's intermediate in `tmp` (bytes), 14-15 unused.
constexpr int DESC_LD = 16;

__global__ void __launch_bounds__(256) resample_h_u8_batch_kernel(const long long* __restrict__ desc, int C, int nxout,
                                                                  const int* __restrict__ tables, unsigned char* __restrict__ tmp) {
  const long long* d = desc + (long)blockIdx.z * DESC_LD;
  const int nrows = (int)d[4];
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)nrows * nxout) return;
  const int r = (int)(i / nxout), x = (int)(i - (long)r * nxout);
  const int xo = (int)d[5] + x, ksize = (int)d[9];
  const int* bounds = tables + d[7];
  const int xmin = bounds[2 * xo], n = bounds[2 * xo + 1];
  const unsigned char* src = (const unsigned char*)d[0];
  const unsigned char* end = src + (d[3] + nrows - 1) * d[1] + d[2] * C;
  h_pixel_any(C, src + (d[3] + r) * d[1] + (long)xmin * C, tables + d[8] + (long)xo * ksize, n, end, tmp + d[13] + ((long)r * nxout + x) * C);
}

__global__ void __launch_bounds__(256) resample_v_u8_norm_batch_kernel(const long long* __restrict__ desc, int C, int nyout, int W,
                                                                       const int* __restrict__ tables, const unsigned char* __restrict__ tmp,
                                                                       const NormP np, float* __restrict__ out) {
  const long long* d = desc + (long)blockIdx.z * DESC_LD;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)nyout * W) return;
  const int y = (int)(i / W), x = (int)(i - (long)y * W);
  const int yo = (int)d[6] + y, ksize = (int)d[12], row0 = (int)d[3];
  const int* bounds = tables + d[10];
  const int ymin = bounds[2 * yo], n = bounds[2 * yo + 1];
  const unsigned char* t = tmp + d[13];
  v_pixel_any(C, t + ((long)(ymin - row0) * W + x) * C, (long)W * C, tables + d[11] + (long)yo * ksize, n, t + d[4] * W * C, np,
              out + (long)blockIdx.z * C * nyout * W + (long)y * W + x, (long)nyout * W, nullptr);
}

struct ClampP { float lo, hi, div; int on; };

// dst[r, i] = sum_j w[xout0+i, j] * f(src[row0+r, clamp(xmin[xout0+i] + j, 0, W-1)]),  f = DepthNorm's clamp and scale
__global__ void __launch_bounds__(256) resample_h_f32_kernel(const float* __restrict__ src, long row_stride, int W, int row0, int nrows,
                                                             const int* __restrict__ bounds, const float* __restrict__ wt, int ksize,
                                                             int xout0, int nxout, const ClampP cp, float* __restrict__ dst) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)nrows * nxout) return;
  const int r = (int)(i / nxout), x = (int)(i - (long)r * nxout);
  const int xo = xout0 + x;
  const int xmin = bounds[2 * xo], n = bounds[2 * xo + 1];
  const float* w = wt + (long)xo * ksize;
  const float* s = src + (long)(row0 + r) * row_stride;
  float acc = 0.f;
  for (int j = 0; j < n; ++j) {
    float v = s[min(max(xmin + j, 0), W - 1)];
    if (cp.on) v = fminf(fmaxf(v, cp.lo), cp.hi) / cp.div;
    acc = fmaf(w[j], v, acc);
  }
  dst[(long)r * nxout + x] = acc;
}

// out[y, x] = (sum_j w[yout0+y, j] * src[clamp(ymin + j, 0, H-1) - row0, x] - mean) / std
__global__ void __launch_bounds__(256) resample_v_f32_norm_kernel(const float* __restrict__ src, int W, int H, int row0,
                                                                  const int* __restrict__ bounds, const float* __restrict__ wt, int ksize,
                                                                  int yout0, int nyout, float mean, float stdv, float* __restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)nyout * W) return;
  const int y = (int)(i / W), x = (int)(i - (long)y * W);
  const int yo = yout0 + y;
  const int ymin = bounds[2 * yo], n = bounds[2 * yo + 1];
  const float* w = wt + (long)yo * ksize;
  float acc = 0.f;
  for (int j = 0; j < n; ++j) {
    const int yy = min(max(ymin + j, 0), H - 1) - row0;
    acc = fmaf(w[j], src[(long)yy * W + x], acc);
  }
  out[(long)y * W + x] = (acc - mean) / stdv;
}

inline unsigned grid1(long n) { return (unsigned)((n + 255) / 256); }

}  // namespace

extern "C" int vl_resample_h_u8(const uint8_t* src, long row_stride, int W, int C, int row0, int nrows, const int* bounds, const int* kk,
                                int ksize, int xout0, int nxout, uint8_t* dst, hipStream_t stream) {
  if (C < 1 || C > 4 || W <= 0 || nrows <= 0 || nxout <= 0 || ksize <= 0 || row0 < 0 || xout0 < 0)
    return vl_set_error("vl_resample_h_u8: need 1<=C<=4, W, nrows, nxout, ksize > 0");
  hipLaunchKernelGGL(resample_h_u8_kernel, dim3(grid1((long)nrows * nxout)), dim3(256), 0, stream, src, row_stride, W, C, row0, nrows, bounds,
                     kk, ksize, xout0, nxout, dst);
  VL_HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int vl_resample_v_u8_norm(const uint8_t* src, int W, int C, int row0, int nrows, const int* bounds, const int* kk, int ksize,
                                     int yout0, int nyout, const float* mean, const float* stdv, float* out, uint8_t* out_u8,
                                     hipStream_t stream) {
  if (C < 1 || C > 4 || W <= 0 || nrows <= 0 || nyout <= 0 || ksize <= 0 || row0 < 0 || yout0 < 0 || (!out && !out_u8) || (out && (!mean || !stdv)))
    return vl_set_error("vl_resample_v_u8_norm: need 1<=C<=4, W, nrows, nyout, ksize > 0, a destination, and mean/std for the float one");
  NormP np{};
  for (int c = 0; c < C && out; ++c) { np.mean[c] = mean[c]; np.std[c] = stdv[c]; }
  hipLaunchKernelGGL(resample_v_u8_norm_kernel, dim3(grid1((long)nyout * W)), dim3(256), 0, stream, src, W, C, row0, nrows, bounds, kk,
                     ksize, yout0, nyout, np, out, out_u8);
  VL_HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int vl_resample_h_f32(const float* src, long row_stride, int W, int row0, int nrows, const int* bounds, const float* weights,
                                 int ksize, int xout0, int nxout, int clamp_on, float clamp_lo, float clamp_hi, float divide_by,
                                 float* dst, hipStream_t stream) {
  if (W <= 0 || nrows <= 0 || nxout <= 0 || ksize <= 0 || row0 < 0 || xout0 < 0 || (clamp_on && divide_by == 0.f))
    return vl_set_error("vl_resample_h_f32: need W, nrows, nxout, ksize > 0 and a non-zero divisor");
  const ClampP cp{clamp_lo, clamp_hi, divide_by, clamp_on};
  hipLaunchKernelGGL(resample_h_f32_kernel, dim3(grid1((long)nrows * nxout)), dim3(256), 0, stream, src, row_stride, W, row0, nrows, bounds,
                     weights, ksize, xout0, nxout, cp, dst);
  VL_HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int vl_resample_v_f32_norm(const float* src, int W, int H, int row0, const int* bounds, const float* weights, int ksize,
                                      int yout0, int nyout, float mean, float stdv, float* out, hipStream_t stream) {
  if (W <= 0 || H <= 0 || nyout <= 0 || ksize <= 0 || row0 < 0 || yout0 < 0 || stdv == 0.f)
    return vl_set_error("vl_resample_v_f32_norm: need W, H, nyout, ksize > 0 and std != 0");
  hipLaunchKernelGGL(resample_v_f32_norm_kernel, dim3(grid1((long)nyout * W)), dim3(256), 0, stream, src, W, H, row0, bounds, weights,
                     ksize, yout0, nyout, mean, stdv, out);
  VL_HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int vl_resample_batch_u8_norm(const int64_t* desc, int n, int C, int crop_h, int crop_w, int max_nrows, const int* tables,
                                         uint8_t* tmp, const float* mean, const float* stdv, float* out, hipStream_t stream) {
  if (n <= 0 || n > 65535 || C < 1 || C > 4 || crop_h <= 0 || crop_w <= 0 || max_nrows <= 0 || !desc || !tables || !tmp || !mean || !stdv || !out)
    return vl_set_error("vl_resample_batch_u8_norm: need 1<=n<=65535, 1<=C<=4, crop and max_nrows > 0, all pointers");
  NormP np{};
  for (int c = 0; c < C; ++c) { np.mean[c] = mean[c]; np.std[c] = stdv[c]; }
  hipLaunchKernelGGL(resample_h_u8_batch_kernel, dim3(grid1((long)max_nrows * crop_w), 1, n), dim3(256), 0, stream, (const long long*)desc, C,
                     crop_w, tables, tmp);
  hipLaunchKernelGGL(resample_v_u8_norm_batch_kernel, dim3(grid1((long)crop_h * crop_w), 1, n), dim3(256), 0, stream, (const long long*)desc,
                     C, crop_h, crop_w, tables, (const unsigned char*)tmp, np, out);
  VL_HIP_OK(hipGetLastError());
  return 0;
}
