// Row-wise backward kernels + optimizer for the trainable part of the path (HBM-bound, wave-per-row).
//   vl_layernorm_bwd      dx = LN'(dy) (+ upstream residual gradient), f32 and optional bf16 copy
//   vl_layernorm_bwd_params   dw += sum_r dy*xhat, db += sum_r dy          (trainable LayerNorms only)
//   vl_colsum             out[j] += sum_r a[r,j]                            (bias gradients)
//   vl_gelu_bf16          y = gelu(u)                                       (recompute of the MLP hidden)
//   vl_adamw_step         torch.optim.AdamW update (decoupled weight decay), one launch per tensor
//   vl_clamp_scalar       logit_scale.clamp_(0, ln 100)   (training/train.py:248-249)
// Autograd counterparts of open_clip/transformer.py:17-34 (LayerNorm), :226-234 (MLP) and of
// `optim.AdamW` as configured in training/depth/depth_tri_main.py:394-419.
#include "vl_common.h"
#include "vitlens_hip.h"

namespace {

__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const bf16_t* p) { return bf2f(*p); }

struct LnBwdP {
  const void* dy; const void* x; const float* mean; const float* rstd; const float* w;
  const float* dres;      // optional upstream gradient of the residual stream (added to dx)
  float* dx; bf16_t* dxb; // outputs (dx may alias dres)
  long dys, xs, dxs;      // row strides
  int rows, D;
};

template <typename TDY, typename TX>
__global__ void __launch_bounds__(256) ln_bwd_kernel(const LnBwdP p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.rows) return;
  const TDY* dy = (const TDY*)p.dy + (long)row * p.dys;
  const TX* x = (const TX*)p.x + (long)row * p.xs;
  const float mu = p.mean[row], rs = p.rstd[row];
  const int D = p.D;
  float s1 = 0.f, s2 = 0.f;
  for (int e = lane; e < D; e += 64) {
    const float g = ld1(dy + e) * p.w[e];
    const float xh = (ld1(x + e) - mu) * rs;
    s1 += g; s2 = fmaf(g, xh, s2);
  }
  s1 = wave_sum(s1) / (float)D; s2 = wave_sum(s2) / (float)D;
  for (int e = lane; e < D; e += 64) {
    const float g = ld1(dy + e) * p.w[e];
    const float xh = (ld1(x + e) - mu) * rs;
    float v = rs * (g - s1 - xh * s2);
    if (p.dres) v += p.dres[(long)row * p.dxs + e];
    if (p.dx) p.dx[(long)row * p.dxs + e] = v;
    if (p.dxb) p.dxb[(long)row * p.dxs + e] = f2bf(v);
  }
}

// thread per column, a slab of rows per block; atomics combine slabs
template <typename TDY, typename TX>
__global__ void __launch_bounds__(256) ln_bwd_params_kernel(const TDY* dy, long dys, const TX* x, long xs, const float* mean,
                                                            const float* rstd, float* dw, float* db, int rows, int D, int slab) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= D) return;
  const int r0 = blockIdx.y * slab, r1 = min(rows, r0 + slab);
  float a = 0.f, b = 0.f;
  for (int r = r0; r < r1; ++r) {
    const float g = ld1(dy + (long)r * dys + j);
    a = fmaf(g, (ld1(x + (long)r * xs + j) - mean[r]) * rstd[r], a);
    b += g;
  }
  atomicAdd(dw + j, a); atomicAdd(db + j, b);
}

template <typename T>
__global__ void __launch_bounds__(256) colsum_kernel(const T* a, long lda, float* out, int rows, int cols, int slab, float scale) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= cols) return;
  const int r0 = blockIdx.y * slab, r1 = min(rows, r0 + slab);
  float s = 0.f;
  for (int r = r0; r < r1; ++r) s += ld1(a + (long)r * lda + j);
  atomicAdd(out + j, s * scale);
}

__global__ void __launch_bounds__(256) gelu_kernel(const bf16_t* u, bf16_t* y, long n) {
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += (long)gridDim.x * 256 * 8) {
    const u32x4 v = *(const u32x4*)(u + i);
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      o[e] = pack2bf(gelu_erf(bf2f((bf16_t)(v[e] & 0xffff))), gelu_erf(bf2f((bf16_t)(v[e] >> 16))));
    *(u32x4*)(y + i) = o;
  }
}

__global__ void __launch_bounds__(256) geglu_kernel(const bf16_t* h, bf16_t* y, long n) {
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 256 * 4) {
    const u32x4 v = *(const u32x4*)(h + 2 * i);
    u32x2 o;
    o[0] = pack2bf(bf2f((bf16_t)(v[0] & 0xffff)) * gelu_erf(bf2f((bf16_t)(v[0] >> 16))),
                   bf2f((bf16_t)(v[1] & 0xffff)) * gelu_erf(bf2f((bf16_t)(v[1] >> 16))));
    o[1] = pack2bf(bf2f((bf16_t)(v[2] & 0xffff)) * gelu_erf(bf2f((bf16_t)(v[2] >> 16))),
                   bf2f((bf16_t)(v[3] & 0xffff)) * gelu_erf(bf2f((bf16_t)(v[3] >> 16))));
    *(u32x2*)(y + i) = o;
  }
}

// AdamW (PyTorch semantics): p *= 1 - lr*wd ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ;
//                            p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ void __launch_bounds__(256) adamw_kernel(float* p, const float* g, float* m, float* v, long n, float lr, float b1,
                                                    float b2, float eps, float wd, float bc1, float bc2_sqrt, float gscale) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float gi = g[i] * gscale;
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    pi -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
    p[i] = pi;
  }
}

__global__ void clamp_scalar_kernel(float* p, float lo, float hi) { *p = fminf(fmaxf(*p, lo), hi); }

// out[i] = a[i] + b[i] (f32), used to merge gradient contributions of shared tensors
__global__ void __launch_bounds__(256) axpy_kernel(float* y, const float* x, float alpha, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = fmaf(alpha, x[i], y[i]);
}

// rows -> sum over the batch index: out[t, :] += sum_b x[(b*T_stride + t_off + t), :]   (pos-embedding grads)
__global__ void __launch_bounds__(256) batch_rowsum_kernel(const float* x, float* out, int B, int T, int D, long bstride, long toff) {
  const long n = (long)T * D;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long t = i / D; const int d = (int)(i - t * D);
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += x[((long)b * bstride + toff + t) * D + d];
    out[i] += s;
  }
}

}  // namespace

extern "C" int vl_set_error(const char* msg);
#define VL_HIP_OK(e) do { hipError_t _e = (e); if (_e != hipSuccess) return vl_set_error(hipGetErrorString(_e)); } while (0)
static int grid_for(long n) { long g = (n + 255) / 256; return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g)); }

extern "C" int vl_layernorm_bwd(const void* dy, int dy_dtype, long dy_stride, const void* x, int x_dtype, long x_stride,
                                const float* mean, const float* rstd, const float* w, const float* dres, float* dx,
                                void* dx_bf16, long dx_stride, int rows, int D, hipStream_t stream) {
  if (rows <= 0 || D <= 0) return vl_set_error("vl_layernorm_bwd: empty problem");
  LnBwdP p{dy, x, mean, rstd, w, dres, dx, (bf16_t*)dx_bf16, dy_stride, x_stride, dx_stride, rows, D};
  const dim3 g((rows + 3) / 4), b(256);
  if (dy_dtype == VL_BF16 && x_dtype == VL_F32) hipLaunchKernelGGL((ln_bwd_kernel<bf16_t, float>), g, b, 0, stream, p);
  else if (dy_dtype == VL_BF16) hipLaunchKernelGGL((ln_bwd_kernel<bf16_t, bf16_t>), g, b, 0, stream, p);
  else if (x_dtype == VL_F32) hipLaunchKernelGGL((ln_bwd_kernel<float, float>), g, b, 0, stream, p);
  else hipLaunchKernelGGL((ln_bwd_kernel<float, bf16_t>), g, b, 0, stream, p);
  VL_HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int vl_layernorm_bwd_params(const void* dy, int dy_dtype, long dy_stride, const void* x, int x_dtype, long x_stride,
                                       const float* mean, const float* rstd, float* dw, float* db, int rows, int D,
                                       hipStream_t stream) {
  if (rows <= 0 || D <= 0) return vl_set_error("vl_layernorm_bwd_params: empty problem");
  const int slab = 256;
  const dim3 g((D + 255) / 256, (rows + slab - 1) / slab), b(256);
  if (dy_dtype == VL_BF16 && x_dtype == VL_F32)
    hipLaunchKernelGGL((ln_bwd_params_kernel<bf16_t, float>), g, b, 0, stream, (const bf16_t*)dy, dy_stride, (const float*)x, x_stride, mean, rstd, dw, db, rows, D, slab);
  else if (dy_dtype == VL_BF16)
    hipLaunchKernelGGL((ln_bwd_params_kernel<bf16_t, bf16_t>), g, b, 0, stream, (const bf16_t*)dy, dy_stride, (const bf16_t*)x, x_stride, mean, rstd, dw, db, rows, D, slab);
  else if (x_dtype == VL_F32)
    hipLaunchKernelGGL((ln_bwd_params_kernel<float, float>), g, b, 0, stream, (const float*)dy, dy_stride, (const float*)x, x_stride, mean, rstd, dw, db, rows, D, slab);
  else
    hipLaunchKernelGGL((ln_bwd_params_kernel<float, bf16_t>), g, b, 0, stream, (const float*)dy, dy_stride, (const bf16_t*)x, x_stride, mean, rstd, dw, db, rows, D, slab);
  VL_HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int vl_colsum(const void* a, int a_dtype, long lda, float* out, int rows, int cols, float scale, hipStream_t stream) {
  if (rows <= 0 || cols <= 0) return vl_set_error("vl_colsum: empty problem");
  const int slab = 256;
  const dim3 g((cols + 255) / 256, (rows + slab - 1) / slab), b(256);
  if (a_dtype == VL_BF16) hipLaunchKernelGGL(colsum_kernel<bf16_t>, g, b, 0, stream, (const bf16_t*)a, lda, out, rows, cols, slab, scale);
  else hipLaunchKernelGGL(colsum_kernel<float>, g, b, 0, stream, (const float*)a, lda, out, rows, cols, slab, scale);
  VL_HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int vl_gelu_bf16(const void* u, void* y, long n, hipStream_t stream) {
  if (n <= 0) return 0;
  if (n & 7) return vl_set_error("vl_gelu_bf16: n must be a multiple of 8");
  hipLaunchKernelGGL(gelu_kernel, dim3(grid_for(n / 8)), dim3(256), 0, stream, (const bf16_t*)u, (bf16_t*)y, n);
  VL_HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int vl_geglu_bf16(const void* h, void* y, long rows, int n_out, hipStream_t stream) {
  const long n = rows * n_out;
  if (n <= 0) return 0;
  if (n_out & 3) return vl_set_error("vl_geglu_bf16: n_out must be a multiple of 4");
  hipLaunchKernelGGL(geglu_kernel, dim3(grid_for(n / 4)), dim3(256), 0, stream, (const bf16_t*)h, (bf16_t*)y, n);
  VL_HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int vl_adamw_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                             float eps, float weight_decay, int step, float grad_scale, hipStream_t stream) {
  if (n <= 0) return 0;
  if (step < 1) return vl_set_error("vl_adamw_step: step counts from 1");
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.0f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n)), dim3(256), 0, stream, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay,
                     bc1, bc2s, grad_scale);
  VL_HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int vl_clamp_scalar(float* p, float lo, float hi, hipStream_t stream) {
  hipLaunchKernelGGL(clamp_scalar_kernel, dim3(1), dim3(1), 0, stream, p, lo, hi);
  VL_HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int vl_axpy_f32(float* y, const float* x, float alpha, long n, hipStream_t stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n)), dim3(256), 0, stream, y, x, alpha, n);
  VL_HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int vl_batch_rowsum(const float* x, float* out, int B, int T, int D, long batch_stride_rows, long row_offset,
                               hipStream_t stream) {
  if (B <= 0 || T <= 0 || D <= 0) return vl_set_error("vl_batch_rowsum: empty problem");
  hipLaunchKernelGGL(batch_rowsum_kernel, dim3(grid_for((long)T * D)), dim3(256), 0, stream, x, out, B, T, D, batch_stride_rows, row_offset);
  VL_HIP_OK(hipGetLastError());
  return 0;
}
