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

__device__ __forceinline__ void ld4(const float* p, float (&v)[4]) {
  const f32x4 t = *(const f32x4*)p; v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
}
__device__ __forceinline__ void ld4(const bf16_t* p, float (&v)[4]) {
  const u32x2 t = *(const u32x2*)p;
  v[0] = bf2f((bf16_t)(t[0] & 0xffff)); v[1] = bf2f((bf16_t)(t[0] >> 16));
  v[2] = bf2f((bf16_t)(t[1] & 0xffff)); v[3] = bf2f((bf16_t)(t[1] >> 16));
}
__device__ __forceinline__ void st4(float* p, const float (&v)[4]) { f32x4 t = {v[0], v[1], v[2], v[3]}; *(f32x4*)p = t; }
__device__ __forceinline__ void st4(bf16_t* p, const float (&v)[4]) {
  u32x2 t; t[0] = pack2bf(v[0], v[1]); t[1] = pack2bf(v[2], v[3]); *(u32x2*)p = t;
}

struct LnBwdP {
  const void* dy; const void* x; const float* mean; const float* rstd; const float* w;
  const void* dres;       // optional upstream gradient of the residual stream (added to dx), gradient-stream dtype
  void* dx; bf16_t* dxb;  // outputs: dx in the gradient-stream dtype (may alias dres), optional bf16 copy
  long dys, xs, dxs;      // row strides
  int rows, D;
};

// One wave per row.  NCH > 0: the row (D = NCH*256 elements) is read ONCE into registers with 8/16-byte accesses;
// NCH == 0: any D, two passes over global memory.  TG = dtype of the residual-gradient stream (dres / dx).
template <int NCH, typename TDY, typename TX, typename TG>
__global__ void __launch_bounds__(256) ln_bwd_kernel(const LnBwdP p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.rows) return;
  const TDY* dy = (const TDY*)p.dy + (long)row * p.dys;
  const TX* x = (const TX*)p.x + (long)row * p.xs;
  const TG* dres = p.dres ? (const TG*)p.dres + (long)row * p.dxs : nullptr;
  TG* dx = p.dx ? (TG*)p.dx + (long)row * p.dxs : nullptr;
  bf16_t* dxb = p.dxb ? p.dxb + (long)row * p.dxs : nullptr;
  const float mu = p.mean[row], rs = p.rstd[row];
  const int D = p.D;
  if constexpr (NCH > 0) {
    float g[NCH][4], xh[NCH][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int e = c * 256 + lane * 4;
      float ww[4];
      ld4(dy + e, g[c]); ld4(x + e, xh[c]); ld4(p.w + e, ww);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        g[c][k] *= ww[k]; xh[c][k] = (xh[c][k] - mu) * rs;
        s1 += g[c][k]; s2 = fmaf(g[c][k], xh[c][k], s2);
      }
    }
    s1 = wave_sum(s1) / (float)D; s2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int e = c * 256 + lane * 4;
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = rs * (g[c][k] - s1 - xh[c][k] * s2);
      if (dres) {
        float r[4]; ld4(dres + e, r);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] += r[k];
      }
      if (dx) st4(dx + e, v);
      if (dxb) st4(dxb + e, v);
    }
  } else {
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
      if (dres) v += ld1(dres + e);
      if (dx) { if constexpr (sizeof(TG) == 4) dx[e] = v; else dx[e] = f2bf(v); }
      if (dxb) dxb[e] = f2bf(v);
    }
  }
}

// LayerNorm weight / bias gradients, deterministic two-stage column reduction (no atomics):
// stage 1: block (column block, row slab) -> partial sums ws[slab][2][D]; stage 2: fixed-order sum over slabs, += into dw / db.
template <typename TDY, typename TX>
__global__ void __launch_bounds__(256) ln_bwd_params_kernel(const TDY* dy, long dys, const TX* x, long xs, const float* mean,
                                                            const float* rstd, float* ws, int rows, int D, int slab) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= D) return;
  const int r0 = blockIdx.y * slab, r1 = min(rows, r0 + slab);
  float a = 0.f, b = 0.f;
  for (int r = r0; r < r1; ++r) {
    const float g = ld1(dy + (long)r * dys + j);
    a = fmaf(g, (ld1(x + (long)r * xs + j) - mean[r]) * rstd[r], a);
    b += g;
  }
  ws[((long)blockIdx.y * 2) * D + j] = a; ws[((long)blockIdx.y * 2 + 1) * D + j] = b;
}

// bf16 dy and bf16 x with 16-byte friendly rows: 8 columns per thread, four row lanes per block (as colsum8_bf16_kernel; the
// one-column-per-thread kernel above ran at 2 TB/s: 132 us per call on [65 792, 1 024]).  Same partial layout, same finalize.
__global__ void __launch_bounds__(256) ln_bwd_params8_bf16_kernel(const bf16_t* dy, long dys, const bf16_t* x, long xs, const float* mean,
                                                                  const float* rstd, float* ws, int rows, int D, int slab) {
  __shared__ float sh[3][64][16];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int j = (blockIdx.x * 64 + cx) * 8;
  const int r0 = blockIdx.y * slab, r1 = min(rows, r0 + slab);
  float a[8], b[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { a[e] = 0.f; b[e] = 0.f; }
  if (j < D) {
    for (int r = r0 + ry; r < r1; r += 8) {
      const bool two = r + 4 < r1;
      const int r2 = two ? r + 4 : r;
      const u32x4 g0 = *(const u32x4*)(dy + (long)r * dys + j), x0 = *(const u32x4*)(x + (long)r * xs + j);
      const u32x4 g1 = *(const u32x4*)(dy + (long)r2 * dys + j), x1 = *(const u32x4*)(x + (long)r2 * xs + j);
      const float m0 = mean[r], s0 = rstd[r], m1 = mean[r2], s1 = two ? rstd[r2] : 0.f;
      const float k1 = two ? 1.f : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float gl = bf2f((bf16_t)(g0[e] & 0xffff)), gh = bf2f((bf16_t)(g0[e] >> 16));
        a[2 * e] = fmaf(gl, (bf2f((bf16_t)(x0[e] & 0xffff)) - m0) * s0, a[2 * e]);
        a[2 * e + 1] = fmaf(gh, (bf2f((bf16_t)(x0[e] >> 16)) - m0) * s0, a[2 * e + 1]);
        b[2 * e] += gl; b[2 * e + 1] += gh;
        const float hl = bf2f((bf16_t)(g1[e] & 0xffff)), hh = bf2f((bf16_t)(g1[e] >> 16));
        a[2 * e] = fmaf(hl, (bf2f((bf16_t)(x1[e] & 0xffff)) - m1) * s1, a[2 * e]);
        a[2 * e + 1] = fmaf(hh, (bf2f((bf16_t)(x1[e] >> 16)) - m1) * s1, a[2 * e + 1]);
        b[2 * e] = fmaf(hl, k1, b[2 * e]); b[2 * e + 1] = fmaf(hh, k1, b[2 * e + 1]);
      }
    }
  }
  if (ry) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { sh[ry - 1][cx][e] = a[e]; sh[ry - 1][cx][8 + e] = b[e]; }
  }
  __syncthreads();
  if (ry == 0 && j < D) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      ws[((long)blockIdx.y * 2) * D + j + e] = (a[e] + sh[0][cx][e]) + (sh[1][cx][e] + sh[2][cx][e]);
      ws[((long)blockIdx.y * 2 + 1) * D + j + e] = (b[e] + sh[0][cx][8 + e]) + (sh[1][cx][8 + e] + sh[2][cx][8 + e]);
    }
  }
}

// fp32 rows that are 16-byte friendly (the Perceiver's fp32 residual-gradient stream: bias gradients of to_out / FF w2): 4 columns
// per thread, four row lanes with four loads each in flight - the one-column-per-thread kernel ran at 2.3 TB/s (116 us on
// [65 536, 1 024] in the C4 step).  Same partial layout, same finalize.
__global__ void __launch_bounds__(256) colsum4_f32_kernel(const float* a, long lda, float* ws, int rows, int cols, int slab) {
  __shared__ float sh[3][64][4];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int j = (blockIdx.x * 64 + cx) * 4;
  const int r0 = blockIdx.y * slab, r1 = min(rows, r0 + slab);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (j < cols) {
    const float* ptr = a + j;
    int r = r0 + ry;
    for (; r + 12 < r1; r += 16) {
      const f32x4 v0 = *(const f32x4*)(ptr + (long)r * lda), v1 = *(const f32x4*)(ptr + (long)(r + 4) * lda);
      const f32x4 v2 = *(const f32x4*)(ptr + (long)(r + 8) * lda), v3 = *(const f32x4*)(ptr + (long)(r + 12) * lda);
      s += v0; s += v1; s += v2; s += v3;
    }
    for (; r < r1; r += 4) s += *(const f32x4*)(ptr + (long)r * lda);
  }
  if (ry) {
#pragma unroll
    for (int e = 0; e < 4; ++e) sh[ry - 1][cx][e] = s[e];
  }
  __syncthreads();
  if (ry == 0 && j < cols) {
#pragma unroll
    for (int e = 0; e < 4; ++e) ws[(long)blockIdx.y * cols + j + e] = (s[e] + sh[0][cx][e]) + (sh[1][cx][e] + sh[2][cx][e]);
  }
}

// bf16 dy with fp32 x (the Perceiver's LayerNorms: bf16 gradient of the normalised rows, fp32 residual stream), 8- / 16-byte
// friendly rows: 4 columns per thread, four row lanes, two rows per iteration (the generic kernel: 139 us on [65 536, 1 024]).
__global__ void __launch_bounds__(256) ln_bwd_params4_bf16_f32_kernel(const bf16_t* dy, long dys, const float* x, long xs, const float* mean,
                                                                      const float* rstd, float* ws, int rows, int D, int slab) {
  __shared__ float sh[3][64][8];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int j = (blockIdx.x * 64 + cx) * 4;
  const int r0 = blockIdx.y * slab, r1 = min(rows, r0 + slab);
  float a[4], b[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { a[e] = 0.f; b[e] = 0.f; }
  if (j < D) {
    for (int r = r0 + ry; r < r1; r += 8) {
      const bool two = r + 4 < r1;
      const int r2 = two ? r + 4 : r;
      const u32x2 g0 = *(const u32x2*)(dy + (long)r * dys + j), g1 = *(const u32x2*)(dy + (long)r2 * dys + j);
      const f32x4 x0 = *(const f32x4*)(x + (long)r * xs + j), x1 = *(const f32x4*)(x + (long)r2 * xs + j);
      const float m0 = mean[r], s0 = rstd[r], m1 = mean[r2], s1 = two ? rstd[r2] : 0.f;
      const float k1 = two ? 1.f : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float gl = bf2f((bf16_t)((e & 1) ? (g0[e >> 1] >> 16) : (g0[e >> 1] & 0xffff)));
        const float hl = bf2f((bf16_t)((e & 1) ? (g1[e >> 1] >> 16) : (g1[e >> 1] & 0xffff)));
        a[e] = fmaf(gl, (x0[e] - m0) * s0, a[e]);
        b[e] += gl;
        a[e] = fmaf(hl, (x1[e] - m1) * s1, a[e]);
        b[e] = fmaf(hl, k1, b[e]);
      }
    }
  }
  if (ry) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { sh[ry - 1][cx][e] = a[e]; sh[ry - 1][cx][4 + e] = b[e]; }
  }
  __syncthreads();
  if (ry == 0 && j < D) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ws[((long)blockIdx.y * 2) * D + j + e] = (a[e] + sh[0][cx][e]) + (sh[1][cx][e] + sh[2][cx][e]);
      ws[((long)blockIdx.y * 2 + 1) * D + j + e] = (b[e] + sh[0][cx][4 + e]) + (sh[1][cx][4 + e] + sh[2][cx][4 + e]);
    }
  }
}

// out_k[j] += scale * sum_s ws[s][k][j]   (k < K planes), fixed summation order.
// One block = 64 columns of one plane x 16 slab groups: thread (j, g) adds the slabs s = g, g+16, ... in order, the 16
// group sums are combined by a fixed tree through LDS.  (The first version gave every column ONE thread and the whole
// grid 4 blocks: 514 dependent loads per thread, 120 us per call = 4 ms of the C3 step.)
__global__ void __launch_bounds__(1024) slab_finalize_kernel(const float* ws, int nslab, int K, int D, float scale, float* out0, float* out1) {
  __shared__ float sh[16][64];
  const int jl = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + jl, k = blockIdx.y;
  float s = 0.f;
  if (j < D)
    for (int t = g; t < nslab; t += 16) s += ws[((long)t * K + k) * D + j];
  sh[g][jl] = s;
  __syncthreads();
  if (g == 0 && j < D) {
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = sh[i][jl];
#pragma unroll
    for (int w = 8; w > 0; w >>= 1)
#pragma unroll
      for (int i = 0; i < w; ++i) a[i] += a[i + w];
    float* o = k == 0 ? out0 : out1;
    o[j] += a[0] * scale;
  }
}

// bias gradients: stage 1 of the same deterministic two-stage column reduction (partials ws[slab][cols])
template <typename T>
__global__ void __launch_bounds__(256) colsum_kernel(const T* a, long lda, float* ws, int rows, int cols, int slab) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= cols) return;
  const int r0 = blockIdx.y * slab, r1 = min(rows, r0 + slab);
  float s = 0.f;
  for (int r = r0; r < r1; ++r) s += ld1(a + (long)r * lda + j);
  ws[(long)blockIdx.y * cols + j] = s;
}

// the same for bf16 inputs whose rows are 16-byte friendly: 8 columns per thread (one 16-byte load per row), four row lanes
// per block with four loads each in flight, reduced through LDS.  (The one-column-per-thread kernel moves 128 bytes per wave
// load: 1.6 TB/s on a [65 792, 4 096] gradient, 0.33 ms - more than both transposes of the path it now stands beside.)
__global__ void __launch_bounds__(256) colsum8_bf16_kernel(const bf16_t* a, long lda, float* ws, int rows, int cols, int slab) {
  __shared__ float sh[3][64][8];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int j = (blockIdx.x * 64 + cx) * 8;
  const int r0 = blockIdx.y * slab, r1 = min(rows, r0 + slab);
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  auto add = [&](const u32x4 v) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { s[2 * e] += bf2f((bf16_t)(v[e] & 0xffff)); s[2 * e + 1] += bf2f((bf16_t)(v[e] >> 16)); }
  };
  if (j < cols) {
    const bf16_t* ptr = a + j;
    int r = r0 + ry;
    for (; r + 12 < r1; r += 16) {
      const u32x4 v0 = *(const u32x4*)(ptr + (long)r * lda), v1 = *(const u32x4*)(ptr + (long)(r + 4) * lda);
      const u32x4 v2 = *(const u32x4*)(ptr + (long)(r + 8) * lda), v3 = *(const u32x4*)(ptr + (long)(r + 12) * lda);
      add(v0); add(v1); add(v2); add(v3);
    }
    for (; r < r1; r += 4) add(*(const u32x4*)(ptr + (long)r * lda));
  }
  if (ry) {
#pragma unroll
    for (int e = 0; e < 8; ++e) sh[ry - 1][cx][e] = s[e];
  }
  __syncthreads();
  if (ry == 0 && j < cols) {
#pragma unroll
    for (int e = 0; e < 8; ++e) ws[(long)blockIdx.y * cols + j + e] = (s[e] + sh[0][cx][e]) + (sh[1][cx][e] + sh[2][cx][e]);
  }
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

// out[i] = x[i] * exp(*log_scale) * mul: the learnable temperature applied ON THE DEVICE (model.py:619 `logit_scale.exp()`):
// the step never reads the scalar on the host
__global__ void __launch_bounds__(256) scale_exp_kernel(const float* x, float* out, long n, const float* log_scale, float mul) {
  const float s = __expf(log_scale[0]) * mul;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = x[i] * s;
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

template <typename TDY, typename TX, typename TG>
static void ln_bwd_launch(const LnBwdP& p, hipStream_t stream) {
  const dim3 g((p.rows + 3) / 4), b(256);
  const bool vec = (p.D % 256 == 0) && (p.dys % 4 == 0) && (p.xs % 4 == 0) && (p.dxs % 4 == 0);
  if (vec && p.D == 1024) hipLaunchKernelGGL((ln_bwd_kernel<4, TDY, TX, TG>), g, b, 0, stream, p);
  else if (vec && p.D == 768) hipLaunchKernelGGL((ln_bwd_kernel<3, TDY, TX, TG>), g, b, 0, stream, p);
  else if (vec && p.D == 512) hipLaunchKernelGGL((ln_bwd_kernel<2, TDY, TX, TG>), g, b, 0, stream, p);
  else if (vec && p.D == 256) hipLaunchKernelGGL((ln_bwd_kernel<1, TDY, TX, TG>), g, b, 0, stream, p);
  else hipLaunchKernelGGL((ln_bwd_kernel<0, TDY, TX, TG>), g, b, 0, stream, p);
}

extern "C" int vl_layernorm_bwd_g(const void* dy, int dy_dtype, long dy_stride, const void* x, int x_dtype, long x_stride,
                                  const float* mean, const float* rstd, const float* w, const void* dres, void* dx, int g_dtype,
                                  void* dx_bf16, long dx_stride, int rows, int D, hipStream_t stream) {
  if (rows <= 0 || D <= 0) return vl_set_error("vl_layernorm_bwd: empty problem");
  LnBwdP p{dy, x, mean, rstd, w, dres, dx, (bf16_t*)dx_bf16, dy_stride, x_stride, dx_stride, rows, D};
  const int key = (dy_dtype == VL_BF16 ? 4 : 0) | (x_dtype == VL_BF16 ? 2 : 0) | (g_dtype == VL_BF16 ? 1 : 0);
  switch (key) {
    case 0: ln_bwd_launch<float, float, float>(p, stream); break;
    case 1: ln_bwd_launch<float, float, bf16_t>(p, stream); break;
    case 2: ln_bwd_launch<float, bf16_t, float>(p, stream); break;
    case 3: ln_bwd_launch<float, bf16_t, bf16_t>(p, stream); break;
    case 4: ln_bwd_launch<bf16_t, float, float>(p, stream); break;
    case 5: ln_bwd_launch<bf16_t, float, bf16_t>(p, stream); break;
    case 6: ln_bwd_launch<bf16_t, bf16_t, float>(p, stream); break;
    default: ln_bwd_launch<bf16_t, bf16_t, bf16_t>(p, stream); break;
  }
  VL_HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int vl_layernorm_bwd(const void* dy, int dy_dtype, long dy_stride, const void* x, int x_dtype, long x_stride,
                                const float* mean, const float* rstd, const float* w, const float* dres, float* dx,
                                void* dx_bf16, long dx_stride, int rows, int D, hipStream_t stream) {
  return vl_layernorm_bwd_g(dy, dy_dtype, dy_stride, x, x_dtype, x_stride, mean, rstd, w, dres, dx, VL_F32, dx_bf16, dx_stride, rows, D, stream);
}

static const int kSlabRows = 256;
extern "C" long vl_colreduce_ws_floats(int rows, int cols, int planes) {
  return (long)((rows + kSlabRows - 1) / kSlabRows) * planes * cols;
}

extern "C" int vl_layernorm_bwd_params(const void* dy, int dy_dtype, long dy_stride, const void* x, int x_dtype, long x_stride,
                                       const float* mean, const float* rstd, float* dw, float* db, int rows, int D,
                                       float* ws, hipStream_t stream) {
  if (rows <= 0 || D <= 0) return vl_set_error("vl_layernorm_bwd_params: empty problem");
  if (!ws) return vl_set_error("vl_layernorm_bwd_params: workspace of vl_colreduce_ws_floats(rows, D, 2) floats required");
  const int slab = kSlabRows, nslab = (rows + slab - 1) / slab;
  const dim3 g((D + 255) / 256, nslab), b(256);
  if (dy_dtype == VL_BF16 && x_dtype == VL_F32 && (D & 3) == 0 && (dy_stride & 3) == 0 && (x_stride & 3) == 0 &&
      ((uintptr_t)dy & 7) == 0 && ((uintptr_t)x & 15) == 0)
    hipLaunchKernelGGL(ln_bwd_params4_bf16_f32_kernel, dim3((D + 255) / 256, nslab), b, 0, stream, (const bf16_t*)dy, dy_stride, (const float*)x, x_stride, mean, rstd, ws, rows, D, slab);
  else if (dy_dtype == VL_BF16 && x_dtype == VL_F32)
    hipLaunchKernelGGL((ln_bwd_params_kernel<bf16_t, float>), g, b, 0, stream, (const bf16_t*)dy, dy_stride, (const float*)x, x_stride, mean, rstd, ws, rows, D, slab);
  else if (dy_dtype == VL_BF16 && (D & 7) == 0 && (dy_stride & 7) == 0 && (x_stride & 7) == 0 && (((uintptr_t)dy | (uintptr_t)x) & 15) == 0)
    hipLaunchKernelGGL(ln_bwd_params8_bf16_kernel, dim3((D + 511) / 512, nslab), b, 0, stream, (const bf16_t*)dy, dy_stride, (const bf16_t*)x, x_stride, mean, rstd, ws, rows, D, slab);
  else if (dy_dtype == VL_BF16)
    hipLaunchKernelGGL((ln_bwd_params_kernel<bf16_t, bf16_t>), g, b, 0, stream, (const bf16_t*)dy, dy_stride, (const bf16_t*)x, x_stride, mean, rstd, ws, rows, D, slab);
  else if (x_dtype == VL_F32)
    hipLaunchKernelGGL((ln_bwd_params_kernel<float, float>), g, b, 0, stream, (const float*)dy, dy_stride, (const float*)x, x_stride, mean, rstd, ws, rows, D, slab);
  else
    hipLaunchKernelGGL((ln_bwd_params_kernel<float, bf16_t>), g, b, 0, stream, (const float*)dy, dy_stride, (const bf16_t*)x, x_stride, mean, rstd, ws, rows, D, slab);
  hipLaunchKernelGGL(slab_finalize_kernel, dim3((D + 63) / 64, 2), dim3(1024), 0, stream, ws, nslab, 2, D, 1.0f, dw, db);
  VL_HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int vl_colsum(const void* a, int a_dtype, long lda, float* out, int rows, int cols, float scale, float* ws, hipStream_t stream) {
  if (rows <= 0 || cols <= 0) return vl_set_error("vl_colsum: empty problem");
  if (!ws) return vl_set_error("vl_colsum: workspace of vl_colreduce_ws_floats(rows, cols, 1) floats required");
  const int slab = kSlabRows, nslab = (rows + slab - 1) / slab;
  const dim3 g((cols + 255) / 256, nslab), b(256);
  if (a_dtype == VL_BF16 && (cols & 7) == 0 && (lda & 7) == 0 && ((uintptr_t)a & 15) == 0)
    hipLaunchKernelGGL(colsum8_bf16_kernel, dim3((cols + 511) / 512, nslab), b, 0, stream, (const bf16_t*)a, lda, ws, rows, cols, slab);
  else if (a_dtype == VL_BF16) hipLaunchKernelGGL(colsum_kernel<bf16_t>, g, b, 0, stream, (const bf16_t*)a, lda, ws, rows, cols, slab);
  else if (a_dtype == VL_F32 && (cols & 3) == 0 && (lda & 3) == 0 && ((uintptr_t)a & 15) == 0)
    hipLaunchKernelGGL(colsum4_f32_kernel, g, b, 0, stream, (const float*)a, lda, ws, rows, cols, slab);
  else hipLaunchKernelGGL(colsum_kernel<float>, g, b, 0, stream, (const float*)a, lda, ws, rows, cols, slab);
  hipLaunchKernelGGL(slab_finalize_kernel, dim3((cols + 63) / 64, 1), dim3(1024), 0, stream, ws, nslab, 1, cols, scale, out, out);
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

extern "C" int vl_scale_exp_f32(const float* x, float* out, long n, const float* log_scale, float mul, hipStream_t stream) {
  if (n <= 0) return 0;
  if (!x || !out || !log_scale) return vl_set_error("vl_scale_exp_f32: null argument");
  hipLaunchKernelGGL(scale_exp_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, out, n, log_scale, mul);
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
