// Column-statistics kernels for the trainable point-cloud tokenizer (PointBERT mini-PointNet,
// open_clip/modal_3d/models/pointbert/dvae.py:179-212): nn.BatchNorm1d over the [B*G*M, C] activation matrix
// (channels = columns, batch = rows), forward (train-mode batch statistics or eval-mode running statistics) and
// backward, plus the backward of the per-group max and the per-group sum.  All HBM-bound: each kernel streams
// the matrix once with 16-byte (bf16x8) accesses; column reductions are two-stage and deterministic (row-chunk
// partials in a caller-provided workspace, then one finalize launch - no atomics).
//   vl_bn_stats        mean/var per column (biased var, as F.batch_norm normalises) + running-stat update
//   vl_bn_apply        y = gamma*(x-mean)*rstd + beta, optional ReLU, bf16 out
//   vl_bn_bwd          dgamma += sum dy'*xhat, dbeta += sum dy'; dx = gamma*rstd*(dy' - [train](mean(dy') + xhat*mean(dy'*xhat)))
//                      with dy' = dy * (y > 0) when the ReLU is fused
//   vl_bn_stats_local / vl_bn_stats_merge / vl_bn_bwd_reduce / vl_bn_bwd_apply
//                      the same two passes split at the point where SyncBatchNorm exchanges data between ranks
//                      (torch.nn.SyncBatchNorm, enabled by --use-bn-sync: training/point_cloud/pc_tri_main.py:372-373):
//                      forward = all-gather of per-rank (mean, M2, count) then a Chan merge in rank order; backward =
//                      all-reduce of (sum dy', sum dy'*xhat) then the elementwise pass with the global count
//   vl_group_max_bwd   df = base + one_hot(argmax over the M rows of a group) * dg      (torch.max(dim) backward)
//   vl_group_sum       out[g,:] = sum over the M rows of group g                          (backward of the expand)
#include "vl_common.h"
#include "vitlens_hip.h"

extern "C" int vl_set_error(const char* msg);
#define VL_HIP_OK(e) do { hipError_t _e = (e); if (_e != hipSuccess) return vl_set_error(hipGetErrorString(_e)); } while (0)

namespace {

// ------------------------------------------------------------------------------------------------ statistics
// block = 4 waves; a wave owns rows r0+w, r0+w+4, ...; lane owns 2 adjacent columns (one 4-byte load).
// Sums are taken about the first row (shift) so that E[d^2]-E[d]^2 does not cancel for columns with |mean| >> std.
__global__ void __launch_bounds__(256) bn_partial_kernel(const bf16_t* __restrict__ x, long ldx, int R, int C, int rows_per_chunk,
                                                         float* __restrict__ ws) {
  __shared__ float red[4][2][128];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 128 + lane * 2;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
  float s1a = 0.f, s1b = 0.f, s2a = 0.f, s2b = 0.f;
  if (c < C) {
    const unsigned int x0 = *(const unsigned int*)(x + c);
    const float sa = bf2f((bf16_t)(x0 & 0xffff)), sb = bf2f((bf16_t)(x0 >> 16));
    for (int r = r0 + w; r < r1; r += 4) {
      const unsigned int v = *(const unsigned int*)(x + (long)r * ldx + c);
      const float a = bf2f((bf16_t)(v & 0xffff)) - sa, b = bf2f((bf16_t)(v >> 16)) - sb;
      s1a += a; s1b += b; s2a = fmaf(a, a, s2a); s2b = fmaf(b, b, s2b);
    }
  }
  red[w][0][lane * 2] = s1a; red[w][0][lane * 2 + 1] = s1b;
  red[w][1][lane * 2] = s2a; red[w][1][lane * 2 + 1] = s2b;
  __syncthreads();
  if (threadIdx.x < 128) {
    const int cc = blockIdx.x * 128 + threadIdx.x;
    if (cc < C) {
      const int t = threadIdx.x;
      ws[((long)blockIdx.y * 2 + 0) * C + cc] = red[0][0][t] + red[1][0][t] + red[2][0][t] + red[3][0][t];
      ws[((long)blockIdx.y * 2 + 1) * C + cc] = red[0][1][t] + red[1][1][t] + red[2][1][t] + red[3][1][t];
    }
  }
}

__global__ void __launch_bounds__(256) bn_stats_finalize_kernel(const bf16_t* __restrict__ x, const float* __restrict__ ws, int nchunk,
                                                                int R, int C, float* mean, float* var, float* rmean, float* rvar,
                                                                float momentum) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int k = 0; k < nchunk; ++k) { s1 += ws[((long)k * 2) * C + c]; s2 += ws[((long)k * 2 + 1) * C + c]; }
  const double m = s1 / R;
  const double v = fmax(s2 / R - m * m, 0.0);
  const float mu = (float)(m + (double)bf2f(x[c]));
  mean[c] = mu; var[c] = (float)v;
  if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * mu;
  if (rvar) rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)(R > 1 ? v * R / (R - 1) : v);   // unbiased, as nn.BatchNorm1d
}

// per-rank statistics for SyncBN: local[c] = mean, local[C+c] = M2 = sum (x - mean)^2, local[2C] = row count (int bits)
__global__ void __launch_bounds__(256) bn_local_finalize_kernel(const bf16_t* __restrict__ x, const float* __restrict__ ws, int nchunk,
                                                                int R, int C, float* __restrict__ local) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c == 0) local[2 * C] = __builtin_bit_cast(float, R);
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int k = 0; k < nchunk; ++k) { s1 += ws[((long)k * 2) * C + c]; s2 += ws[((long)k * 2 + 1) * C + c]; }
  const double m = s1 / R;
  local[c] = (float)(m + (double)bf2f(x[c]));
  local[C + c] = (float)fmax(s2 - s1 * m, 0.0);
}

// Chan et al. pairwise merge of the ranks' (count, mean, M2) in rank order: every rank computes the same bits
__global__ void __launch_bounds__(256) bn_merge_kernel(const float* __restrict__ gathered, int W, int C, float* mean, float* var,
                                                       float* rmean, float* rvar, float momentum, int* total) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const long ld = 2L * C + 1;
  double n = 0.0, mu = 0.0, m2 = 0.0;
  for (int r = 0; r < W; ++r) {
    const float* g = gathered + r * ld;
    const double nk = (double)__builtin_bit_cast(int, g[2 * C]);
    if (nk <= 0.0) continue;
    const double d = (double)g[c] - mu, nn = n + nk;
    mu += d * (nk / nn);
    m2 += (double)g[C + c] + d * d * (n * nk / nn);
    n = nn;
  }
  const double v = n > 0.0 ? m2 / n : 0.0;
  mean[c] = (float)mu; var[c] = (float)v;
  if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mu;
  if (rvar) rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)(n > 1.0 ? m2 / (n - 1.0) : v);
  if (c == 0 && total) *total = (int)n;
}

// ------------------------------------------------------------------------------------------------ apply
__global__ void __launch_bounds__(256) bn_apply_kernel(const bf16_t* __restrict__ x, long ldx, const float* __restrict__ mean,
                                                       const float* __restrict__ var, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, int relu, bf16_t* __restrict__ out,
                                                       long ldo, long R, int C) {
  const int c8 = C >> 3;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= R * c8) return;
  const long r = i / c8; const int c = (int)(i - r * c8) * 8;
  const u32x4 v = *(const u32x4*)(x + r * ldx + c);
  u32x4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float y[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int cc = c + 2 * k + h;
      const float xv = bf2f((bf16_t)(h ? (v[k] >> 16) : (v[k] & 0xffff)));
      const float s = gamma[cc] * __builtin_amdgcn_rsqf(var[cc] + eps);
      float t = fmaf(xv - mean[cc], s, beta[cc]);
      if (relu) t = fmaxf(t, 0.f);
      y[h] = t;
    }
    o[k] = pack2bf(y[0], y[1]);
  }
  *(u32x4*)(out + r * ldo + c) = o;
}

// Column-stationary form of the same arithmetic (C/8 a divisor of 256): a thread keeps the coefficients of its 8 channels in
// registers and walks down the rows (round 3: the thread-per-element form re-read 4-6 per-channel parameters per element
// through L1 - 48 loads for 32 bytes of payload - and ran at an eighth of the HBM rate: 6.7 ms per call on the C5 step's
// [2.1 M, 512] activations, profiles/r03_bench_c5_kernel_stats.csv).
__global__ void __launch_bounds__(256) bn_apply_rows_kernel(const bf16_t* __restrict__ x, long ldx, const float* __restrict__ mean,
                                                            const float* __restrict__ var, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps, int relu, bf16_t* __restrict__ out,
                                                            long ldo, long R, int C) {
  const int tpr = C >> 3, rpb = 256 / tpr;
  const int c = (threadIdx.x % tpr) * 8;
  float s[8], mu[8], b[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s[e] = gamma[c + e] * __builtin_amdgcn_rsqf(var[c + e] + eps); mu[e] = mean[c + e]; b[e] = beta[c + e]; }
  for (long r = (long)blockIdx.x * rpb + threadIdx.x / tpr; r < R; r += (long)gridDim.x * rpb) {
    const u32x4 v = *(const u32x4*)(x + r * ldx + c);
    u32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float y[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float xv = bf2f((bf16_t)(h ? (v[k] >> 16) : (v[k] & 0xffff)));
        float t = fmaf(xv - mu[2 * k + h], s[2 * k + h], b[2 * k + h]);
        if (relu) t = fmaxf(t, 0.f);
        y[h] = t;
      }
      o[k] = pack2bf(y[0], y[1]);
    }
    *(u32x4*)(out + r * ldo + c) = o;
  }
}

// ------------------------------------------------------------------------------------------------ backward
struct BnBwdP {
  const bf16_t* dy; long lddy; const bf16_t* x; long ldx;
  const float* mean; const float* var; const float* gamma; const float* beta;
  float eps; int relu, train;
  float* ws; int nchunk, rows_per_chunk;
  float* dgamma; float* dbeta;
  bf16_t* dx; long lddx; int R, C;
  const float* fin;        // [2C]: what the elementwise pass subtracts - the two means, or (SyncBN) the two global sums
  const int* total;        // SyncBN: device pointer to the global row count (fin holds sums), else NULL
  float* sums;             // SyncBN reduce pass: raw local sums out [2C], else NULL
};

__global__ void __launch_bounds__(256) bn_bwd_partial_kernel(const BnBwdP p) {
  __shared__ float red[4][2][128];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 128 + lane * 2;
  const int r0 = blockIdx.y * p.rows_per_chunk, r1 = min(p.R, r0 + p.rows_per_chunk);
  float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
  if (c < p.C) {
    float mu[2], rs[2], g[2], b[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) { mu[h] = p.mean[c + h]; rs[h] = __builtin_amdgcn_rsqf(p.var[c + h] + p.eps); g[h] = p.gamma[c + h]; b[h] = p.beta[c + h]; }
    for (int r = r0 + w; r < r1; r += 4) {
      const unsigned int xv = *(const unsigned int*)(p.x + (long)r * p.ldx + c);
      const unsigned int dv = *(const unsigned int*)(p.dy + (long)r * p.lddy + c);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float xh = (bf2f((bf16_t)(h ? (xv >> 16) : (xv & 0xffff))) - mu[h]) * rs[h];
        float d = bf2f((bf16_t)(h ? (dv >> 16) : (dv & 0xffff)));
        if (p.relu && fmaf(xh, g[h], b[h]) <= 0.f) d = 0.f;
        s1[h] += d; s2[h] = fmaf(d, xh, s2[h]);
      }
    }
  }
  red[w][0][lane * 2] = s1[0]; red[w][0][lane * 2 + 1] = s1[1];
  red[w][1][lane * 2] = s2[0]; red[w][1][lane * 2 + 1] = s2[1];
  __syncthreads();
  if (threadIdx.x < 128) {
    const int cc = blockIdx.x * 128 + threadIdx.x;
    if (cc < p.C) {
      const int t = threadIdx.x;
      p.ws[((long)blockIdx.y * 2 + 0) * p.C + cc] = red[0][0][t] + red[1][0][t] + red[2][0][t] + red[3][0][t];
      p.ws[((long)blockIdx.y * 2 + 1) * p.C + cc] = red[0][1][t] + red[1][1][t] + red[2][1][t] + red[3][1][t];
    }
  }
}

// sums -> parameter gradients (accumulated) and the two per-column means the apply pass needs (stored after the partials)
__global__ void __launch_bounds__(256) bn_bwd_finalize_kernel(const BnBwdP p) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= p.C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int k = 0; k < p.nchunk; ++k) { s1 += p.ws[((long)k * 2) * p.C + c]; s2 += p.ws[((long)k * 2 + 1) * p.C + c]; }
  if (p.dbeta) p.dbeta[c] += (float)s1;
  if (p.dgamma) p.dgamma[c] += (float)s2;
  if (p.sums) { p.sums[c] = (float)s1; p.sums[p.C + c] = (float)s2; return; }
  float* fin = p.ws + (long)p.nchunk * 2 * p.C;
  fin[c] = p.train ? (float)(s1 / p.R) : 0.f;
  fin[p.C + c] = p.train ? (float)(s2 / p.R) : 0.f;
}

__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const BnBwdP p) {
  const int c8 = p.C >> 3;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)p.R * c8) return;
  const long r = i / c8; const int c = (int)(i - r * c8) * 8;
  const float* fin = p.fin;
  const float inv_n = p.total ? 1.0f / (float)*p.total : 1.0f;
  const u32x4 xv = *(const u32x4*)(p.x + r * p.ldx + c);
  const u32x4 dv = *(const u32x4*)(p.dy + r * p.lddy + c);
  u32x4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float y[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int cc = c + 2 * k + h;
      const float rs = __builtin_amdgcn_rsqf(p.var[cc] + p.eps), g = p.gamma[cc];
      const float xh = (bf2f((bf16_t)(h ? (xv[k] >> 16) : (xv[k] & 0xffff))) - p.mean[cc]) * rs;
      float d = bf2f((bf16_t)(h ? (dv[k] >> 16) : (dv[k] & 0xffff)));
      if (p.relu && fmaf(xh, g, p.beta[cc]) <= 0.f) d = 0.f;
      y[h] = g * rs * (d - fin[cc] * inv_n - xh * (fin[p.C + cc] * inv_n));
    }
    o[k] = pack2bf(y[0], y[1]);
  }
  *(u32x4*)(p.dx + r * p.lddx + c) = o;
}

// column-stationary form (see bn_apply_rows_kernel)
__global__ void __launch_bounds__(256) bn_bwd_apply_rows_kernel(const BnBwdP p) {
  const int tpr = p.C >> 3, rpb = 256 / tpr;
  const int c = (threadIdx.x % tpr) * 8;
  const float inv_n = p.total ? 1.0f / (float)*p.total : 1.0f;
  float rs[8], mu[8], g[8], b[8], f1[8], f2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    rs[e] = __builtin_amdgcn_rsqf(p.var[c + e] + p.eps); mu[e] = p.mean[c + e]; g[e] = p.gamma[c + e]; b[e] = p.beta[c + e];
    f1[e] = p.fin[c + e] * inv_n; f2[e] = p.fin[p.C + c + e] * inv_n;
  }
  for (long r = (long)blockIdx.x * rpb + threadIdx.x / tpr; r < p.R; r += (long)gridDim.x * rpb) {
    const u32x4 xv = *(const u32x4*)(p.x + r * p.ldx + c);
    const u32x4 dv = *(const u32x4*)(p.dy + r * p.lddy + c);
    u32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float y[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int e = 2 * k + h;
        const float xh = (bf2f((bf16_t)(h ? (xv[k] >> 16) : (xv[k] & 0xffff))) - mu[e]) * rs[e];
        float d = bf2f((bf16_t)(h ? (dv[k] >> 16) : (dv[k] & 0xffff)));
        if (p.relu && fmaf(xh, g[e], b[e]) <= 0.f) d = 0.f;
        y[h] = g[e] * rs[e] * (d - f1[e] - xh * f2[e]);
      }
      o[k] = pack2bf(y[0], y[1]);
    }
    *(u32x4*)(p.dx + r * p.lddx + c) = o;
  }
}

// ------------------------------------------------------------------------------------------------ group ops
// thread per (group, column pair); lanes run along columns so every row access is coalesced
__global__ void __launch_bounds__(256) group_max_bwd_kernel(const bf16_t* __restrict__ f, long ldf, const bf16_t* __restrict__ dg, long lddg,
                                                            const bf16_t* __restrict__ base, long ldb, bf16_t* __restrict__ out, long ldo,
                                                            long groups, int M, int C) {
  const int c2 = C >> 1;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= groups * c2) return;
  const long g = i / c2; const int c = (int)(i - g * c2) * 2;
  const bf16_t* fp = f + g * M * ldf + c;
  float best[2] = {-INFINITY, -INFINITY}; int arg[2] = {0, 0};
  for (int m = 0; m < M; ++m) {
    const unsigned int v = *(const unsigned int*)(fp + (long)m * ldf);
    const float a = bf2f((bf16_t)(v & 0xffff)), b = bf2f((bf16_t)(v >> 16));
    if (a > best[0]) { best[0] = a; arg[0] = m; }
    if (b > best[1]) { best[1] = b; arg[1] = m; }
  }
  const unsigned int dv = *(const unsigned int*)(dg + g * lddg + c);
  const float d0 = bf2f((bf16_t)(dv & 0xffff)), d1 = bf2f((bf16_t)(dv >> 16));
  for (int m = 0; m < M; ++m) {
    float y0 = 0.f, y1 = 0.f;
    if (base) {
      const unsigned int bv = *(const unsigned int*)(base + (g * M + m) * ldb + c);
      y0 = bf2f((bf16_t)(bv & 0xffff)); y1 = bf2f((bf16_t)(bv >> 16));
    }
    if (m == arg[0]) y0 += d0;
    if (m == arg[1]) y1 += d1;
    *(unsigned int*)(out + (g * M + m) * ldo + c) = pack2bf(y0, y1);
  }
}

__global__ void __launch_bounds__(256) group_sum_kernel(const bf16_t* __restrict__ x, long ldx, bf16_t* __restrict__ out, long ldo,
                                                        long groups, int M, int C) {
  const int c2 = C >> 1;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= groups * c2) return;
  const long g = i / c2; const int c = (int)(i - g * c2) * 2;
  float s0 = 0.f, s1 = 0.f;
  for (int m = 0; m < M; ++m) {
    const unsigned int v = *(const unsigned int*)(x + (g * M + m) * ldx + c);
    s0 += bf2f((bf16_t)(v & 0xffff)); s1 += bf2f((bf16_t)(v >> 16));
  }
  *(unsigned int*)(out + g * ldo + c) = pack2bf(s0, s1);
}

inline unsigned grid1(long n) { return (unsigned)((n + 255) / 256); }
// the column-stationary apply kernels: C/8 threads per row must tile a 256-thread block
inline bool rows_form_ok(int C) { const int tpr = C >> 3; return tpr >= 1 && tpr <= 256 && 256 % tpr == 0; }
inline unsigned rows_grid(long R, int C) {
  const long rpb = 256 / (C >> 3), blocks = (R + rpb - 1) / rpb;
  return (unsigned)(blocks < 4096 ? blocks : 4096);          // 16 blocks per CU, each walking down the rows
}
inline void launch_bwd_apply(const BnBwdP& p, hipStream_t stream) {
  if (rows_form_ok(p.C)) hipLaunchKernelGGL(bn_bwd_apply_rows_kernel, dim3(rows_grid(p.R, p.C)), dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid1((long)p.R * (p.C >> 3))), dim3(256), 0, stream, p);
}

inline bool ws_shape_ok(int R, int nchunk) { return nchunk >= 1 && nchunk <= 65535 && R >= 1; }

}  // namespace

extern "C" int vl_bn_stats(const void* x, long ldx, int R, int C, float* ws, int nchunk, float* mean, float* var,
                           float* running_mean, float* running_var, float momentum, hipStream_t stream) {
  if (!ws_shape_ok(R, nchunk) || C <= 0 || (C & 1) || (ldx & 1)) return vl_set_error("vl_bn_stats: need R>=1, even C and ldx, 1<=nchunk<=65535");
  const int rpc = (R + nchunk - 1) / nchunk;
  hipLaunchKernelGGL(bn_partial_kernel, dim3((C + 127) / 128, nchunk), dim3(256), 0, stream, (const bf16_t*)x, ldx, R, C, rpc, ws);
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, (const bf16_t*)x, ws, nchunk, R, C, mean, var,
                     running_mean, running_var, momentum);
  VL_HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int vl_bn_apply(const void* x, long ldx, const float* mean, const float* var, const float* gamma, const float* beta,
                           float eps, int relu, void* out, long ldo, long R, int C, hipStream_t stream) {
  if (R <= 0 || C <= 0 || (C & 7) || (ldx & 7) || (ldo & 7)) return vl_set_error("vl_bn_apply: C, ldx, ldo must be multiples of 8");
  if (rows_form_ok(C))
    hipLaunchKernelGGL(bn_apply_rows_kernel, dim3(rows_grid(R, C)), dim3(256), 0, stream, (const bf16_t*)x, ldx, mean, var, gamma, beta,
                       eps, relu, (bf16_t*)out, ldo, R, C);
  else
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid1(R * (C >> 3))), dim3(256), 0, stream, (const bf16_t*)x, ldx, mean, var, gamma, beta, eps,
                       relu, (bf16_t*)out, ldo, R, C);
  VL_HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int vl_bn_bwd(const void* dy, long lddy, const void* x, long ldx, const float* mean, const float* var, const float* gamma,
                         const float* beta, float eps, int relu, int train, float* ws, int nchunk, float* dgamma, float* dbeta,
                         void* dx, long lddx, int R, int C, hipStream_t stream) {
  if (!ws_shape_ok(R, nchunk) || C <= 0 || (C & 7) || (ldx & 7) || (lddy & 7) || (lddx & 7))
    return vl_set_error("vl_bn_bwd: C and strides must be multiples of 8, 1<=nchunk<=65535");
  BnBwdP p{(const bf16_t*)dy, lddy, (const bf16_t*)x, ldx, mean, var, gamma, beta, eps, relu, train, ws, nchunk,
           (R + nchunk - 1) / nchunk, dgamma, dbeta, (bf16_t*)dx, lddx, R, C, ws + (long)nchunk * 2 * C, nullptr, nullptr};
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3((C + 127) / 128, nchunk), dim3(256), 0, stream, p);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, p);
  if (dx) launch_bwd_apply(p, stream);
  VL_HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int vl_bn_stats_local(const void* x, long ldx, int R, int C, float* ws, int nchunk, float* local, hipStream_t stream) {
  if (!ws_shape_ok(R, nchunk) || C <= 0 || (C & 1) || (ldx & 1)) return vl_set_error("vl_bn_stats_local: need R>=1, even C and ldx, 1<=nchunk<=65535");
  const int rpc = (R + nchunk - 1) / nchunk;
  hipLaunchKernelGGL(bn_partial_kernel, dim3((C + 127) / 128, nchunk), dim3(256), 0, stream, (const bf16_t*)x, ldx, R, C, rpc, ws);
  hipLaunchKernelGGL(bn_local_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, (const bf16_t*)x, ws, nchunk, R, C, local);
  VL_HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int vl_bn_stats_merge(const float* gathered, int W, int C, float* mean, float* var, float* running_mean,
                                 float* running_var, float momentum, int* total, hipStream_t stream) {
  if (W <= 0 || C <= 0) return vl_set_error("vl_bn_stats_merge: need W >= 1 and C >= 1");
  hipLaunchKernelGGL(bn_merge_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, gathered, W, C, mean, var, running_mean, running_var,
                     momentum, total);
  VL_HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int vl_bn_bwd_reduce(const void* dy, long lddy, const void* x, long ldx, const float* mean, const float* var,
                                const float* gamma, const float* beta, float eps, int relu, float* ws, int nchunk, float* dgamma,
                                float* dbeta, float* sums, int R, int C, hipStream_t stream) {
  if (!ws_shape_ok(R, nchunk) || C <= 0 || (C & 7) || (ldx & 7) || (lddy & 7) || !sums)
    return vl_set_error("vl_bn_bwd_reduce: C and strides must be multiples of 8, 1<=nchunk<=65535, sums required");
  BnBwdP p{(const bf16_t*)dy, lddy, (const bf16_t*)x, ldx, mean, var, gamma, beta, eps, relu, 1, ws, nchunk,
           (R + nchunk - 1) / nchunk, dgamma, dbeta, nullptr, 0, R, C, nullptr, nullptr, sums};
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3((C + 127) / 128, nchunk), dim3(256), 0, stream, p);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, p);
  VL_HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int vl_bn_bwd_apply(const void* dy, long lddy, const void* x, long ldx, const float* mean, const float* var,
                               const float* gamma, const float* beta, float eps, int relu, const float* sums, const int* total,
                               void* dx, long lddx, int R, int C, hipStream_t stream) {
  if (R <= 0 || C <= 0 || (C & 7) || (ldx & 7) || (lddy & 7) || (lddx & 7) || !sums || !total || !dx)
    return vl_set_error("vl_bn_bwd_apply: C and strides must be multiples of 8; sums, total and dx required");
  BnBwdP p{(const bf16_t*)dy, lddy, (const bf16_t*)x, ldx, mean, var, gamma, beta, eps, relu, 1, nullptr, 0, 0, nullptr, nullptr,
           (bf16_t*)dx, lddx, R, C, sums, total, nullptr};
  launch_bwd_apply(p, stream);
  VL_HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int vl_group_max_bwd(const void* f, long ldf, const void* dg, long lddg, const void* base, long ldb, void* out, long ldo,
                                long groups, int M, int C, hipStream_t stream) {
  if (groups <= 0 || M <= 0 || C <= 0 || (C & 1) || (ldf & 1) || (lddg & 1) || (ldo & 1) || (base && (ldb & 1)))
    return vl_set_error("vl_group_max_bwd: even C and strides required");
  hipLaunchKernelGGL(group_max_bwd_kernel, dim3(grid1(groups * (C >> 1))), dim3(256), 0, stream, (const bf16_t*)f, ldf, (const bf16_t*)dg,
                     lddg, (const bf16_t*)base, ldb, (bf16_t*)out, ldo, groups, M, C);
  VL_HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int vl_group_sum(const void* x, long ldx, void* out, long ldo, long groups, int M, int C, hipStream_t stream) {
  if (groups <= 0 || M <= 0 || C <= 0 || (C & 1) || (ldx & 1) || (ldo & 1)) return vl_set_error("vl_group_sum: even C and strides required");
  hipLaunchKernelGGL(group_sum_kernel, dim3(grid1(groups * (C >> 1))), dim3(256), 0, stream, (const bf16_t*)x, ldx, (bf16_t*)out, ldo, groups, M, C);
  VL_HIP_OK(hipGetLastError());
  return 0;
}
