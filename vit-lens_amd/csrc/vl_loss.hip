// InfoNCE (CLIP contrastive) loss pieces on gfx950.
//
// Replaces F.cross_entropy over rows and columns of `logit_scale * x @ y.T` and its autograd
// (open_clip/loss.py:129-136,158-163,300-306,377-383).  The logits matrix is produced once
// by vl_gemm_bf16 (fp32 out); these kernels read it ONCE for the row statistics, once for the
// column statistics and once to emit the gradient matrix G = dL/dlogits (bf16) together with
// its transpose, so the two feature-gradient GEMMs run as plain NT GEMMs.
//   labels: row r <-> column (r + label_off)     (label_off = rank*b under --local-loss)
#include "vl_common.h"
#include "vitlens_hip.h"

namespace {

// one wave per row: online (max, sum) -> lse[r]; diag[r] = logits[r, r+off]
__global__ void __launch_bounds__(256) row_lse_kernel(const float* lg, long ld, int R, int C, int off,
                                                      float* lse, float* diag) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const float* row = lg + (long)r * ld;
  float m = -INFINITY, s = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float v = row[c];
    const float mn = fmaxf(m, v);
    s = s * __expf(m - mn) + __expf(v - mn);
    m = mn;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
    const float mn = fmaxf(m, m2);
    const float a = (m == -INFINITY) ? 0.f : s * __expf(m - mn);
    const float b = (m2 == -INFINITY) ? 0.f : s2 * __expf(m2 - mn);
    s = a + b; m = mn;
  }
  if (lane == 0) {
    lse[r] = m + __logf(s);
    if (diag) { const int c = r + off; diag[r] = (c >= 0 && c < C) ? row[c] : 0.f; }
  }
}

// column partials over a chunk of 64 rows: thread per column (coalesced)
__global__ void __launch_bounds__(256) col_part_kernel(const float* lg, long ld, int R, int C,
                                                       float* pm, float* ps) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const int r0 = blockIdx.y * 64, r1 = min(R, r0 + 64);
  float m = -INFINITY, s = 0.f;
  for (int r = r0; r < r1; ++r) {
    const float v = lg[(long)r * ld + c];
    const float mn = fmaxf(m, v);
    s = s * __expf(m - mn) + __expf(v - mn);
    m = mn;
  }
  pm[(long)blockIdx.y * C + c] = m; ps[(long)blockIdx.y * C + c] = s;
}
__global__ void __launch_bounds__(256) col_comb_kernel(const float* pm, const float* ps, int nchunk, int C, float* lse) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float m = -INFINITY, s = 0.f;
  for (int k = 0; k < nchunk; ++k) {
    const float m2 = pm[(long)k * C + c], s2 = ps[(long)k * C + c];
    const float mn = fmaxf(m, m2);
    const float a = (m == -INFINITY) ? 0.f : s * __expf(m - mn);
    s = a + s2 * __expf(m2 - mn); m = mn;
  }
  lse[c] = m + __logf(s);
}

// loss_out[0] += w_row * mean_r(row_lse[r] - diag[r]) + w_col * mean_r(col_lse[r+off] - diag[r])
__global__ void __launch_bounds__(256) ce_reduce_kernel(const float* row_lse, const float* col_lse, const float* diag,
                                                        int R, int C, int off, float w_row, float w_col, float* loss_out) {
  float a = 0.f;
  for (int r = threadIdx.x; r < R; r += 256) {
    float v = 0.f;
    if (row_lse) v += w_row * (row_lse[r] - diag[r]);
    const int c = r + off;
    if (col_lse && c >= 0 && c < C) v += w_col * (col_lse[c] - diag[r]);
    a += v;
  }
  __shared__ float sh[4];
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss_out, (sh[0] + sh[1] + sh[2] + sh[3]) / (float)R);
}

// G[r,c] = w_row/R * (softmax_row - onehot) + w_col/R * (softmax_col - onehot); also G^T;
// part[block] = sum over the block of G*l (stage 1 of the deterministic d/dscale reduction: no fp32 atomics)
__global__ void __launch_bounds__(256) grad_kernel(const float* lg, long ld, int R, int C, int off,
                                                   const float* row_lse, const float* col_lse, float w_row, float w_col,
                                                   bf16_t* G, long ldg, bf16_t* GT, long ldgt, float* part) {
  __shared__ float tile[32][33];
  __shared__ float red[4];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const float invR = 1.0f / (float)R;
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + k * 8, c = c0 + tx;
    float g = 0.f;
    if (r < R && c < C) {
      const float l = lg[(long)r * ld + c];
      const float hot = (c == r + off) ? 1.f : 0.f;
      if (row_lse) g += w_row * invR * (__expf(l - row_lse[r]) - hot);
      if (col_lse) g += w_col * invR * (__expf(l - col_lse[c]) - hot);
      acc = fmaf(g, l, acc);
    }
    tile[ty + k * 8][tx] = g;
    if (G && r < R && c < ldg) G[(long)r * ldg + c] = f2bf(g);   // pad columns [C, ldg) get zeros
  }
  __syncthreads();
  if (GT) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c0 + ty + k * 8, r = r0 + tx;
      if (c < C && r < ldgt) GT[(long)c * ldgt + r] = f2bf(r < R ? tile[tx][ty + k * 8] : 0.f);
    }
  }
  if (part) {
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.y * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

// out[0] += scale * sum_i part[i]: one block, every thread a fixed strided subsequence, fixed-order tree on top
__global__ void __launch_bounds__(1024) part_finalize_kernel(const float* part, long n, float scale, float* out) {
  __shared__ float sh[16];
  float a = 0.f;
  for (long i = threadIdx.x; i < n; i += 1024) a += part[i];
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += sh[w];
    out[0] += t * scale;
  }
}

// backward of F.normalize: dx = (df - f * <f, df>) / max(||x||, eps)  (f = normalised feature)
__global__ void __launch_bounds__(256) l2norm_bwd_kernel(const float* f, const float* df, const float* nrm, float* dx,
                                                         int rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* fr = f + (long)row * D; const float* dr = df + (long)row * D;
  float s = 0.f;
  for (int e = lane; e < D; e += 64) s = fmaf(fr[e], dr[e], s);
  s = wave_sum(s);
  const float inv = 1.0f / fmaxf(nrm[row], eps);
  for (int e = lane; e < D; e += 64) dx[(long)row * D + e] = (dr[e] - fr[e] * s) * inv;
}

// out[c, r] (bf16, ld) = in[r, c] (f32 or bf16) ; pads columns r in [R, ld) with zeros.  Generic small-tile version
// (any R, C); the weight-gradient path uses transpose64_kernel below.
template <typename TIN>
__global__ void __launch_bounds__(256) transpose_bf16_kernel(const TIN* in, long ldi, int R, int C, bf16_t* out, long ldo) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + k * 8, c = c0 + tx;
    float v = 0.f;
    if (r < R && c < C) {
      if constexpr (sizeof(TIN) == 4) v = ((const float*)in)[(long)r * ldi + c];
      else v = bf2f(((const bf16_t*)in)[(long)r * ldi + c]);
    }
    tile[ty + k * 8][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + k * 8, r = r0 + tx;
    if (c < C && r < ldo) out[(long)c * ldo + r] = f2bf(tile[tx][ty + k * 8]);
  }
}

// Transpose for the weight-gradient GEMMs (dW = dY^T X needs both operands with the token axis contiguous), fused with
// the bias gradient: a block owns 64 columns x 256 rows (four 64x64 tiles), reads with 16-byte loads, goes through an
// XOR-swizzled 8 KB LDS tile (conflict-free both ways), writes whole 128-byte lines of the transposed matrix
// (8 lanes x 16 bytes), and - if `colws` is given - leaves the column sums of its 256 rows in colws[blockIdx.y][C]
// (stage 1 of the deterministic two-stage bias-gradient reduction; no separate pass over dY).
// Requires C % 64 == 0 and ldo % 8 == 0; rows beyond R read as zero (the pad columns of the output).
template <typename TIN>
__global__ void __launch_bounds__(256) transpose64_kernel(const TIN* in, long ldi, int R, int C, bf16_t* out, long ldo, float* colws) {
  __shared__ __attribute__((aligned(16))) bf16_t tile[64 * 64];
  const int c0 = blockIdx.x * 64;
  const int t = threadIdx.x;
  const int lrow = t >> 3, lch = t & 7;                 // load: row (0..31, +32), 16-byte chunk of 8 columns
  const int gch = t & 7, gc = t >> 3;                   // gather: row chunk (8 rows), column (0..31, +32)
  float csum[2] = {0.f, 0.f};
  for (int rt = 0; rt < 4; ++rt) {
    const int r0 = blockIdx.y * 256 + rt * 64;
    if (r0 >= ldo) break;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = lrow + 32 * h;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (r0 + r < R) {
        if constexpr (sizeof(TIN) == 2) {
          v = *(const u32x4*)((const bf16_t*)in + (long)(r0 + r) * ldi + c0 + lch * 8);
        } else {
          const f32x4 a = *(const f32x4*)((const float*)in + (long)(r0 + r) * ldi + c0 + lch * 8);
          const f32x4 b = *(const f32x4*)((const float*)in + (long)(r0 + r) * ldi + c0 + lch * 8 + 4);
          v[0] = pack2bf(a[0], a[1]); v[1] = pack2bf(a[2], a[3]); v[2] = pack2bf(b[0], b[1]); v[3] = pack2bf(b[2], b[3]);
        }
      }
      *(u32x4*)(tile + r * 64 + ((lch ^ ((r >> 3) & 7)) << 3)) = v;
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = gc + 32 * h;
      unsigned int w[4];
      float sacc = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = gch * 8 + j;
        const bf16_t e = tile[r * 64 + ((((c >> 3) ^ gch) << 3) | (c & 7))];       // (r >> 3) & 7 == gch
        sacc += bf2f(e);
        if (j & 1) w[j >> 1] |= ((unsigned int)e) << 16; else w[j >> 1] = e;
      }
      if (r0 + gch * 8 < ldo) {
        const u32x4 o = {w[0], w[1], w[2], w[3]};
        *(u32x4*)(out + (long)(c0 + c) * ldo + r0 + gch * 8) = o;
      }
      // sum over the 8 row chunks = the 8 lanes of this column
      sacc += __shfl_xor(sacc, 1, 64); sacc += __shfl_xor(sacc, 2, 64); sacc += __shfl_xor(sacc, 4, 64);
      csum[h] += sacc;
    }
    __syncthreads();
  }
  if (colws && gch == 0) {
    colws[(long)blockIdx.y * C + c0 + gc] = csum[0];
    colws[(long)blockIdx.y * C + c0 + gc + 32] = csum[1];
  }
}

// out[j] += scale * sum_s ws[s][j], fixed order, 8 independent partial sums in flight per thread
__global__ void __launch_bounds__(256) colws_finalize_kernel(const float* ws, int nslab, int C, float scale, float* out) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= C) return;
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int s = 0;
  for (; s + 8 <= nslab; s += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] += ws[(long)(s + u) * C + j];
  }
  for (; s < nslab; ++s) a[0] += ws[(long)s * C + j];
  out[j] += (((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]))) * scale;
}

// x f32 [R,D] -> bf16 [R,3D]: hi = bf16(x), lo = bf16(x - hi);  pattern 0: [hi|lo|hi], 1: [hi|hi|lo].
// A GEMM of a pattern-0 matrix against a pattern-1 matrix (K = 3D) yields xh.yh + xl.yh + xh.yl: the
// logits to ~2^-17 relative instead of 2^-9, at 3x the (negligible) FLOPs of the B x B x D product.
__global__ void __launch_bounds__(256) split3_kernel(const float* x, bf16_t* out, long rows, int D, int pattern) {
  const long n = rows * D;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long r = i / D; const int d = (int)(i - r * D);
    const float v = x[i];
    const bf16_t hi = f2bf(v);
    const bf16_t lo = f2bf(v - bf2f(hi));
    bf16_t* o = out + r * 3 * D;
    o[d] = hi;
    o[D + d] = pattern == 0 ? lo : hi;
    o[2 * D + d] = pattern == 0 ? hi : lo;
  }
}

}  // namespace

extern "C" int vl_set_error(const char* msg);
#define VL_HIP_OK(e) do { hipError_t _e = (e); if (_e != hipSuccess) return vl_set_error(hipGetErrorString(_e)); } while (0)

extern "C" int vl_ce_stats(const float* logits, long ld, int R, int C, int label_off, float* row_lse, float* col_lse,
                           float* diag, float* col_ws, hipStream_t stream) {
  if (R <= 0 || C <= 0) return vl_set_error("vl_ce_stats: empty problem");
  if (row_lse || diag) {
    if (!row_lse) return vl_set_error("vl_ce_stats: diag requires row_lse");
    hipLaunchKernelGGL(row_lse_kernel, dim3((R + 3) / 4), dim3(256), 0, stream, logits, ld, R, C, label_off, row_lse, diag);
  }
  if (col_lse) {
    if (!col_ws) return vl_set_error("vl_ce_stats: col_lse needs a workspace of 2*ceil(R/64)*C floats");
    const int nch = (R + 63) / 64;
    float* pm = col_ws; float* ps = col_ws + (long)nch * C;
    hipLaunchKernelGGL(col_part_kernel, dim3((C + 255) / 256, nch), dim3(256), 0, stream, logits, ld, R, C, pm, ps);
    hipLaunchKernelGGL(col_comb_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, pm, ps, nch, C, col_lse);
  }
  VL_HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int vl_ce_loss_accum(const float* row_lse, const float* col_lse, const float* diag, int R, int C, int label_off,
                                float w_row, float w_col, float* loss_inout, hipStream_t stream) {
  hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(256), 0, stream, row_lse, col_lse, diag, R, C, label_off, w_row, w_col, loss_inout);
  VL_HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int vl_ce_grad(const float* logits, long ld, int R, int C, int label_off, const float* row_lse, const float* col_lse,
                          float w_row, float w_col, void* G, long ldg, void* GT, long ldgt, float logit_scale,
                          float* dscale_inout, float* ws, hipStream_t stream) {
  if (R <= 0 || C <= 0) return vl_set_error("vl_ce_grad: empty problem");
  if (dscale_inout && !ws) return vl_set_error("vl_ce_grad: d/dscale needs a workspace of vl_ce_grad_ws_floats(R, C, ldg, ldgt) floats");
  const int gc = (int)(((G && ldg > C ? ldg : C) + 31) / 32), gr = (int)(((GT && ldgt > R ? ldgt : R) + 31) / 32);
  hipLaunchKernelGGL(grad_kernel, dim3(gc, gr), dim3(256), 0, stream, logits, ld, R, C, label_off, row_lse, col_lse, w_row, w_col,
                     (bf16_t*)G, ldg, (bf16_t*)GT, ldgt, dscale_inout ? ws : nullptr);
  if (dscale_inout)
    hipLaunchKernelGGL(part_finalize_kernel, dim3(1), dim3(1024), 0, stream, ws, (long)gc * gr, 1.0f / logit_scale, dscale_inout);
  VL_HIP_OK(hipGetLastError());
  return 0;
}

extern "C" long vl_ce_grad_ws_floats(int R, int C, long ldg, long ldgt) {
  const long gc = ((ldg > C ? ldg : C) + 31) / 32, gr = ((ldgt > R ? ldgt : R) + 31) / 32;
  return gc * gr;
}

extern "C" int vl_l2_normalize_bwd(const float* f, const float* df, const float* norms, float* dx, int rows, int D, float eps,
                                   hipStream_t stream) {
  if (rows <= 0) return vl_set_error("vl_l2_normalize_bwd: empty problem");
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, f, df, norms, dx, rows, D, eps);
  VL_HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int vl_transpose_colsum_bf16(const void* in, int in_dtype, long ldi, int R, int C, void* out, long ldo,
                                        float* colsum_out, float colsum_scale, float* ws, hipStream_t stream) {
  if (R <= 0 || C <= 0) return vl_set_error("vl_transpose_colsum_bf16: empty problem");
  if (ldo < R || (ldo & 7) || (C & 63) || (ldi & 7)) return vl_set_error("vl_transpose_colsum_bf16: need ldo >= R, ldo % 8 == 0, C % 64 == 0, ldi % 8 == 0");
  if (colsum_out && !ws) return vl_set_error("vl_transpose_colsum_bf16: colsum needs a workspace of ceil(ldo/256)*C floats");
  const int nslab = (int)((ldo + 255) / 256);
  const dim3 g(C / 64, nslab);
  float* cw = colsum_out ? ws : nullptr;
  if (in_dtype == VL_F32) hipLaunchKernelGGL(transpose64_kernel<float>, g, dim3(256), 0, stream, (const float*)in, ldi, R, C, (bf16_t*)out, ldo, cw);
  else hipLaunchKernelGGL(transpose64_kernel<bf16_t>, g, dim3(256), 0, stream, (const bf16_t*)in, ldi, R, C, (bf16_t*)out, ldo, cw);
  if (colsum_out) hipLaunchKernelGGL(colws_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, ws, nslab, C, colsum_scale, colsum_out);
  VL_HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int vl_transpose_to_bf16(const void* in, int in_dtype, long ldi, int R, int C, void* out, long ldo, hipStream_t stream) {
  if (R <= 0 || C <= 0) return vl_set_error("vl_transpose_to_bf16: empty problem");
  if (ldo < R) return vl_set_error("vl_transpose_to_bf16: ldo < R");
  if (R >= 256 && !(ldo & 7) && !(C & 63) && !(ldi & 7) && !(((uintptr_t)in | (uintptr_t)out) & 15))
    return vl_transpose_colsum_bf16(in, in_dtype, ldi, R, C, out, ldo, nullptr, 0.f, nullptr, stream);
  const dim3 g((C + 31) / 32, (int)((ldo + 31) / 32));
  if (in_dtype == VL_F32) hipLaunchKernelGGL(transpose_bf16_kernel<float>, g, dim3(256), 0, stream, (const float*)in, ldi, R, C, (bf16_t*)out, ldo);
  else hipLaunchKernelGGL(transpose_bf16_kernel<bf16_t>, g, dim3(256), 0, stream, (const bf16_t*)in, ldi, R, C, (bf16_t*)out, ldo);
  VL_HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int vl_split_bf16x3(const float* x, void* out, long rows, int D, int pattern, hipStream_t stream) {
  if (rows <= 0 || D <= 0) return vl_set_error("vl_split_bf16x3: empty problem");
  long g = (rows * D + 255) / 256; if (g > 4096) g = 4096;
  hipLaunchKernelGGL(split3_kernel, dim3((int)g), dim3(256), 0, stream, x, (bf16_t*)out, rows, D, pattern);
  VL_HIP_OK(hipGetLastError());
  return 0;
}
