// Row-wise (HBM-bound) kernels: LayerNorm forward (+ fused token assembly / gather), L2 normalise,
// patch extraction (im2col), text embedding.  One 64-lane wave per row, 16-byte vector accesses,
// wave-level reductions only (no LDS, no barriers).
//
// Reference ops replaced: LayerNorm / LayerNormFp32 (open_clip/transformer.py:17-34), the cls
// concat + positional add + ln_pre of VisionTransformer.forward (:756-772), x[:,0] -> ln_post
// (:653-657,783-785), ln_final + EOT gather (open_clip/model.py:537-539), F.normalize (:522),
// conv1 patchify (transformer.py:464-470,674-676; DepthTokenizer.py:22-28; AST_tokenizer.py:22-50),
// token_embedding + positional_embedding (model.py:531-533).
#include "vl_common.h"
#include "vitlens_hip.h"

namespace {

__device__ __forceinline__ void load4(const float* p, float (&v)[4]) {
  const f32x4 t = *(const f32x4*)p; v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
}
__device__ __forceinline__ void load4(const bf16_t* p, float (&v)[4]) {
  const u32x2 t = *(const u32x2*)p;
  v[0] = bf2f((bf16_t)(t[0] & 0xffff)); v[1] = bf2f((bf16_t)(t[0] >> 16));
  v[2] = bf2f((bf16_t)(t[1] & 0xffff)); v[3] = bf2f((bf16_t)(t[1] >> 16));
}
__device__ __forceinline__ void store4(float* p, const float (&v)[4]) {
  f32x4 t = {v[0], v[1], v[2], v[3]}; *(f32x4*)p = t;
}
__device__ __forceinline__ void store4(bf16_t* p, const float (&v)[4]) {
  u32x2 t; t[0] = pack2bf(v[0], v[1]); t[1] = pack2bf(v[2], v[3]); *(u32x2*)p = t;
}
struct f16_t { uint16_t v; };      // IEEE-half output rows (the frozen text tower's GEMM inputs)
__device__ __forceinline__ void store4(f16_t* p, const float (&v)[4]) {
  u32x2 t; t[0] = pack2h(v[0], v[1]); t[1] = pack2h(v[2], v[3]); *(u32x2*)p = t;
}
__device__ __forceinline__ void store1(f16_t* p, float v) { p->v = f2h(v); }
__device__ __forceinline__ float load1(const float* p) { return *p; }
__device__ __forceinline__ float load1(const bf16_t* p) { return bf2f(*p); }
__device__ __forceinline__ void store1(float* p, float v) { *p = v; }
__device__ __forceinline__ void store1(bf16_t* p, float v) { *p = f2bf(v); }

struct LnP {
  const void* x;        // MODE 0: rows [*, D] with stride xs (elements); MODE 1: tokens [B, T, D]
  const int64_t* ridx;  // MODE 0 optional: per output row r, source row = r*rmul + ridx[r]
  long xs; long rmul;
  const float* w; const float* b;
  void* y; long ys;
  void* y2;             // optional second output: the un-normalised assembled row (MODE 1), res dtype
  float* mean; float* rstd;  // optional [rows]
  int rows, D; float eps;
  // MODE 1 (assemble): row r = (bb, l) with l in [0, T]; l==0 -> cls + pos[0]; else tok[bb][l-1] + pos[l] (+ pos2[l-1])
  const float* cls; const float* pos; const float* pos2; int T;
};

// NCH = number of 256-element chunks held in registers (0 = generic any-D path)
template <int NCH, typename TIN, typename TOUT, int MODE>
__global__ void __launch_bounds__(256) ln_rows_kernel(const LnP p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.rows) return;
  const int D = p.D;
  const TIN* x;
  int l = 0;
  if constexpr (MODE == 0) {
    const long src = p.ridx ? (long)row * p.rmul + p.ridx[row] : (long)row;
    x = (const TIN*)p.x + src * p.xs;
  } else {
    const int bb = row / (p.T + 1); l = row - bb * (p.T + 1);
    x = (const TIN*)p.x + ((long)bb * p.T + (l > 0 ? l - 1 : 0)) * D;
  }
  TOUT* y = (TOUT*)p.y + (long)row * p.ys;
  const float invD = 1.0f / (float)D;

  if constexpr (NCH > 0) {
    float v[NCH][4];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int e = c * 256 + lane * 4;
      if constexpr (MODE == 0) {
        load4(x + e, v[c]);
      } else {
        float t[4], ps[4];
        if (l == 0) load4(p.cls + e, t); else load4(x + e, t);
        load4(p.pos + (long)l * D + e, ps);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[c][k] = t[k] + ps[k];
        if (p.pos2 && l > 0) {
          load4(p.pos2 + (long)(l - 1) * D + e, ps);
#pragma unroll
          for (int k = 0; k < 4; ++k) v[c][k] += ps[k];
        }
      }
    }
    if constexpr (MODE == 1) {
      if (p.y2) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) store4((float*)p.y2 + (long)row * D + c * 256 + lane * 4, v[c]);
      }
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
    const float mu = wave_sum(s) * invD;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int k = 0; k < 4; ++k) { const float d = v[c][k] - mu; q = fmaf(d, d, q); }
    const float rs = rsqrtf(wave_sum(q) * invD + p.eps);
    if (p.mean && lane == 0) { p.mean[row] = mu; p.rstd[row] = rs; }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int e = c * 256 + lane * 4;
      float ww[4], bb[4], o[4];
      load4(p.w + e, ww); load4(p.b + e, bb);
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = fmaf((v[c][k] - mu) * rs, ww[k], bb[k]);
      store4(y + e, o);
    }
  } else {
    auto fetch = [&](int e) -> float {
      if constexpr (MODE == 0) return load1(x + e);
      else {
        float t = (l == 0) ? p.cls[e] : load1(x + e);
        t += p.pos[(long)l * D + e];
        if (p.pos2 && l > 0) t += p.pos2[(long)(l - 1) * D + e];
        return t;
      }
    };
    float s = 0.f;
    for (int e = lane; e < D; e += 64) s += fetch(e);
    const float mu = wave_sum(s) * invD;
    float q = 0.f;
    for (int e = lane; e < D; e += 64) { const float d = fetch(e) - mu; q = fmaf(d, d, q); }
    const float rs = rsqrtf(wave_sum(q) * invD + p.eps);
    if (p.mean && lane == 0) { p.mean[row] = mu; p.rstd[row] = rs; }
    for (int e = lane; e < D; e += 64) {
      const float t = fetch(e);
      if constexpr (MODE == 1) { if (p.y2) ((float*)p.y2)[(long)row * D + e] = t; }
      store1(y + e, fmaf((t - mu) * rs, p.w[e], p.b[e]));
    }
  }
}

template <typename TIN, typename TOUT, int MODE>
hipError_t ln_launch(const LnP& p, hipStream_t s) {
  const dim3 grid((p.rows + 3) / 4), block(256);
  const bool vec = (p.D % 256 == 0) && (p.xs % 4 == 0) && (p.ys % 4 == 0);
  const int nch = vec ? p.D / 256 : 0;
  switch (nch) {
    case 1: hipLaunchKernelGGL((ln_rows_kernel<1, TIN, TOUT, MODE>), grid, block, 0, s, p); break;
    case 2: hipLaunchKernelGGL((ln_rows_kernel<2, TIN, TOUT, MODE>), grid, block, 0, s, p); break;
    case 3: hipLaunchKernelGGL((ln_rows_kernel<3, TIN, TOUT, MODE>), grid, block, 0, s, p); break;
    case 4: hipLaunchKernelGGL((ln_rows_kernel<4, TIN, TOUT, MODE>), grid, block, 0, s, p); break;
    case 5: hipLaunchKernelGGL((ln_rows_kernel<5, TIN, TOUT, MODE>), grid, block, 0, s, p); break;
    case 8: hipLaunchKernelGGL((ln_rows_kernel<8, TIN, TOUT, MODE>), grid, block, 0, s, p); break;
    default: hipLaunchKernelGGL((ln_rows_kernel<0, TIN, TOUT, MODE>), grid, block, 0, s, p); break;
  }
  return hipGetLastError();
}

template <int MODE>
hipError_t ln_dispatch(const LnP& p, int in_dt, int out_dt, hipStream_t s) {
  if (in_dt == VL_F32 && out_dt == VL_F16) return ln_launch<float, f16_t, MODE>(p, s);
  if (in_dt == VL_F32 && out_dt == VL_BF16) return ln_launch<float, bf16_t, MODE>(p, s);
  if (in_dt == VL_BF16 && out_dt == VL_BF16) return ln_launch<bf16_t, bf16_t, MODE>(p, s);
  if (in_dt == VL_F32 && out_dt == VL_F32) return ln_launch<float, float, MODE>(p, s);
  return ln_launch<bf16_t, float, MODE>(p, s);
}

// ---- L2 normalise rows (F.normalize, eps 1e-12) -> f32 and optional bf16 copy -------------------
__global__ void __launch_bounds__(256) l2norm_kernel(const float* x, float* y, bf16_t* yb, float* nrm,
                                                     int rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (long)row * D;
  float s = 0.f;
  for (int e = lane; e < D; e += 64) { const float v = xr[e]; s = fmaf(v, v, s); }
  const float n = sqrtf(wave_sum(s));
  const float inv = 1.0f / fmaxf(n, eps);
  if (nrm && lane == 0) nrm[row] = n;
  for (int e = lane; e < D; e += 64) {
    const float v = xr[e] * inv;
    if (y) y[(long)row * D + e] = v;
    if (yb) yb[(long)row * D + e] = f2bf(v);
  }
}

// ---- im2col for Conv2d(stride, no padding, no bias): [N,C,H,W] f32 -> patches bf16 [N*gh*gw, Kp] ------
// column order = (c, i, j) row-major = conv weight.reshape(width, -1) order; columns >= C*kh*kw are zero.
// `transpose_hw`: read the input as x[n, c, j_w, i_h] i.e. the AST tokenizer's transpose(2,3)
// (modal_audio/models/AST_tokenizer.py:46-47) fused into the gather.
struct I2cP { const float* x; void* out; int N, C, H, W, kh, kw, sh, sw, gh, gw, Kp, transpose_hw; };
template <typename TOUT>
__global__ void __launch_bounds__(256) im2col_kernel(const I2cP p) {
  const long total = (long)p.N * p.gh * p.gw * p.Kp;
  const int K = p.C * p.kh * p.kw;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int col = (int)(i % p.Kp);
    const long tok = i / p.Kp;
    float v = 0.f;
    if (col < K) {
      const int gw_ = (int)(tok % p.gw);
      const int gh_ = (int)((tok / p.gw) % p.gh);
      const int n = (int)(tok / ((long)p.gw * p.gh));
      const int c = col / (p.kh * p.kw);
      const int ij = col - c * (p.kh * p.kw);
      const int ii = ij / p.kw, jj = ij - ii * p.kw;
      const int hh = gh_ * p.sh + ii, ww = gw_ * p.sw + jj;   // coordinates in the conv input [H, W]
      if (p.transpose_hw)   // stored tensor is [N, W(stored rows = time), H(stored cols = freq)]
        v = p.x[((long)n * p.C + c) * p.H * p.W + (long)ww * p.H + hh];
      else
        v = p.x[((long)n * p.C + c) * p.H * p.W + (long)hh * p.W + ww];
    }
    store1((TOUT*)p.out + i, v);
  }
}

// ---- text embedding: out[b, l, :] = tok_emb[ids[b,l]] + pos[l] ------------------------------------
template <typename TOUT>
__global__ void __launch_bounds__(256) text_embed_kernel(const int64_t* ids, const float* emb, const float* pos,
                                                         TOUT* out, int rows, int L, int D, int vocab) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  long id = ids[row]; id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const float* e = emb + id * D;
  const float* ps = pos + (long)(row % L) * D;
  for (int i = lane; i < D; i += 64) store1(out + (long)row * D + i, e[i] + ps[i]);
}

// ---- elementwise helpers ------------------------------------------------------------------------
__global__ void __launch_bounds__(256) cast_f32_bf16_kernel(const float* x, bf16_t* y, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = f2bf(x[i]);
}
// y[r, :] = x[r, :] + t[r % T, :]   (x+pos for Lens tokens; transformer.py:743-745)
template <typename TIN, typename TOUT>
__global__ void __launch_bounds__(256) add_rows_kernel(const TIN* x, const float* t, TOUT* y, long rows, int T, int D) {
  const long total = rows * D;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / D; const int d = (int)(i - r * D);
    store1(y + i, load1(x + i) + t[(r % T) * D + d]);
  }
}

}  // namespace

extern "C" int vl_set_error(const char* msg);
#define VL_HIP_OK(e) do { hipError_t _e = (e); if (_e != hipSuccess) return vl_set_error(hipGetErrorString(_e)); } while (0)

static int grid_for(long n) { long g = (n + 255) / 256; return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g)); }

extern "C" int vl_layernorm_fwd(const void* x, int x_dtype, long x_row_stride, const int64_t* row_index, long row_mul,
                                const float* w, const float* b, void* y, int y_dtype, long y_row_stride,
                                float* mean, float* rstd, int rows, int D, float eps, hipStream_t stream) {
  if (rows <= 0 || D <= 0) return vl_set_error("vl_layernorm_fwd: empty problem");
  LnP p{}; p.x = x; p.ridx = row_index; p.xs = x_row_stride; p.rmul = row_mul; p.w = w; p.b = b; p.y = y; p.ys = y_row_stride;
  p.mean = mean; p.rstd = rstd; p.rows = rows; p.D = D; p.eps = eps;
  VL_HIP_OK(ln_dispatch<0>(p, x_dtype, y_dtype, stream));
  return 0;
}

// Row statistics (mean, rstd) for the LayerNorm folded into a GEMM epilogue (vl_gemm_lnfold_bf16): rows below m_main from the
// partial (sum, sum of squares) pairs the producing GEMM left per 64-column slice (vl_gemm_res_rowstats_bf16; summed in slice
// order: fixed), the remaining rows from the bf16 rows themselves (two-pass, as ln_rows_kernel).
namespace {
__global__ void __launch_bounds__(256) row_stats_kernel(const float* part, int P, const bf16_t* x, long xs, int D, int m_main,
                                                        int rows, float eps, float* mean, float* rstd, const float* lw,
                                                        const float* lb, bf16_t* y, long ys, int y_row0) {
  const int nb_main = (m_main + 255) >> 8;
  const float invD = 1.0f / (float)D;
  if ((int)blockIdx.x < nb_main) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= m_main) return;
    const vl_f32x2* pp = (const vl_f32x2*)part + (size_t)row * P;
    float s1 = 0.f, s2 = 0.f;
    for (int i = 0; i < P; ++i) { const vl_f32x2 v = pp[i]; s1 += v[0]; s2 += v[1]; }
    float mu = s1 * invD;
    const float ex2 = s2 * invD;
    float var = fmaxf(fmaf(-mu, mu, ex2), 0.f);
    // E[x^2] - mean^2 in fp32 loses log2(E[x^2] / var) bits: a row whose mean dwarfs its spread (|mean| > ~30 sigma) would get
    // a variance made of rounding noise, clamped to 0 at worst.  Such a row takes the two-pass formula on the stored values,
    // exactly as the leftover rows below do (one thread walks the row: rare, and correct beats fast here)
    if (var < 1e-3f * ex2) {
      const bf16_t* xr = x + (size_t)row * xs;
      float s = 0.f;
      for (int e = 0; e < D; ++e) s += bf2f(xr[e]);
      mu = s * invD;
      float q = 0.f;
      for (int e = 0; e < D; ++e) { const float d = bf2f(xr[e]) - mu; q = fmaf(d, d, q); }
      var = q * invD;
    }
    mean[row] = mu; rstd[row] = rsqrtf(var + eps);
    return;
  }
  const int lane = threadIdx.x & 63;
  const int row = m_main + ((int)blockIdx.x - nb_main) * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xr = x + (size_t)row * xs;
  if ((D & 511) == 0 && D <= 2048 && (xs & 7) == 0 && (((uintptr_t)x) & 15) == 0) {
    // the row once, in registers (8 consecutive values per lane and 512-column slab): this kernel is ~350 launches of a C3
    // step that do nothing but wait for memory, and the three passes below are three round trips
    u32x4 v[4];
    const int nc = D >> 9;
#pragma unroll
    for (int c = 0; c < 4; ++c) if (c < nc) v[c] = *(const u32x4*)(xr + (c * 64 + lane) * 8);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) if (c < nc)
#pragma unroll
      for (int e = 0; e < 4; ++e) s += bf2f((bf16_t)(v[c][e] & 0xffff)) + bf2f((bf16_t)(v[c][e] >> 16));
    const float mu = wave_sum(s) * invD;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) if (c < nc)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d0 = bf2f((bf16_t)(v[c][e] & 0xffff)) - mu, d1 = bf2f((bf16_t)(v[c][e] >> 16)) - mu;
        q = fmaf(d0, d0, q); q = fmaf(d1, d1, q);
      }
    const float rs = rsqrtf(wave_sum(q) * invD + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    if (y && row >= y_row0) {
      bf16_t* yr = y + (size_t)(row - y_row0) * ys;
#pragma unroll
      for (int c = 0; c < 4; ++c) if (c < nc) {
        const int e0 = (c * 64 + lane) * 8;
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          o[e] = pack2bf(fmaf((bf2f((bf16_t)(v[c][e] & 0xffff)) - mu) * rs, lw[e0 + 2 * e], lb[e0 + 2 * e]),
                         fmaf((bf2f((bf16_t)(v[c][e] >> 16)) - mu) * rs, lw[e0 + 2 * e + 1], lb[e0 + 2 * e + 1]));
        if ((ys & 7) == 0 && (((uintptr_t)y) & 15) == 0) *(u32x4*)(yr + e0) = o;
        else
#pragma unroll
          for (int e = 0; e < 4; ++e) { yr[e0 + 2 * e] = (bf16_t)(o[e] & 0xffff); yr[e0 + 2 * e + 1] = (bf16_t)(o[e] >> 16); }
      }
    }
    return;
  }
  float s = 0.f;
  for (int e = lane; e < D; e += 64) s += bf2f(xr[e]);
  const float mu = wave_sum(s) * invD;
  float q = 0.f;
  for (int e = lane; e < D; e += 64) { const float d = bf2f(xr[e]) - mu; q = fmaf(d, d, q); }
  const float rs = rsqrtf(wave_sum(q) * invD + eps);
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
  // the leftover rows of the CONSUMING GEMM (rows >= y_row0) also get their LayerNorm output here: those rows run layernorm + a
  // small-tile GEMM instead of the folded epilogue, and a separate 256-row LayerNorm launch per fold cost 8.6 us x 392 per C3 step
  if (y && row >= y_row0) {
    bf16_t* yr = y + (size_t)(row - y_row0) * ys;
    for (int e = lane; e < D; e += 64) yr[e] = f2bf(fmaf((bf2f(xr[e]) - mu) * rs, lw[e], lb[e]));
  }
}
}  // namespace

extern "C" int vl_ln_row_stats(const float* row_part, int P, const void* x_bf16, long x_row_stride, int D, int m_main, int rows,
                               float eps, float* mean, float* rstd, const float* ln_w, const float* ln_b, void* y_left,
                               long y_row_stride, int y_row0, hipStream_t stream) {
  if (rows <= 0 || D <= 0 || m_main < 0 || m_main > rows) return vl_set_error("vl_ln_row_stats: bad shape");
  if (m_main > 0 && (!row_part || P <= 0 || (((uintptr_t)row_part) & 7))) return vl_set_error("vl_ln_row_stats: partial statistics missing");
  if (m_main > 0 && P * 64 != D) return vl_set_error("vl_ln_row_stats: P must be D / 64 (one partial pair per 64-column slice of the row)");
  if (!mean || !rstd) return vl_set_error("vl_ln_row_stats: outputs missing");
  if (!x_bf16) return vl_set_error("vl_ln_row_stats: rows missing (the leftover rows and ill-conditioned rows are read from them)");
  if (y_left && (!ln_w || !ln_b || y_row0 < m_main || y_row0 > rows || y_row_stride < D))
    return vl_set_error("vl_ln_row_stats: y_left needs ln_w, ln_b, m_main <= y_row0 <= rows and a row stride >= D");
  const int nb = ((m_main + 255) >> 8) + (rows - m_main + 3) / 4;
  hipLaunchKernelGGL(row_stats_kernel, dim3(nb), dim3(256), 0, stream, row_part, P, (const bf16_t*)x_bf16, x_row_stride, D, m_main,
                     rows, eps, mean, rstd, ln_w, ln_b, (bf16_t*)y_left, y_row_stride, y_row0);
  VL_HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int vl_assemble_ln_pre(const void* tokens, int tok_dtype, const float* cls, const float* pos, const float* pos2,
                                  const float* w, const float* b, void* y, int y_dtype, float* xpre, float* mean, float* rstd,
                                  int B, int T, int D, float eps, hipStream_t stream) {
  if (B <= 0 || T <= 0 || D <= 0) return vl_set_error("vl_assemble_ln_pre: empty problem");
  LnP p{}; p.x = tokens; p.xs = D; p.w = w; p.b = b; p.y = y; p.ys = D; p.rows = B * (T + 1); p.D = D; p.eps = eps;
  p.cls = cls; p.pos = pos; p.pos2 = pos2; p.T = T; p.y2 = xpre; p.mean = mean; p.rstd = rstd;
  VL_HIP_OK(ln_dispatch<1>(p, tok_dtype, y_dtype, stream));
  return 0;
}

extern "C" int vl_l2_normalize(const float* x, float* y, void* y_bf16, float* norms, int rows, int D, float eps,
                               hipStream_t stream) {
  if (rows <= 0 || D <= 0) return vl_set_error("vl_l2_normalize: empty problem");
  hipLaunchKernelGGL(l2norm_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, x, y, (bf16_t*)y_bf16, norms, rows, D, eps);
  VL_HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int vl_im2col_bf16(const float* x, void* patches, int N, int C, int H, int W, int kh, int kw, int sh, int sw,
                              int Kp, int transpose_hw, hipStream_t stream) {
  if (N <= 0 || H < kh || W < kw) return vl_set_error("vl_im2col_bf16: bad shape");
  if (Kp < C * kh * kw) return vl_set_error("vl_im2col_bf16: Kp smaller than C*kh*kw");
  I2cP p{x, patches, N, C, H, W, kh, kw, sh, sw, (H - kh) / sh + 1, (W - kw) / sw + 1, Kp, transpose_hw};
  hipLaunchKernelGGL(im2col_kernel<bf16_t>, dim3(grid_for((long)N * p.gh * p.gw * Kp)), dim3(256), 0, stream, p);
  VL_HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int vl_im2col_f32(const float* x, float* patches, int N, int C, int H, int W, int kh, int kw, int sh, int sw,
                             int Kp, int transpose_hw, hipStream_t stream) {
  if (N <= 0 || H < kh || W < kw) return vl_set_error("vl_im2col_f32: bad shape");
  if (Kp < C * kh * kw) return vl_set_error("vl_im2col_f32: Kp smaller than C*kh*kw");
  I2cP p{x, patches, N, C, H, W, kh, kw, sh, sw, (H - kh) / sh + 1, (W - kw) / sw + 1, Kp, transpose_hw};
  hipLaunchKernelGGL(im2col_kernel<float>, dim3(grid_for((long)N * p.gh * p.gw * Kp)), dim3(256), 0, stream, p);
  VL_HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int vl_text_embed(const int64_t* ids, const float* tok_emb, const float* pos, void* out, int out_dtype,
                             int B, int L, int D, int vocab, hipStream_t stream) {
  if (B <= 0 || L <= 0) return vl_set_error("vl_text_embed: empty problem");
  const int rows = B * L;
  if (out_dtype == VL_F32)
    hipLaunchKernelGGL(text_embed_kernel<float>, dim3((rows + 3) / 4), dim3(256), 0, stream, ids, tok_emb, pos, (float*)out, rows, L, D, vocab);
  else
    hipLaunchKernelGGL(text_embed_kernel<bf16_t>, dim3((rows + 3) / 4), dim3(256), 0, stream, ids, tok_emb, pos, (bf16_t*)out, rows, L, D, vocab);
  VL_HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int vl_cast_f32_bf16(const float* x, void* y, long n, hipStream_t stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, (bf16_t*)y, n);
  VL_HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int vl_add_rows(const void* x, int x_dtype, const float* table, void* y, int y_dtype, long rows, int T, int D,
                           hipStream_t stream) {
  if (rows <= 0) return 0;
  const dim3 g(grid_for(rows * D)), b(256);
  if (x_dtype == VL_F32 && y_dtype == VL_F32) hipLaunchKernelGGL((add_rows_kernel<float, float>), g, b, 0, stream, (const float*)x, table, (float*)y, rows, T, D);
  else if (x_dtype == VL_F32) hipLaunchKernelGGL((add_rows_kernel<float, bf16_t>), g, b, 0, stream, (const float*)x, table, (bf16_t*)y, rows, T, D);
  else if (y_dtype == VL_F32) hipLaunchKernelGGL((add_rows_kernel<bf16_t, float>), g, b, 0, stream, (const bf16_t*)x, table, (float*)y, rows, T, D);
  else hipLaunchKernelGGL((add_rows_kernel<bf16_t, bf16_t>), g, b, 0, stream, (const bf16_t*)x, table, (bf16_t*)y, rows, T, D);
  VL_HIP_OK(hipGetLastError());
  return 0;
}
