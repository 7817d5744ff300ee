// True fp32 arithmetic for INFERENCE (`precision="fp32"` of the reference's factory, open_clip/factory.py:260-295,
// training/precision.py:5-12: no autocast, every nn.Linear / attention product in fp32).  gfx950 has no xf32 / TF32 mode;
// it has fp32-input MFMA at the fp32 VECTOR rate (64 flop / clock / SIMD = 157 TFLOP/s, 1/16 of bf16) whose result is the
// exact fmaf chain.  Two kernels, written for correctness and a sane fraction of that rate - this path exists so that a user
// who asks for fp32 evaluation gets fp32 features (1e-5 of the CPU path), not so that anything trains on it:
//   gemm_f32_kernel   C = act(alpha A W^T + bias) (+ res): 128x128 tile per workgroup, 4 waves of 64x64 (2x2 blocks of
//                     `v_mfma_f32_32x32x2_f32`), 16-deep k-slabs through LDS (rows padded to 17 floats: the fragment read -
//                     32 consecutive rows at one k - touches 32 banks), next slab in registers under the MFMAs of the current
//   attn_f32_kernel   softmax(q k^T + causal mask) v with one THREAD per query row (q and the output row in registers), keys
//                     and values staged in LDS 64 at a time and read as broadcasts, online softmax per 16 keys
// Replaces nn.Linear / F.multi_head_attention_forward of VisionTransformer / TextTransformer (open_clip/transformer.py:
// 226-272, 241-252) under precision="fp32".
#include "vl_common.h"
#include "vitlens_hip.h"

extern "C" int vl_set_error(const char* msg);

namespace {

struct GemmF32P {
  const float* A; const float* W; const float* bias; const float* res; float* out;
  int M, N, K, lda, ldw, ldo;
  float alpha;
};

constexpr int FT = 128, FK = 16, FP = FK + 1;

template <int ACT>
__global__ void __launch_bounds__(256) gemm_f32_kernel(const GemmF32P p) {
  __shared__ float As[FT * FP], Ws[FT * FP];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid & 1, wn = wid >> 1;
  const int tiles_n = (p.N + FT - 1) / FT;
  const int m0 = (blockIdx.x / tiles_n) * FT, n0 = (blockIdx.x % tiles_n) * FT;
  // staging: thread t moves float4 #t and #t+256 of each operand's 128 x 16 slab: row = idx / 4, k = (idx % 4) * 4
  const int srow0 = tid >> 2, sk = (tid & 3) * 4;
  auto load_slab = [&](const float* base, int ld, int row_lim, int r0, int k0, f32x4 (&v)[2]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      int row = r0 + srow0 + u * 64;
      row = row < row_lim ? row : row_lim - 1;                       // rows beyond the matrix: any valid row, masked at the store
      const float* src = base + (size_t)row * ld + k0 + sk;
      if (k0 + sk + 3 < p.K) v[u] = *(const f32x4*)src;
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[u][e] = (k0 + sk + e < p.K) ? src[e] : 0.f;
      }
    }
  };
  auto store_slab = [&](float* dst, const f32x4 (&v)[2]) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) dst[(srow0 + u * 64) * FP + sk + e] = v[u][e];
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  f32x4 ra[2], rw[2];
  load_slab(p.A, p.lda, p.M, m0, 0, ra);
  load_slab(p.W, p.ldw, p.N, n0, 0, rw);
  const int fr = lane & 31, fk = lane >> 5;
  for (int k0 = 0; k0 < p.K; k0 += FK) {
    __syncthreads();                                                 // the previous slab's fragment reads are done
    store_slab(As, ra); store_slab(Ws, rw);
    __syncthreads();
    if (k0 + FK < p.K) { load_slab(p.A, p.lda, p.M, m0, k0 + FK, ra); load_slab(p.W, p.ldw, p.N, n0, k0 + FK, rw); }
#pragma unroll
    for (int kk = 0; kk < FK / 2; ++kk) {
      float a[2], w[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = As[(wm * 64 + i * 32 + fr) * FP + kk * 2 + fk];
#pragma unroll
      for (int j = 0; j < 2; ++j) w[j] = Ws[(wn * 64 + j * 32 + fr) * FP + kk * 2 + fk];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], w[j], acc[i][j], 0, 0, 0);
    }
  }
  // D[i][j]: lane owns column j = lane & 31 and the rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of the 32 x 32 block
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 64 + j * 32 + fr;
      if (n >= p.N) continue;
      const float b = p.bias ? p.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
        if (m >= p.M) continue;
        float v = fmaf(acc[i][j][r], p.alpha, b);
        if constexpr (ACT == VL_ACT_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
        else if constexpr (ACT == VL_ACT_RELU) v = fmaxf(v, 0.f);
        if (p.res) v += p.res[(size_t)m * p.ldo + n];
        p.out[(size_t)m * p.ldo + n] = v;
      }
    }
}

struct AttnF32P {
  const float* q; const float* k; const float* v;
  long s[9];              // (batch, head, row) strides of q, k, v in elements
  float* out; float* lse;
  int B, H, Lq, Lk, causal;
  float scale;
};

constexpr int AKC = 64;   // keys per LDS chunk

template <int DH>
__global__ void __launch_bounds__(256) attn_f32_kernel(const AttnF32P p) {
  __shared__ float Ks[AKC * DH], Vs[AKC * DH];
  const int b = blockIdx.z, h = blockIdx.y, tid = threadIdx.x;
  const int qi = blockIdx.x * 256 + tid;
  const bool live = qi < p.Lq;
  const float* Kg = p.k + b * p.s[3] + h * p.s[4];
  const float* Vg = p.v + b * p.s[6] + h * p.s[7];
  float q[DH], o[DH];
  {
    const float* Qg = p.q + b * p.s[0] + h * p.s[1] + (long)(live ? qi : p.Lq - 1) * p.s[2];
#pragma unroll
    for (int d = 0; d < DH; d += 4) {
      const f32x4 t = *(const f32x4*)(Qg + d);
#pragma unroll
      for (int e = 0; e < 4; ++e) { q[d + e] = t[e] * p.scale; o[d + e] = 0.f; }
    }
  }
  float m = -INFINITY, l = 0.f;
  // the last key any query of this workgroup may see (causal: nothing beyond the workgroup's last row)
  const int k_hi = p.causal ? min(p.Lk, (int)(blockIdx.x * 256 + 256)) : p.Lk;
  for (int kc = 0; kc < k_hi; kc += AKC) {
    __syncthreads();
    for (int i = tid; i < AKC * DH / 4; i += 256) {
      const int row = i / (DH / 4), c = (i % (DH / 4)) * 4;
      f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
      if (kc + row < p.Lk) { kv = *(const f32x4*)(Kg + (long)(kc + row) * p.s[5] + c); vv = *(const f32x4*)(Vg + (long)(kc + row) * p.s[8] + c); }
      *(f32x4*)(Ks + row * DH + c) = kv; *(f32x4*)(Vs + row * DH + c) = vv;
    }
    __syncthreads();
    if (!live) continue;
    for (int j0 = 0; j0 < AKC && kc + j0 < k_hi; j0 += 16) {
      float s[16];
      float bm = -INFINITY;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int key = kc + j0 + j;
        float acc = 0.f;
#pragma unroll
        for (int d = 0; d < DH; d += 4) {
          const f32x4 kv = *(const f32x4*)(Ks + (j0 + j) * DH + d);
          acc = fmaf(q[d], kv[0], acc); acc = fmaf(q[d + 1], kv[1], acc); acc = fmaf(q[d + 2], kv[2], acc); acc = fmaf(q[d + 3], kv[3], acc);
        }
        s[j] = (key < p.Lk && (!p.causal || key <= qi)) ? acc : -INFINITY;
        bm = fmaxf(bm, s[j]);
      }
      if (bm == -INFINITY) continue;                                 // every key of the block masked for this row
      if (bm > m) {
        const float f = __expf(m - bm);                              // (m = -inf on the first block: f = 0)
        l *= f;
#pragma unroll
        for (int d = 0; d < DH; ++d) o[d] *= f;
        m = bm;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float pj = __expf(s[j] - m);                           // masked keys: exp(-inf) = 0
        l += pj;
#pragma unroll
        for (int d = 0; d < DH; d += 4) {
          const f32x4 vv = *(const f32x4*)(Vs + (j0 + j) * DH + d);
          o[d] = fmaf(pj, vv[0], o[d]); o[d + 1] = fmaf(pj, vv[1], o[d + 1]); o[d + 2] = fmaf(pj, vv[2], o[d + 2]); o[d + 3] = fmaf(pj, vv[3], o[d + 3]);
        }
      }
    }
  }
  if (!live) return;
  const float inv = 1.0f / l;
  float* dst = p.out + ((size_t)b * p.Lq + qi) * (p.H * DH) + h * DH;
#pragma unroll
  for (int d = 0; d < DH; d += 4) *(f32x4*)(dst + d) = f32x4{o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv};
  if (p.lse) p.lse[((size_t)b * p.H + h) * p.Lq + qi] = m + __logf(l);
}

}  // namespace

extern "C" int vl_gemm_f32(const float* A, const float* W, const float* bias, float* out, const float* res, int M, int N, int K,
                           int lda, int ldw, int ldo, float alpha, int act, hipStream_t stream) {
  if (!A || !W || !out) return vl_set_error("vl_gemm_f32: null operand");
  if (M <= 0 || N <= 0 || K <= 0) return vl_set_error("vl_gemm_f32: empty problem");
  if ((K & 3) || (lda & 3) || (ldw & 3)) return vl_set_error("vl_gemm_f32: K, lda, ldw must be multiples of 4 (16-byte rows)");
  if ((((uintptr_t)A) | ((uintptr_t)W)) & 15) return vl_set_error("vl_gemm_f32: operands must be 16-byte aligned");
  const GemmF32P p{A, W, bias, res, out, M, N, K, lda, ldw, ldo, alpha};
  const dim3 grid(((M + FT - 1) / FT) * ((N + FT - 1) / FT));
  if (act == VL_ACT_NONE) hipLaunchKernelGGL(gemm_f32_kernel<VL_ACT_NONE>, grid, dim3(256), 0, stream, p);
  else if (act == VL_ACT_GELU) hipLaunchKernelGGL(gemm_f32_kernel<VL_ACT_GELU>, grid, dim3(256), 0, stream, p);
  else if (act == VL_ACT_RELU) hipLaunchKernelGGL(gemm_f32_kernel<VL_ACT_RELU>, grid, dim3(256), 0, stream, p);
  else return vl_set_error("vl_gemm_f32: act must be none, GELU or ReLU");
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : vl_set_error(hipGetErrorString(e));
}

extern "C" int vl_attn_fwd_f32(const float* q, const float* k, const float* v, const long* strides, float* out, float* lse,
                               int B, int H, int Lq, int Lk, int dh, float scale, int causal, hipStream_t stream) {
  if (B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0) return vl_set_error("vl_attn_fwd_f32: empty problem");
  if (dh != 32 && dh != 64) return vl_set_error("vl_attn_fwd_f32: head dim must be 32 or 64 (the query and its output row live in registers)");
  if (!strides || !q || !k || !v || !out) return vl_set_error("vl_attn_fwd_f32: null operand");
  for (int i = 0; i < 9; ++i)
    if (strides[i] & 3) return vl_set_error("vl_attn_fwd_f32: operand strides must be multiples of 4 elements (16-byte rows)");
  if ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v) | ((uintptr_t)out)) & 15) return vl_set_error("vl_attn_fwd_f32: operands must be 16-byte aligned");
  AttnF32P p{q, k, v, {0}, out, lse, B, H, Lq, Lk, causal, scale};
  for (int i = 0; i < 9; ++i) p.s[i] = strides[i];
  const dim3 grid((Lq + 255) / 256, H, B);
  if (dh == 64) hipLaunchKernelGGL(attn_f32_kernel<64>, grid, dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(attn_f32_kernel<32>, grid, dim3(256), 0, stream, p);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : vl_set_error(hipGetErrorString(e));
}
