// Fused attention backward for gfx950 (two kernels, no atomics, scores recomputed in registers).
//
// Autograd of F.multi_head_attention_forward / the Perceiver einsum attention
// (open_clip/transformer.py:241-252, open_clip/perceiver.py:128-145) for the trainable towers.
//   kernel A: one wave per 32 QUERIES  ->  dQ      (loops over key tiles)
//   kernel B: one wave per 32 KEYS     ->  dK, dV  (loops over query tiles)
// Both recompute P = exp2(S2 - lse2) from the forward's log-sum-exp and use the same swapped-operand
// trick as the forward: the wave's own index (query in A, key in B) is the MFMA column, so P / dS
// live 16-per-lane in accumulator layout and feed the next MFMA as its column operand straight from
// registers, while the row operand (K^T / Q^T / dO^T, stored transposed) is read with the matching
// k-slot permutation.
//   inputs : q (pre-scaled by scale*log2e), k, v   [B,H,L,64|32] row-major
//            qt, kt                               [B,H,dh,Lp]   transposed copies
//            dO [B,H,Lq,dh], dOt [B,H,dh,Lqp], lse [B,H,Lq] (natural log), delta [B,H,Lq] = rowsum(dO*O)
//   outputs: dq -> dQKV[(b*Lq+q), 0*D + h*dh + d], dk -> [.., 1*D ..], dv -> [.., 2*D ..]  (token-major, bf16)
#include "vl_common.h"
#include "vitlens_hip.h"

namespace {

constexpr int KC = 288;
constexpr int VS = KC + 4;

struct AttnBwdP {
  const bf16_t *q, *k, *v, *qt, *kt, *dO, *dOt;
  const float *lse, *delta;
  bf16_t *dq, *dk, *dv;     // token-major destinations (already offset to the q / k / v column block)
  long ld_dq, ld_dkv;       // row strides (elements) of the dq and dk/dv destinations
  int B, H, Lq, Lk, Lqp, Lkp, causal;
  float scale;              // softmax scale (dq = scale * dS K ; dk = ln2 * dS^T Q2)
};

// stage `KC` rows of a row-major [L, DH] matrix into a swizzled LDS image (rows >= nvalid zero-filled)
template <int DH>
__device__ __forceinline__ void stage_rows(unsigned char* dst, const bf16_t* src, int row0, int L, int tid, int nthr) {
  constexpr int RB = DH * 2, CH = RB / 16, RSH = (DH == 64) ? 1 : 2, NP = KC * CH;
  for (int base = 0; base < NP; base += 4 * nthr) {
    u32x4 t[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = base + u * nthr + tid;
      const int row = i / CH, c = i % CH;
      t[u] = u32x4{0u, 0u, 0u, 0u};
      if (i < NP && row0 + row < L) t[u] = *(const u32x4*)(src + (size_t)(row0 + row) * DH + c * 8);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = base + u * nthr + tid;
      const int row = i / CH, c = i % CH;
      if (i < NP) *(u32x4*)(dst + row * RB + ((c ^ ((row >> RSH) & (CH - 1))) * 16)) = t[u];
    }
  }
}
// stage KC columns [col0, col0+KC) of a transposed [DH, Lp] matrix into LDS rows of stride VS (cols >= L zeroed)
template <int DH>
__device__ __forceinline__ void stage_cols(bf16_t* dst, const bf16_t* src, int col0, int L, int Lp, int tid, int nthr) {
  constexpr int NP = DH * (KC / 8);
  for (int base = 0; base < NP; base += 4 * nthr) {
    u32x4 t[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = base + u * nthr + tid;
      const int d = i / (KC / 8), kp = i % (KC / 8);
      const int c = col0 + kp * 8;
      t[u] = u32x4{0u, 0u, 0u, 0u};
      if (i < NP && c + 8 <= Lp) t[u] = *(const u32x4*)(src + (size_t)d * Lp + c);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = base + u * nthr + tid;
      if (i >= NP) continue;
      const int d = i / (KC / 8), kp = i % (KC / 8);
      const int c = col0 + kp * 8;
      u32x4 w = t[u];
      if (c + 8 > L) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          unsigned int x = w[e];
          if (c + 2 * e >= L) x &= 0xffff0000u;
          if (c + 2 * e + 1 >= L) x &= 0x0000ffffu;
          w[e] = x;
        }
      }
      u32x2* o = (u32x2*)(dst + d * VS + kp * 8);
      u32x2 lo = {w[0], w[1]}, hi = {w[2], w[3]};
      o[0] = lo; o[1] = hi;
    }
  }
}

template <int DH>
__device__ __forceinline__ bf16x8 frag_rows(const unsigned char* base, int row, int ks, int fg) {
  constexpr int RB = DH * 2, CH = RB / 16, RSH = (DH == 64) ? 1 : 2;
  return *(const bf16x8*)(base + row * RB + (((ks * 2 + fg) ^ ((row >> RSH) & (CH - 1))) * 16));
}
__device__ __forceinline__ bf16x8 frag_cols(const bf16_t* base, int d, int col) {
  const bf16x4 a = *(const bf16x4*)(base + d * VS + col);
  const bf16x4 b = *(const bf16x4*)(base + d * VS + col + 8);
  bf16x8 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) { r[e] = a[e]; r[4 + e] = b[e]; }
  return r;
}

// ------------------------------------------------------------------------------------------- kernel A: dQ
template <int DH>
__global__ void __launch_bounds__(576) attn_bwd_dq_kernel(const AttnBwdP p) {
  constexpr int RB = DH * 2, KS = DH / 16, DT = DH / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sK = smem;
  unsigned char* sV = smem + KC * RB;
  bf16_t* sKt = (bf16_t*)(smem + 2 * KC * RB);

  const int b = blockIdx.z, h = blockIdx.y;
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6), nwq = nthr >> 6;
  const int fr = lane & 31, fg = lane >> 5;
  const size_t bh = (size_t)b * p.H + h;
  const int q0 = (blockIdx.x * nwq + wid) * 32;
  const bool active = q0 < p.Lq;
  const int qidx = q0 + fr;
  const int qrow = qidx < p.Lq ? qidx : p.Lq - 1;

  bf16x8 qf[KS], dof[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    qf[ks] = *(const bf16x8*)(p.q + (bh * p.Lq + qrow) * DH + ks * 16 + fg * 8);
    dof[ks] = *(const bf16x8*)(p.dO + (bh * p.Lq + qrow) * DH + ks * 16 + fg * 8);
  }
  const float lse2 = p.lse[bh * p.Lq + qrow] * 1.4426950408889634f;
  const float dlt = p.delta[bh * p.Lq + qrow];

  f32x16 dq[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[t][r] = 0.f;

  const int blk_q_hi = min(p.Lq - 1, (int)(blockIdx.x * nwq + nwq) * 32 - 1);
  for (int kc0 = 0; kc0 < p.Lk; kc0 += KC) {
    if (p.causal && kc0 > blk_q_hi) break;
    __syncthreads();
    stage_rows<DH>(sK, p.k + bh * p.Lk * DH, kc0, p.Lk, tid, nthr);
    stage_rows<DH>(sV, p.v + bh * p.Lk * DH, kc0, p.Lk, tid, nthr);
    stage_cols<DH>(sKt, p.kt + bh * DH * (size_t)p.Lkp, kc0, p.Lk, p.Lkp, tid, nthr);
    __syncthreads();
    if (!active) continue;
    int ntile = (min(p.Lk - kc0, KC) + 31) >> 5;
    if (p.causal) ntile = min(ntile, ((q0 + 31 - kc0) >> 5) + 1);
    for (int kt = 0; kt < ntile; ++kt) {
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<DH>(sK, kt * 32 + fr, ks, fg), qf[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<DH>(sV, kt * 32 + fr, ks, fg), dof[ks], dp, 0, 0, 0);
      }
      const int key0 = kc0 + kt * 32 + fg * 4;
      float ds[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = key0 + (r & 3) + 8 * (r >> 2);
        float pr = __builtin_amdgcn_exp2f(s[r] - lse2);
        if (key >= p.Lk || (p.causal && key > qidx)) pr = 0.f;
        ds[r] = pr * (dp[r] - dlt);
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        bf16x8 df;
#pragma unroll
        for (int e = 0; e < 8; ++e) df[e] = (__bf16)ds[c * 8 + e];
#pragma unroll
        for (int t = 0; t < DT; ++t)
          dq[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(sKt, t * 32 + fr, kt * 32 + c * 16 + fg * 4), df, dq[t], 0, 0, 0);
      }
    }
  }
  if (!active || qidx >= p.Lq) return;
  bf16_t* og = p.dq + ((size_t)b * p.Lq + qidx) * p.ld_dq + h * DH;
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      u32x2 w;
      w[0] = pack2bf(dq[t][qd * 4 + 0] * p.scale, dq[t][qd * 4 + 1] * p.scale);
      w[1] = pack2bf(dq[t][qd * 4 + 2] * p.scale, dq[t][qd * 4 + 3] * p.scale);
      *(u32x2*)(og + t * 32 + qd * 8 + fg * 4) = w;
    }
}

// ------------------------------------------------------------------------------------- kernel B: dK, dV
template <int DH>
__global__ void __launch_bounds__(576) attn_bwd_dkv_kernel(const AttnBwdP p) {
  constexpr int RB = DH * 2, KS = DH / 16, DT = DH / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sQ = smem;
  unsigned char* sdO = smem + KC * RB;
  bf16_t* sQt = (bf16_t*)(smem + 2 * KC * RB);
  bf16_t* sdOt = sQt + DH * VS;
  float* sLse = (float*)(sdOt + DH * VS);
  float* sDel = sLse + KC;

  const int b = blockIdx.z, h = blockIdx.y;
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6), nwk = nthr >> 6;
  const int fr = lane & 31, fg = lane >> 5;
  const size_t bh = (size_t)b * p.H + h;
  const int k0 = (blockIdx.x * nwk + wid) * 32;
  const bool active = k0 < p.Lk;
  const int kidx = k0 + fr;
  const int krow = kidx < p.Lk ? kidx : p.Lk - 1;

  bf16x8 kf[KS], vf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    kf[ks] = *(const bf16x8*)(p.k + (bh * p.Lk + krow) * DH + ks * 16 + fg * 8);
    vf[ks] = *(const bf16x8*)(p.v + (bh * p.Lk + krow) * DH + ks * 16 + fg * 8);
  }
  f32x16 dk[DT], dv[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[t][r] = 0.f; dv[t][r] = 0.f; }

  for (int qc0 = 0; qc0 < p.Lq; qc0 += KC) {
    __syncthreads();
    stage_rows<DH>(sQ, p.q + bh * p.Lq * DH, qc0, p.Lq, tid, nthr);
    stage_rows<DH>(sdO, p.dO + bh * p.Lq * DH, qc0, p.Lq, tid, nthr);
    stage_cols<DH>(sQt, p.qt + bh * DH * (size_t)p.Lqp, qc0, p.Lq, p.Lqp, tid, nthr);
    stage_cols<DH>(sdOt, p.dOt + bh * DH * (size_t)p.Lqp, qc0, p.Lq, p.Lqp, tid, nthr);
    for (int i = tid; i < KC; i += nthr) {
      const bool ok = qc0 + i < p.Lq;
      sLse[i] = ok ? p.lse[bh * p.Lq + qc0 + i] * 1.4426950408889634f : 0.f;
      sDel[i] = ok ? p.delta[bh * p.Lq + qc0 + i] : 0.f;
    }
    __syncthreads();
    if (!active) continue;
    const int ntile = (min(p.Lq - qc0, KC) + 31) >> 5;
    int t0 = 0;
    if (p.causal) t0 = max(0, (k0 - qc0) >> 5);      // queries before this key tile never see it
    for (int qt = t0; qt < ntile; ++qt) {
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<DH>(sQ, qt * 32 + fr, ks, fg), kf[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<DH>(sdO, qt * 32 + fr, ks, fg), vf[ks], dp, 0, 0, 0);
      }
      // rows of the accumulator = queries  (r&3) + 8*(r>>2) + 4*fg of this tile; column = this lane's key
      float pv[16], ds[16];
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int ql = qt * 32 + qd * 8 + fg * 4;
        const f32x4 l4 = *(const f32x4*)(sLse + ql);
        const f32x4 d4 = *(const f32x4*)(sDel + ql);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = qd * 4 + e;
          const int qg = qc0 + ql + e;
          float pr = __builtin_amdgcn_exp2f(s[r] - l4[e]);
          if (qg >= p.Lq || (p.causal && kidx > qg)) pr = 0.f;
          pv[r] = pr;
          ds[r] = pr * (dp[r] - d4[e]);
        }
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        bf16x8 pf, df;
#pragma unroll
        for (int e = 0; e < 8; ++e) { pf[e] = (__bf16)pv[c * 8 + e]; df[e] = (__bf16)ds[c * 8 + e]; }
#pragma unroll
        for (int t = 0; t < DT; ++t) {
          const int col = qt * 32 + c * 16 + fg * 4;
          dv[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(sdOt, t * 32 + fr, col), pf, dv[t], 0, 0, 0);
          dk[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(sQt, t * 32 + fr, col), df, dk[t], 0, 0, 0);
        }
      }
    }
  }
  if (!active || kidx >= p.Lk) return;
  const float ln2 = 0.6931471805599453f;
  bf16_t* gk = p.dk + ((size_t)b * p.Lk + kidx) * p.ld_dkv + h * DH;
  bf16_t* gv = p.dv + ((size_t)b * p.Lk + kidx) * p.ld_dkv + h * DH;
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const int d = t * 32 + qd * 8 + fg * 4;
      u32x2 w;
      w[0] = pack2bf(dk[t][qd * 4 + 0] * ln2, dk[t][qd * 4 + 1] * ln2);
      w[1] = pack2bf(dk[t][qd * 4 + 2] * ln2, dk[t][qd * 4 + 3] * ln2);
      *(u32x2*)(gk + d) = w;
      w[0] = pack2bf(dv[t][qd * 4 + 0], dv[t][qd * 4 + 1]);
      w[1] = pack2bf(dv[t][qd * 4 + 2], dv[t][qd * 4 + 3]);
      *(u32x2*)(gv + d) = w;
    }
}

// delta[b,h,l] = sum_d dO[b,h,l,d] * O[b*L+l, h*dh+d]
__global__ void __launch_bounds__(256) attn_delta_kernel(const bf16_t* dO, const bf16_t* o, float* delta, int B, int H, int L, int dh) {
  const long n = (long)B * H * L;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int l = (int)(i % L); const long bh = i / L; const int h = (int)(bh % H); const long b = bh / H;
    const bf16_t* a = dO + i * dh;
    const bf16_t* c = o + (b * L + l) * (long)(H * dh) + h * dh;
    float s = 0.f;
    for (int d = 0; d < dh; d += 8) {
      const u32x4 x = *(const u32x4*)(a + d), y = *(const u32x4*)(c + d);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s = fmaf(bf2f((bf16_t)(x[e] & 0xffff)), bf2f((bf16_t)(y[e] & 0xffff)), s);
        s = fmaf(bf2f((bf16_t)(x[e] >> 16)), bf2f((bf16_t)(y[e] >> 16)), s);
      }
    }
    delta[i] = s;
  }
}

}  // namespace

extern "C" int vl_set_error(const char* msg);

extern "C" int vl_attn_delta(const void* dO, const void* o, float* delta, int B, int H, int L, int dh, hipStream_t stream) {
  if (B <= 0 || H <= 0 || L <= 0 || (dh & 7)) return vl_set_error("vl_attn_delta: bad shape");
  long n = (long)B * H * L; long g = (n + 255) / 256; if (g > 8192) g = 8192;
  hipLaunchKernelGGL(attn_delta_kernel, dim3((int)g), dim3(256), 0, stream, (const bf16_t*)dO, (const bf16_t*)o, delta, B, H, L, dh);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : vl_set_error(hipGetErrorString(e));
}

extern "C" int vl_attn_bwd_bf16(const void* q, const void* k, const void* v, const void* qt, const void* kt,
                                const void* dO, const void* dOt, const float* lse, const float* delta,
                                void* dq, void* dk, void* dv, long ld_dq, long ld_dkv,
                                int B, int H, int Lq, int Lk, int Lqp, int Lkp, int dh, int causal, float scale,
                                hipStream_t stream) {
  if (B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0) return vl_set_error("vl_attn_bwd_bf16: empty problem");
  if (dh != 64 && dh != 32) return vl_set_error("vl_attn_bwd_bf16: head dim must be 32 or 64");
  if ((Lqp & 7) || (Lkp & 7) || Lqp < Lq || Lkp < Lk) return vl_set_error("vl_attn_bwd_bf16: bad padded lengths");
  AttnBwdP p{(const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)qt, (const bf16_t*)kt,
             (const bf16_t*)dO, (const bf16_t*)dOt, lse, delta, (bf16_t*)dq, (bf16_t*)dk, (bf16_t*)dv,
             ld_dq, ld_dkv, B, H, Lq, Lk, Lqp, Lkp, causal, scale};
  const int qtiles = (Lq + 31) / 32, ktiles = (Lk + 31) / 32;
  const int nwq = qtiles <= 9 ? qtiles : 8, nwk = ktiles <= 9 ? ktiles : 8;
  const size_t smA = (size_t)2 * KC * dh * 2 + (size_t)dh * VS * 2;
  const size_t smB = (size_t)2 * KC * dh * 2 + (size_t)2 * dh * VS * 2 + 2 * KC * sizeof(float);
  hipError_t e;
#define VL_LAUNCH_BWD(DHV)                                                                                                   \
  do {                                                                                                                        \
    static bool set_##DHV = false;                                                                                            \
    if (!set_##DHV) {                                                                                                         \
      e = hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<DHV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smA);    \
      if (e != hipSuccess) return vl_set_error(hipGetErrorString(e));                                                         \
      e = hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<DHV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smB);   \
      if (e != hipSuccess) return vl_set_error(hipGetErrorString(e));                                                         \
      set_##DHV = true;                                                                                                       \
    }                                                                                                                         \
    hipLaunchKernelGGL(attn_bwd_dq_kernel<DHV>, dim3((qtiles + nwq - 1) / nwq, H, B), dim3(nwq * 64), smA, stream, p);        \
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<DHV>, dim3((ktiles + nwk - 1) / nwk, H, B), dim3(nwk * 64), smB, stream, p);       \
  } while (0)
  if (dh == 64) VL_LAUNCH_BWD(64); else VL_LAUNCH_BWD(32);
#undef VL_LAUNCH_BWD
  e = hipGetLastError();
  return e == hipSuccess ? 0 : vl_set_error(hipGetErrorString(e));
}
