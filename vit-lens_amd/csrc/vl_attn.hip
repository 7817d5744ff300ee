// Fused (flash-style) attention forward for gfx950.
//
// Replaces F.multi_head_attention_forward inside nn.MultiheadAttention
// (open_clip/transformer.py:241-252; additive causal mask :870-876) and the einsum attention of
// the Perceiver "Lens" (open_clip/perceiver.py:128-145).  The S = QK^T matrix and the
// probabilities never reach HBM.
//
// Layout contract (produced by vl_gemm_qkv_bf16): q,k [B,H,L,DH] bf16, q already multiplied by
// softmax_scale*log2(e); vt [B,H,DH,Lkp] = V transposed (key index contiguous).
//
// One workgroup = one (batch, head) x up to 16 query tiles of 32 rows (one wave per tile).  K and
// V^T for a chunk of 288 keys sit in LDS (74 KB -> two workgroups per CU) and are shared by all
// query waves.  Operands are swapped -- S^T = K·Q^T and O^T = V^T·P^T -- so that with the
// 32x32x16 MFMA accumulator layout every lane owns ONE query column: the softmax row reduction is
// 16 in-register values + one cross-half exchange, the running max / sum / rescale are per-lane
// scalars, and P feeds the second MFMA straight from registers (the k-slot permutation of the
// accumulator layout is absorbed by reading V^T with the same permutation).
#include "vl_common.h"
#include "vitlens_hip.h"

namespace {

constexpr int KC = 288;        // keys per LDS chunk (9 tiles of 32)
constexpr int VS = KC + 4;     // V^T LDS row stride in elements: 584 B -> conflict-free ds_read_b64

struct AttnP {
  const bf16_t *q, *k, *vt;
  bf16_t* out;   // [B, Lq, H*DH]
  float* lse;    // [B, H, Lq] (natural-log domain of the scaled scores) or null
  int B, H, Lq, Lk, Lkp, causal;
};

template <int DH>
__global__ void __launch_bounds__(576, 5) attn_fwd_kernel(const AttnP p) {
  constexpr int RB = DH * 2;          // K row bytes in LDS
  constexpr int CH = RB / 16;         // 16-byte chunks per row
  constexpr int RSH = (DH == 64) ? 1 : 2;  // rows per 256-B bank row = 2^RSH
  constexpr int KS = DH / 16;         // MFMA k-steps for S
  constexpr int DT = DH / 32;         // 32-row tiles of O^T
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sK = smem;                       // [KC][RB] swizzled
  bf16_t* sV = (bf16_t*)(smem + KC * RB);         // [DH][VS]

  const int b = blockIdx.z, h = blockIdx.y;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwq = nthr >> 6;
  const int fr = lane & 31, fg = lane >> 5;
  const size_t bh = (size_t)b * p.H + h;
  const bf16_t* Kg = p.k + bh * p.Lk * DH;
  const bf16_t* Vg = p.vt + bh * DH * (size_t)p.Lkp;

  const int q0 = (blockIdx.x * nwq + wid) * 32;
  const bool wave_active = q0 < p.Lq;
  int qrow = q0 + fr; if (qrow >= p.Lq) qrow = p.Lq - 1;
  const int qidx = q0 + fr;

  bf16x8 qf[KS];
  {
    const bf16_t* Qg = p.q + (bh * p.Lq + qrow) * DH;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const bf16x8*)(Qg + ks * 16 + fg * 8);
  }

  float m_run = -1e30f, l_run = 0.f;
  f32x16 o[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;

  const int q_hi = q0 + 31;  // last query row of this wave's tile
  const int blk_q_hi = min(p.Lq - 1, (int)(blockIdx.x * nwq + nwq) * 32 - 1);

  for (int kc0 = 0; kc0 < p.Lk; kc0 += KC) {
    if (p.causal && kc0 > blk_q_hi) break;   // uniform across the workgroup
    __syncthreads();
    // ---- stage K chunk + V^T chunk: issue 4+4 independent 16-byte loads per thread, then store ----
    // (one load->store at a time exposes the full memory latency 8x per workgroup)
    constexpr int NPK = KC * CH, NPV = DH * (KC / 8);
    for (int base = 0; base < NPK; base += 4 * nthr) {
      u32x4 kv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = base + u * nthr + tid;
        const int row = i / CH, c = i % CH;
        kv[u] = u32x4{0u, 0u, 0u, 0u};
        if (i < NPK && kc0 + row < p.Lk) kv[u] = *(const u32x4*)(Kg + (size_t)(kc0 + row) * DH + c * 8);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = base + u * nthr + tid;
        const int row = i / CH, c = i % CH;
        if (i < NPK) *(u32x4*)(sK + row * RB + ((c ^ ((row >> RSH) & (CH - 1))) * 16)) = kv[u];
      }
    }
    for (int base = 0; base < NPV; base += 4 * nthr) {
      u32x4 vv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = base + u * nthr + tid;
        const int d = i / (KC / 8), kp = i % (KC / 8);
        const int key = kc0 + kp * 8;
        vv[u] = u32x4{0u, 0u, 0u, 0u};
        if (i < NPV && key + 8 <= p.Lkp) vv[u] = *(const u32x4*)(Vg + (size_t)d * p.Lkp + key);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = base + u * nthr + tid;
        if (i >= NPV) continue;
        const int d = i / (KC / 8), kp = i % (KC / 8);
        const int key = kc0 + kp * 8;
        u32x4 v = vv[u];
        if (key + 8 > p.Lk) {  // boundary / pad piece: clear keys >= Lk (pad columns are uninitialised)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            unsigned int w = v[e];
            if (key + 2 * e >= p.Lk) w &= 0xffff0000u;
            if (key + 2 * e + 1 >= p.Lk) w &= 0x0000ffffu;
            v[e] = w;
          }
        }
        u32x2* dst = (u32x2*)(sV + d * VS + kp * 8);
        u32x2 lo = {v[0], v[1]}, hi = {v[2], v[3]};
        dst[0] = lo; dst[1] = hi;
      }
    }
    __syncthreads();
    if (!wave_active) continue;

    int ntile = (min(p.Lk - kc0, KC) + 31) >> 5;
    if (p.causal) ntile = min(ntile, ((q_hi - kc0) >> 5) + 1);
    for (int kt = 0; kt < ntile; ++kt) {
      // ---- S^T tile: rows = keys, cols = queries ----
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
      const unsigned char* kbase = sK + (kt * 32 + fr) * RB;
      const int ksw = (fr >> RSH) & (CH - 1);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 kf = *(const bf16x8*)(kbase + (((ks * 2 + fg) ^ ksw) * 16));
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
      }
      const int key0 = kc0 + kt * 32 + fg * 4;
      const bool need_mask = (key0 - fg * 4 + 32 > p.Lk) || (p.causal && key0 - fg * 4 + 31 > q0);
      if (need_mask) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = key0 + (r & 3) + 8 * (r >> 2);
          if (key >= p.Lk || (p.causal && key > qidx)) s[r] = -INFINITY;
        }
      }
      float mx = s[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      float ps = 0.f;
      float pv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) { pv[r] = __builtin_amdgcn_exp2f(s[r] - m_new); ps += pv[r]; }
      l_run = l_run * alpha + ps;
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
      // ---- O^T += V^T · P^T ----
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        bf16x8 pf;
#pragma unroll
        for (int e = 0; e < 8; ++e) pf[e] = (__bf16)pv[c * 8 + e];
#pragma unroll
        for (int t = 0; t < DT; ++t) {
          const bf16_t* vrow = sV + (t * 32 + fr) * VS + kt * 32 + c * 16 + fg * 4;
          const bf16x4 v0 = *(const bf16x4*)(vrow);
          const bf16x4 v1 = *(const bf16x4*)(vrow + 8);
          bf16x8 vf;
#pragma unroll
          for (int e = 0; e < 4; ++e) { vf[e] = v0[e]; vf[4 + e] = v1[e]; }
          o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[t], 0, 0, 0);
        }
      }
    }
  }

  if (!wave_active) return;
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  if (qidx < p.Lq) {
    bf16_t* og = p.out + ((size_t)b * p.Lq + qidx) * (p.H * DH) + h * DH;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int d = t * 32 + qd * 8 + fg * 4;
        u32x2 w;
        w[0] = pack2bf(o[t][qd * 4 + 0] * inv, o[t][qd * 4 + 1] * inv);
        w[1] = pack2bf(o[t][qd * 4 + 2] * inv, o[t][qd * 4 + 3] * inv);
        *(u32x2*)(og + d) = w;
      }
    if (p.lse && fg == 0)
      p.lse[bh * p.Lq + qidx] = (m_run + __log2f(l_tot)) * 0.6931471805599453f;
  }
}

}  // namespace

extern "C" int vl_set_error(const char* msg);

extern "C" int vl_attn_fwd_bf16(const void* q, const void* k, const void* vt, void* out, float* lse,
                                int B, int H, int Lq, int Lk, int Lkp, int dh, int causal,
                                hipStream_t stream) {
  if (B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0) return vl_set_error("vl_attn_fwd_bf16: empty problem");
  if (dh != 64 && dh != 32) return vl_set_error("vl_attn_fwd_bf16: head dim must be 32 or 64");
  if (Lkp < Lk || (Lkp & 7)) return vl_set_error("vl_attn_fwd_bf16: Lkp must be >= Lk and a multiple of 8");
  AttnP p{(const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)vt, (bf16_t*)out, lse, B, H, Lq, Lk, Lkp, causal};
  const int qtiles = (Lq + 31) / 32;
  const int nwq = qtiles <= 9 ? qtiles : 8;   // one wave per 32-query tile; 257 tokens -> 9 waves, one workgroup per (b,h)
  const int gx = (qtiles + nwq - 1) / nwq;
  const size_t smem = (size_t)KC * dh * 2 + (size_t)dh * VS * 2;
  hipError_t e;
  if (dh == 64) {
    static bool set64 = false;
    if (!set64) { e = hipFuncSetAttribute((const void*)attn_fwd_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); if (e != hipSuccess) return vl_set_error(hipGetErrorString(e)); set64 = true; }
    hipLaunchKernelGGL(attn_fwd_kernel<64>, dim3(gx, H, B), dim3(nwq * 64), smem, stream, p);
  } else {
    hipLaunchKernelGGL(attn_fwd_kernel<32>, dim3(gx, H, B), dim3(nwq * 64), smem, stream, p);
  }
  e = hipGetLastError();
  if (e != hipSuccess) return vl_set_error(hipGetErrorString(e));
  return 0;
}
