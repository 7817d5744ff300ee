// Fused attention backward for gfx950 (two kernels, no atomics, scores recomputed in registers).
//
// Autograd of F.multi_head_attention_forward / the Perceiver einsum attention
// (open_clip/transformer.py:241-252, open_clip/perceiver.py:128-145) for the trainable towers.
//   kernel A: one wave per 32 QUERIES  ->  dQ, delta = rowsum(dO * O)   (loops over key tiles)
//   kernel B: one wave per 32 KEYS     ->  dK, dV                       (loops over query tiles)
// Both recompute P = exp2(S2 - lse2) from the forward's log-sum-exp and use the same swapped-operand trick as the
// forward: the wave's own index (query in A, key in B) is the MFMA column, so P / dS live 16-per-lane in accumulator
// layout and feed the next MFMA as its column operand straight from registers, while the row operand (K^T / Q^T / dO^T)
// is read from a transposed LDS image whose row order already matches the accumulator layout (vl_attn_common.h).
//
//   inputs : q, k, v, dO, O  strided [B,H,L,dh] views - column blocks of the packed in-projection output, the
//            out-projection's input gradient and the forward's token-major output, all read in place (round 1 needed
//            head-split AND transposed copies of q, k, dO written by GEMM epilogues and a separate delta kernel);
//            q is multiplied by qscale = scale*log2(e) as it is loaded, exactly as in the forward; lse [B,H,Lq].
//   outputs: dq -> dQKV[(b*Lq+q), 0*D + h*dh + d], dk -> [.., 1*D ..], dv -> [.., 2*D ..]  (token-major, bf16);
//            delta [B,H,Lq] fp32 is produced by kernel A and consumed by kernel B (same stream).
//
// VALU trimming (the round-1 kernels were VALU/latency bound, MFMA 18-25 % busy): in B, -lse2 and -delta enter as the C
// operands of the S and dP MFMAs (four ds_read_b128 per tile from LDS; padded query rows carry -inf / 0 and need no
// masking), so the elementwise work per 32x32 tile is 16 exp2, 16 multiplies and the bf16 packing.  A sits at the
// 128-register cap of its two-workgroups-per-CU layout and has no room for the two constant C blocks: it subtracts
// lse2 / delta with packed fp32 operations (v_pk_add_f32 / v_pk_mul_f32, two elements per instruction).
//
// Head dims 72..128 (ViT-H/14: 80, ViT-bigG/14: 104) run in DH = 128 instantiations, zero-padded inside LDS / registers;
// only the real columns are read and written.  B then walks its four output d-tiles in two passes (S, dP recomputed).
//
// L = 257 (8 whole tiles + one row): the lone last query (A) / key (B) is not given a ninth wave; the 8 waves share it
// alongside their own tiles - wave w takes tile w with the MFMA operand roles swapped (the lone row is row 0 of the A
// operand), pushes p / dS through a 128-byte LDS scratch to make them an A operand, and the per-wave partial dQ (or
// dK, dV) rows are summed in a fixed order through LDS.
#include "vl_attn_common.h"
#include "vitlens_hip.h"

namespace {
using namespace vlattn;

constexpr int NC = 288;        // queries per LDS chunk of kernel B: 9 tiles of 32
constexpr int NCA = 160;       // keys per LDS chunk of kernel A: 5 tiles (two workgroups per CU)
constexpr int NWMAX = 8;
constexpr float LOG2E = 1.4426950408889634f;

struct AttnBwdP {
  TV q, k, v, dO, o;
  const float* lse;
  float* delta;
  bf16_t *dq, *dk, *dv;     // token-major destinations (already offset to the q / k / v column block)
  long ld_dq, ld_dkv;       // row strides (elements) of the dq and dk/dv destinations
  int B, H, Lq, Lk, causal;
  int dh;                   // real head dim (= DH, or 72..128 in the DH = 128 instantiation)
  float qscale;             // q is multiplied by this at load (softmax scale * log2 e)
  float scale;              // softmax scale (dq = scale * dS K ; dk = ln2 * dS^T Q2)
  VL_PROF_FIELD
  int l_main;               // queries (A) / keys (B) handled by per-wave tiles (L, or L-1 when the last row is shared)
};

__device__ __forceinline__ bf16x8 load_frag(const bf16_t* row, int ks, int fg, float scale) {
  u32x4 raw = *(const u32x4*)(row + ks * 16 + fg * 8);
  if (scale != 1.0f) raw = scale_bf16x8(raw, scale);
  return __builtin_bit_cast(bf16x8, raw);
}

// the 8 values of an accumulator-layout row vector that lane (fr == 0, fg) of an A operand needs for slice c
__device__ __forceinline__ bf16x8 gather_row0(const float* scratch, int c, int fr, int fg) {
  bf16x8 r = zero_bf8();
  if (fr == 0) {
    const f32x4 lo = *(const f32x4*)(scratch + c * 16 + fg * 4);
    const f32x4 hi = *(const f32x4*)(scratch + c * 16 + 8 + fg * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { r[e] = (__bf16)lo[e]; r[4 + e] = (__bf16)hi[e]; }
  }
  return r;
}
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------------------------------- kernel A: dQ (+ delta)
// NCA keys per LDS chunk.  NCA = 160 (62 KB of LDS, <= 128 VGPRs) puts TWO workgroups on a CU, so one workgroup's staging
// (a memory round trip that is bandwidth-bound chip-wide: every workgroup asks for its 40 KB at once) overlaps the
// other's MFMA work; measured against the single-chunk 1-workgroup-per-CU version (load + compute strictly serial).
template <int DH, int NCA, bool TAILQ>
__global__ void __launch_bounds__(NWMAX * 64, DH == 128 ? 2 : 4) attn_bwd_dq_kernel(const AttnBwdP p) {
  constexpr int RB = DH * 2, KS = DH / 16, DT = DH / 32, TS = NCA + 8, CH = DH / 8;
  constexpr bool PAD = DH == 128;                 // head dims 72..128 run zero-padded to 128
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sK = smem;
  unsigned char* sV = smem + NCA * RB;
  bf16_t* sKt = (bf16_t*)(smem + 2 * NCA * RB);
  float* sP = (float*)(sKt + DH * TS);            // [NWMAX][32]  (TAILQ)
  float* sPart = sP + NWMAX * 32;                 // [NWMAX][DH]  (TAILQ)
  bf16_t* sTail = (bf16_t*)(sPart + NWMAX * DH);  // [3][DH] the shared row of q, dO, O, fetched with the first chunk (TAILQ)

  const int b = blockIdx.z, h = blockIdx.y;
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6), nwq = nthr >> 6;
  const int fr = lane & 31, fg = lane >> 5;
  const size_t bh = (size_t)b * p.H + h;
  const bf16_t* Qb = p.q.p + b * p.q.sb + h * p.q.sh;
  const bf16_t* Kb = p.k.p + b * p.k.sb + h * p.k.sh;
  const bf16_t* Vb = p.v.p + b * p.v.sb + h * p.v.sh;
  const bf16_t* Gb = p.dO.p + b * p.dO.sb + h * p.dO.sh;
  const bf16_t* Ob = p.o.p + b * p.o.sb + h * p.o.sh;
  const int q0 = (blockIdx.x * nwq + wid) * 32;
  const bool active = q0 < p.l_main;
  const int qidx = q0 + fr;
  const int qrow = qidx < p.l_main ? qidx : p.l_main - 1;
  const int dhr = PAD ? p.dh : DH, nch = dhr >> 3;        // real head dim, valid 16-byte chunks per operand row

  VL_PROF_STAMP(p, 0);
  // one memory round trip for everything the workgroup needs first: the per-lane q / dO / O rows and the log-sum-exp
  // are requested BEFORE the first chunk is staged and consumed after it (loads return in order)
  u32x4 qraw[KS], graw[KS], oraw[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const u32x4 z = {0u, 0u, 0u, 0u};
    qraw[ks] = z; graw[ks] = z; oraw[ks] = z;
    if (!PAD || ks * 2 + fg < nch) {
      qraw[ks] = *(const u32x4*)(Qb + (long)qrow * p.q.sr + ks * 16 + fg * 8);
      graw[ks] = *(const u32x4*)(Gb + (long)qrow * p.dO.sr + ks * 16 + fg * 8);
      oraw[ks] = *(const u32x4*)(Ob + (long)qrow * p.o.sr + ks * 16 + fg * 8);
    }
  }
  const float lse_raw = p.lse[bh * p.Lq + qrow];
  [[maybe_unused]] const float lse_tail = TAILQ ? p.lse[bh * p.Lq + p.Lq - 1] : 0.f;   // (uniform: a scalar load)
  [[maybe_unused]] u32x4 tailraw = {0u, 0u, 0u, 0u};
  if constexpr (TAILQ) {
    if (tid < 3 * CH && tid % CH < nch) {
      const int m = tid / CH, c = tid % CH;
      const bf16_t* src = m == 0 ? Qb + (long)(p.Lq - 1) * p.q.sr : (m == 1 ? Gb + (long)(p.Lq - 1) * p.dO.sr : Ob + (long)(p.Lq - 1) * p.o.sr);
      tailraw = *(const u32x4*)(src + c * 8);
    }
  }
  stage2<DH, NCA, true, true, true, false, 2>(StageSrc{sK, sKt, Kb, p.k.sr, 1.f, nch}, StageSrc{sV, nullptr, Vb, p.v.sr, 1.f, nch},
                                              0, p.Lk, tid, nthr);
  if constexpr (TAILQ) {
    if (tid < 3 * CH) *(u32x4*)(sTail + tid * 8) = tailraw;
  }
  VL_PROF_STAMP(p, 1);
  bf16x8 qf[KS], dof[KS];
  float dlt = 0.f;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    qf[ks] = __builtin_bit_cast(bf16x8, p.qscale != 1.0f ? scale_bf16x8(qraw[ks], p.qscale) : qraw[ks]);
    dof[ks] = __builtin_bit_cast(bf16x8, graw[ks]);
    dlt = dot8(dof[ks], __builtin_bit_cast(bf16x8, oraw[ks]), dlt);
  }
  dlt = xhalf_sum(dlt);
  if (active && fg == 0 && qidx < p.l_main) p.delta[bh * p.Lq + qidx] = dlt;
  const float lse2 = lse_raw * LOG2E;

  f32x16 dq[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t) dq[t] = zero16();
  if constexpr (TAILQ) {                           // the shared last query's partial rows accumulate in LDS (per wave)
    if (fg == 0) {
#pragma unroll
      for (int t = 0; t < DT; ++t) sPart[wid * DH + t * 32 + fr] = 0.f;
    }
  }

  const int blk_q_hi = min(p.l_main - 1, (int)(blockIdx.x * nwq + nwq) * 32 - 1);
  for (int kc0 = 0; kc0 < p.Lk; kc0 += NCA) {
    if (p.causal && kc0 > blk_q_hi && !TAILQ) break;
    if (kc0 > 0) {
      __syncthreads();
      // (fewer loads in flight here: the accumulators are live)
      stage2<DH, NCA, true, true, true, false, 1>(StageSrc{sK, sKt, Kb, p.k.sr, 1.f, nch}, StageSrc{sV, nullptr, Vb, p.v.sr, 1.f, nch},
                                                  kc0, p.Lk, tid, nthr);
    }
    __syncthreads();
    VL_PROF_STAMP(p, 2);
    const int ctile = (min(p.Lk - kc0, NCA) + 31) >> 5;
    if (active) {
      int ntile = ctile;
      if (p.causal) ntile = min(ntile, ((q0 + 31 - kc0) >> 5) + 1);
      for (int kt = 0; kt < ntile; ++kt) {
        f32x16 s = zero16(), dp = zero16();
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<DH>(sK, kt * 32 + fr, ks, fg), qf[ks], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<DH>(sV, kt * 32 + fr, ks, fg), dof[ks], dp, 0, 0, 0);
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
        float ds[16];
#pragma unroll
        for (int r = 0; r < 16; r += 2) {          // packed adds / multiplies: (s - lse2), (dp - delta), p * (dp - delta)
          vl_f32x2 sv = {s[r], s[r + 1]}, dv = {dp[r], dp[r + 1]};
          sv -= lse2; dv -= dlt;
          const vl_f32x2 pv = {__builtin_amdgcn_exp2f(sv[0]), __builtin_amdgcn_exp2f(sv[1])};
          dv *= pv;
          ds[r] = dv[0]; ds[r + 1] = dv[1];
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const bf16x8 df = pack8(ds + c * 8);
#pragma unroll
          for (int t = 0; t < DT; ++t)
            dq[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_t<NCA>(sKt, t * 32 + fr, kt, c, fg), df, dq[t], 0, 0, 0);
        }
      }
    }
    if constexpr (TAILQ) {
      // ---- the shared last query row against key tile(s) wid, wid + 8, ... of this chunk (lane = key; the row sees
      //      every key: host guarantees no causal cut) ----
      if (wid < ctile) {
        // delta of the row: lane d multiplies dO[d] * O[d], summed over the wave
        float dloc = 0.f;
        for (int d = lane; d < DH; d += 64) dloc = fmaf(bf2f(sTail[DH + d]), bf2f(sTail[2 * DH + d]), dloc);
        const float dT = wave_sum_dpp(dloc);
        if (kc0 == 0 && wid == 0 && lane == 0) p.delta[bh * p.Lq + p.Lq - 1] = dT;
        const float lT = lse_tail * LOG2E;
        float* myP = sP + wid * 32;
        for (int kt = wid; kt < ctile; kt += nwq) {
          // A operands: row 0 = the row of q / dO (lanes fr == 0), rows 1..31 zero; S then dP, one accumulator at a time
          f32x16 acc = zero16();
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr == 0 ? load_frag(sTail, ks, fg, p.qscale) : zero_bf8(),
                                                          frag_rows<DH>(sK, kt * 32 + fr, ks, fg), acc, 0, 0, 0);
          const float s0 = acc[0];
          acc = zero16();
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr == 0 ? load_frag(sTail + DH, ks, fg, 1.f) : zero_bf8(),
                                                          frag_rows<DH>(sV, kt * 32 + fr, ks, fg), acc, 0, 0, 0);
          // row 0 of D = slot 0 of the lanes with fg == 0; lane fr <-> key kc0 + kt*32 + fr
          const bool valid = fg == 0 && kc0 + kt * 32 + fr < p.Lk;
          const float dsv = valid ? __builtin_amdgcn_exp2f(s0 - lT) * (acc[0] - dT) : 0.f;
          wave_lds_sync();
          if (fg == 0) myP[fr] = dsv;
          wave_lds_sync();
#pragma unroll
          for (int t = 0; t < DT; ++t) {
            acc = zero16();
#pragma unroll
            for (int c = 0; c < 2; ++c)
              acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gather_row0(myP, c, fr, fg), frag_t<NCA>(sKt, t * 32 + fr, kt, c, fg),
                                                            acc, 0, 0, 0);
            if (fg == 0) sPart[wid * DH + t * 32 + fr] += acc[0];
          }
        }
      }
    }
  }
  VL_PROF_STAMP(p, 3);
  if (active)
    store_rows_t<DT>(dq, p.scale, p.dq + ((size_t)b * p.Lq + qrow) * p.ld_dq + h * dhr, fg, qidx < p.l_main, nch);
  VL_PROF_STAMP(p, 4);
  if constexpr (TAILQ) {
    __syncthreads();
    if (wid == 0) {
      for (int d = lane; d < dhr; d += 64) {
        float acc = 0.f;
        for (int w = 0; w < nwq; ++w) acc += sPart[w * DH + d];
        p.dq[((size_t)b * p.Lq + p.Lq - 1) * p.ld_dq + h * dhr + d] = f2bf(acc * p.scale);
      }
    }
  }
  VL_PROF_STAMP(p, 5);
}

// ------------------------------------------------------------------------------------- kernel B: dK, dV
// NCB queries per LDS chunk (288 = the whole 257-token sequence; 96 for the padded 128-wide head dims, whose images are
// twice as large).  The padded instantiation also walks its 4 output d-tiles in two passes of TPP = 2 (dK, dV, K, V for 128
// columns would need 256 accumulator / operand registers): S and dP are recomputed per pass.
template <int DH, int NCB, bool TAILK>
__global__ void __launch_bounds__(NWMAX * 64, 2) attn_bwd_dkv_kernel(const AttnBwdP p) {
  constexpr int RB = DH * 2, KS = DH / 16, DT = DH / 32, TS = NCB + 8;
  constexpr bool PAD = DH == 128;
  constexpr int TPP = PAD ? 2 : DT;              // output d-tiles per pass
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sQ = smem;
  unsigned char* sdO = smem + NCB * RB;
  bf16_t* sQt = (bf16_t*)(smem + 2 * NCB * RB);
  bf16_t* sdOt = sQt + DH * TS;
  float* sNegL = (float*)(sdOt + DH * TS);        // -lse2 per query of the chunk (-inf on padded rows)
  float* sNegD = sNegL + NCB;                      // -delta                       (0 on padded rows)
  float* sP = sNegD + NCB;                         // [NWMAX][2][32]   (TAILK)
  float* sPart = sP + NWMAX * 64;                 // [NWMAX][2*DH]    (TAILK)
  bf16_t* sTail = (bf16_t*)(sPart + NWMAX * 2 * DH);   // [2][DH] the shared key's K and V rows, fetched up front (TAILK)
  constexpr int CH = DH / 8;

  const int b = blockIdx.z, h = blockIdx.y;
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6), nwk = nthr >> 6;
  const int fr = lane & 31, fg = lane >> 5;
  const size_t bh = (size_t)b * p.H + h;
  const bf16_t* Qb = p.q.p + b * p.q.sb + h * p.q.sh;
  const bf16_t* Kb = p.k.p + b * p.k.sb + h * p.k.sh;
  const bf16_t* Vb = p.v.p + b * p.v.sb + h * p.v.sh;
  const bf16_t* Gb = p.dO.p + b * p.dO.sb + h * p.dO.sh;
  const int k0 = (blockIdx.x * nwk + wid) * 32;
  const bool active = k0 < p.l_main;
  const int kidx = k0 + fr;
  const int krow = kidx < p.l_main ? kidx : p.l_main - 1;
  const int dhr = PAD ? p.dh : DH, nch = dhr >> 3;        // real head dim, valid 16-byte chunks per operand row

  VL_PROF_STAMP(p, 0);
  [[maybe_unused]] u32x4 tailraw = {0u, 0u, 0u, 0u};
  if constexpr (TAILK) {
    if (tid < 2 * CH && tid % CH < nch)
      tailraw = *(const u32x4*)((tid < CH ? Kb + (long)(p.Lk - 1) * p.k.sr : Vb + (long)(p.Lk - 1) * p.v.sr) + (tid % CH) * 8);
  }
  bf16x8 kf[KS], vf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    kf[ks] = zero_bf8(); vf[ks] = zero_bf8();
    if (!PAD || ks * 2 + fg < nch) {
      kf[ks] = load_frag(Kb + (long)krow * p.k.sr, ks, fg, 1.f);
      vf[ks] = load_frag(Vb + (long)krow * p.v.sr, ks, fg, 1.f);
    }
  }
  const float ln2 = 0.6931471805599453f;
#pragma unroll 1
  for (int tp = 0; tp < DT; tp += TPP) {         // one pass for head dims up to 64
  f32x16 dk[TPP], dv[TPP];
#pragma unroll
  for (int t = 0; t < TPP; ++t) { dk[t] = zero16(); dv[t] = zero16(); }
  [[maybe_unused]] float dkT[TPP], dvT[TPP];      // the shared last key's partial rows: slot 0 of per-tile MFMAs (TAILK)
#pragma unroll
  for (int t = 0; t < TPP; ++t) { dkT[t] = 0.f; dvT[t] = 0.f; }

  for (int qc0 = 0; qc0 < p.Lq; qc0 += NCB) {
    __syncthreads();
    // (log-sum-exp / delta of the chunk requested before the staging loads: one round trip)
    float lraw[(NCB + NWMAX * 64 - 1) / (NWMAX * 64)], draw[(NCB + NWMAX * 64 - 1) / (NWMAX * 64)];
    if (nthr == NWMAX * 64) {
#pragma unroll
      for (int j = 0; j < (NCB + NWMAX * 64 - 1) / (NWMAX * 64); ++j) {
        const int i = tid + j * NWMAX * 64;
        const bool ok = i < NCB && qc0 + i < p.Lq;
        lraw[j] = ok ? p.lse[bh * p.Lq + qc0 + i] : INFINITY;
        draw[j] = ok ? p.delta[bh * p.Lq + qc0 + i] : 0.f;
      }
    }
    stage2<DH, NCB, true, true, true, true>(StageSrc{sQ, sQt, Qb, p.q.sr, p.qscale, nch}, StageSrc{sdO, sdOt, Gb, p.dO.sr, 1.f, nch},
                                           qc0, p.Lq, tid, nthr);
    if constexpr (TAILK) {
      if (qc0 == 0 && tid < 2 * CH) *(u32x4*)(sTail + tid * 8) = tailraw;
    }
    if (nthr == NWMAX * 64) {
#pragma unroll
      for (int j = 0; j < (NCB + NWMAX * 64 - 1) / (NWMAX * 64); ++j) {
        const int i = tid + j * NWMAX * 64;
        if (i < NCB) { sNegL[i] = -lraw[j] * LOG2E; sNegD[i] = -draw[j]; }
      }
    } else {
      for (int i = tid; i < NCB; i += nthr) {
        const bool ok = qc0 + i < p.Lq;
        sNegL[i] = ok ? -p.lse[bh * p.Lq + qc0 + i] * LOG2E : -INFINITY;
        sNegD[i] = ok ? -p.delta[bh * p.Lq + qc0 + i] : 0.f;
      }
    }
    VL_PROF_STAMP(p, 1);
    __syncthreads();
    VL_PROF_STAMP(p, 2);
    const int ntile = (min(p.Lq - qc0, NCB) + 31) >> 5;
    if (active) {
      int t0 = 0;
      if (p.causal) t0 = max(0, (k0 - qc0) >> 5);      // queries before this key tile never see it
      for (int qt = t0; qt < ntile; ++qt) {
        // rows of the accumulator = queries (r&3) + 8*(r>>2) + 4*fg of this tile; column = this lane's key
        f32x16 s, dp;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const f32x4 l4 = *(const f32x4*)(sNegL + qt * 32 + qd * 8 + fg * 4);
          const f32x4 d4 = *(const f32x4*)(sNegD + qt * 32 + qd * 8 + fg * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { s[qd * 4 + e] = l4[e]; dp[qd * 4 + e] = d4[e]; }
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<DH>(sQ, qt * 32 + fr, ks, fg), kf[ks], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<DH>(sdO, qt * 32 + fr, ks, fg), vf[ks], dp, 0, 0, 0);
        }
        // (fragments of the transposed images requested ahead of the exponentials - except in the padded instantiation,
        //  which has no registers to hold them)
        [[maybe_unused]] bf16x8 gtf[2][TPP], qtf[2][TPP];
        if constexpr (!PAD) {
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t = 0; t < TPP; ++t) {
              gtf[c][t] = frag_t<NCB>(sdOt, (tp + t) * 32 + fr, qt, c, fg);
              qtf[c][t] = frag_t<NCB>(sQt, (tp + t) * 32 + fr, qt, c, fg);
            }
        }
        if (p.causal && qc0 + qt * 32 < k0 + 31) {      // diagonal tile: key > query is masked
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int qg = qc0 + qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fg;
            if (kidx > qg) s[r] = -INFINITY;
          }
        }
        float pv[16], ds[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { pv[r] = __builtin_amdgcn_exp2f(s[r]); ds[r] = pv[r] * dp[r]; }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const bf16x8 pf = pack8(pv + c * 8), df = pack8(ds + c * 8);
#pragma unroll
          for (int t = 0; t < TPP; ++t) {
            dv[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PAD ? frag_t<NCB>(sdOt, (tp + t) * 32 + fr, qt, c, fg) : gtf[c][t], pf, dv[t], 0, 0, 0);
            dk[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PAD ? frag_t<NCB>(sQt, (tp + t) * 32 + fr, qt, c, fg) : qtf[c][t], df, dk[t], 0, 0, 0);
          }
        }
      }
    }
    VL_PROF_STAMP(p, 3);
    if constexpr (TAILK) {
      // ---- the shared last key against query tile(s) wid, wid + 8, ... (lane = query) ----
      const int kT = p.Lk - 1;
      float* myP = sP + wid * 64;
      for (int qt = wid; qt < ntile; qt += nwk) {
        if (p.causal && qc0 + qt * 32 + 31 < kT) continue;
        // A operands: row 0 = the key / value row (lanes fr == 0), rows 1..31 zero; S then dP through one accumulator
        f32x16 acc = zero16();
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr == 0 ? load_frag(sTail, ks, fg, 1.f) : zero_bf8(),
                                                        frag_rows<DH>(sQ, qt * 32 + fr, ks, fg), acc, 0, 0, 0);
        const float s0 = acc[0];
        acc = zero16();
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr == 0 ? load_frag(sTail + DH, ks, fg, 1.f) : zero_bf8(),
                                                        frag_rows<DH>(sdO, qt * 32 + fr, ks, fg), acc, 0, 0, 0);
        const float dp0 = acc[0];
        float pT = 0.f, dsT = 0.f;
        if (fg == 0 && !(p.causal && qc0 + qt * 32 + fr < kT)) {
          pT = __builtin_amdgcn_exp2f(s0 + sNegL[qt * 32 + fr]);         // padded rows: -inf -> 0
          dsT = pT * (dp0 + sNegD[qt * 32 + fr]);
        }
        wave_lds_sync();
        if (fg == 0) { myP[fr] = pT; myP[32 + fr] = dsT; }
        wave_lds_sync();
#pragma unroll
        for (int t = 0; t < TPP; ++t) {
          acc = zero16();
#pragma unroll
          for (int c = 0; c < 2; ++c)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gather_row0(myP, c, fr, fg), frag_t<NCB>(sdOt, (tp + t) * 32 + fr, qt, c, fg), acc, 0, 0, 0);
          dvT[t] += acc[0];
          acc = zero16();
#pragma unroll
          for (int c = 0; c < 2; ++c)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gather_row0(myP + 32, c, fr, fg), frag_t<NCB>(sQt, (tp + t) * 32 + fr, qt, c, fg), acc, 0, 0, 0);
          dkT[t] += acc[0];
        }
      }
    }
  }
  VL_PROF_STAMP(p, 4);
  if (active) {
    store_rows_t<TPP>(dk, ln2, p.dk + ((size_t)b * p.Lk + krow) * p.ld_dkv + h * dhr + tp * 32, fg, kidx < p.l_main, nch - tp * 4);
    store_rows_t<TPP>(dv, 1.f, p.dv + ((size_t)b * p.Lk + krow) * p.ld_dkv + h * dhr + tp * 32, fg, kidx < p.l_main, nch - tp * 4);
  }
  if constexpr (TAILK) {
    if (fg == 0) {
#pragma unroll
      for (int t = 0; t < TPP; ++t) {
        sPart[wid * 2 * DH + (tp + t) * 32 + fr] = dvT[t];
        sPart[wid * 2 * DH + DH + (tp + t) * 32 + fr] = dkT[t];
      }
    }
  }
  }   // passes over the output d-tiles
  if constexpr (TAILK) {
    __syncthreads();
    if (wid == 0) {
      for (int d = lane; d < dhr; d += 64) {
        float av = 0.f, ak = 0.f;
        for (int w = 0; w < nwk; ++w) { av += sPart[w * 2 * DH + d]; ak += sPart[w * 2 * DH + DH + d]; }
        const size_t row = ((size_t)b * p.Lk + p.Lk - 1) * p.ld_dkv + h * dhr + d;
        p.dv[row] = f2bf(av);
        p.dk[row] = f2bf(ak * ln2);
      }
    }
  }
  VL_PROF_STAMP(p, 5);
}

}  // namespace

extern "C" int vl_set_error(const char* msg);

template <int DH, bool TQ, bool TK>
static int launch_bwd(const AttnBwdP& pin, int lq_main, int lk_main, hipStream_t stream) {
  constexpr int NCB = DH == 128 ? 96 : NC;
  const size_t smA = (size_t)2 * NCA * DH * 2 + (size_t)DH * (NCA + 8) * 2 + (size_t)NWMAX * 32 * 4 + (size_t)NWMAX * DH * 4 +
                     (size_t)3 * DH * 2;
  const size_t smB = (size_t)2 * NCB * DH * 2 + (size_t)2 * DH * (NCB + 8) * 2 + (size_t)2 * NCB * 4 + (size_t)NWMAX * 64 * 4 +
                     (size_t)NWMAX * 2 * DH * 4 + (size_t)2 * DH * 2;
  static const hipError_t attrA = hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<DH, NCA, TQ>,
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smA);
  static const hipError_t attrB = hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<DH, NCB, TK>,
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smB);
  if (attrA != hipSuccess) return vl_set_error(hipGetErrorString(attrA));
  if (attrB != hipSuccess) return vl_set_error(hipGetErrorString(attrB));
  AttnBwdP p = pin;
  const int qtiles = (lq_main + 31) / 32, ktiles = (lk_main + 31) / 32;
  const int nwq = qtiles < NWMAX ? qtiles : NWMAX, nwk = ktiles < NWMAX ? ktiles : NWMAX;
  p.l_main = lq_main;
  hipLaunchKernelGGL((attn_bwd_dq_kernel<DH, NCA, TQ>), dim3((qtiles + nwq - 1) / nwq, p.H, p.B), dim3(nwq * 64), smA, stream, p);
  p.l_main = lk_main;
#ifdef VL_ATTN_PROF
  if (p.prof) p.prof += (size_t)p.B * p.H * ((qtiles + nwq - 1) / nwq) * 8;     // kernel B's stamps follow kernel A's
#endif
  hipLaunchKernelGGL((attn_bwd_dkv_kernel<DH, NCB, TK>), dim3((ktiles + nwk - 1) / nwk, p.H, p.B), dim3(nwk * 64), smB, stream, p);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : vl_set_error(hipGetErrorString(e));
}

extern "C" int vl_attn_bwd_bf16(const void* q, const void* k, const void* v, const void* dO, const void* o,
                                const long* strides, const float* lse, float* delta, void* dq, void* dk, void* dv,
                                long ld_dq, long ld_dkv, int B, int H, int Lq, int Lk, int dh, float qscale, int causal,
                                float scale, hipStream_t stream) {
  if (B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0) return vl_set_error("vl_attn_bwd_bf16: empty problem");
  if (dh != 32 && dh != 64 && !(dh > 64 && dh <= 128 && (dh & 7) == 0))
    return vl_set_error("vl_attn_bwd_bf16: head dim must be 32, 64, or a multiple of 8 in (64, 128] (run zero-padded to 128)");
  if (!strides || !delta || !lse) return vl_set_error("vl_attn_bwd_bf16: strides, lse and the delta workspace are required");
  for (int i = 0; i < 15; ++i)
    if (strides[i] & 7) return vl_set_error("vl_attn_bwd_bf16: operand strides must be multiples of 8 elements (16-byte rows)");
  if ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v) | ((uintptr_t)dO) | ((uintptr_t)o)) & 15)
    return vl_set_error("vl_attn_bwd_bf16: operands must be 16-byte aligned");
  if ((((uintptr_t)dq) | ((uintptr_t)dk) | ((uintptr_t)dv)) & 15 || (ld_dq & 7) || (ld_dkv & 7))
    return vl_set_error("vl_attn_bwd_bf16: gradient destinations must be 16-byte aligned with row strides multiple of 8");
  const long* s = strides;
  AttnBwdP p{TV{(const bf16_t*)q, s[0], s[1], s[2]},   TV{(const bf16_t*)k, s[3], s[4], s[5]},
             TV{(const bf16_t*)v, s[6], s[7], s[8]},   TV{(const bf16_t*)dO, s[9], s[10], s[11]},
             TV{(const bf16_t*)o, s[12], s[13], s[14]}, lse, delta, (bf16_t*)dq, (bf16_t*)dk, (bf16_t*)dv,
             ld_dq, ld_dkv, B, H, Lq, Lk, causal, dh, qscale, scale,
#ifdef VL_ATTN_PROF
             vl_attn_prof_buf,
#endif
             0};
  // one row beyond whole tiles (257 tokens): shared by the 8 waves of the single workgroup instead of a ninth wave
  const bool tq = (Lq % 32 == 1) && Lq > 32 && Lq - 1 <= NWMAX * 32 && (!causal || Lk <= Lq);
  const bool tk = (Lk % 32 == 1) && Lk > 32 && Lk - 1 <= NWMAX * 32;      // (the query chunks are walked inside the kernel)
  const int lqm = tq ? Lq - 1 : Lq, lkm = tk ? Lk - 1 : Lk;
#define VL_BWD(DHV)                                                                       \
  (tq ? (tk ? launch_bwd<DHV, true, true>(p, lqm, lkm, stream) : launch_bwd<DHV, true, false>(p, lqm, lkm, stream)) \
      : (tk ? launch_bwd<DHV, false, true>(p, lqm, lkm, stream) : launch_bwd<DHV, false, false>(p, lqm, lkm, stream)))
  return dh == 64 ? VL_BWD(64) : (dh == 32 ? VL_BWD(32) : VL_BWD(128));
#undef VL_BWD
}
