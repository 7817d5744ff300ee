// Fused (flash-style) attention forward for gfx950.
//
// Replaces F.multi_head_attention_forward inside nn.MultiheadAttention
// (open_clip/transformer.py:241-252; additive causal mask :870-876) and the einsum attention of
// the Perceiver "Lens" (open_clip/perceiver.py:128-145).  The S = QK^T matrix and the
// probabilities never reach HBM.
//
// Layout contract: q, k, v are strided [B,H,L,DH] views (vl_attn_common.h) -- normally column blocks of the packed
// in-projection output [tokens, 3*width]; q is multiplied by qscale = softmax_scale*log2(e) when it is loaded (the
// reference scales q in bf16 too, functional.py via transformer.py:241).  (Round 1 took head-split copies and V
// transposed from the GEMM epilogue: 2-byte scattered stores that doubled the in-projection's epilogue time; the
// transposition now happens while V is staged into LDS.)
//
// One workgroup = one (batch, head) x up to 8 query tiles of 32 rows (one wave per tile), two workgroups per CU
// (<= 80 KB LDS, <= 128 VGPRs) so that one workgroup's staging overlaps the other's MFMA/VALU work.  K and V^T for a
// chunk of 288 keys sit in LDS and are shared by all query waves.  Operands are swapped -- S^T = K.Q^T and
// O^T = V^T.P^T -- so that with the 32x32x16 MFMA accumulator layout every lane owns ONE query column: the softmax row
// reduction is 16 in-register values + one cross-half exchange, the running max / sum are per-lane scalars, and P feeds
// the second MFMA straight from registers.  The k-slot permutation of the accumulator layout is absorbed by the ORDER
// in which V^T's key index is laid out in LDS (bits 2 and 3 of the key index swapped), so every V^T fragment is one
// ds_read_b128.
//
// The kernel is VALU-bound (16 v_exp_f32 per lane per 32x32 tile against 8 MFMAs), so the softmax arithmetic is trimmed
// to the exponentials: the running maximum enters as the MFMA's C operand (a register block holding -m), p = exp2(s)
// needs no subtraction, and the block / the output accumulators are rescaled only when the maximum grew by more than
// 2^8 (wave-uniform branch; p <= 256 in between, exact in fp32 / bf16 range).
//
// L = 257 (ViT-L/14 at 224: 8 full tiles + ONE row): the lone last query would cost a ninth wave nine tile passes for
// one row.  Instead the 8 waves finish their own tiles and then share it: wave w computes the row's scores against key
// tile w with the roles of the MFMA operands swapped (lane = key), reduces across lanes, multiplies by V through one
// more MFMA and the 9 partial (max, sum, O) triples are merged through LDS.
#include "vl_attn_common.h"
#include "vitlens_hip.h"

namespace {
using namespace vlattn;

constexpr int KC = 288;        // keys per LDS chunk (9 tiles of 32)
constexpr int VSP = KC + 8;    // V^T LDS row stride in elements: 592 B = 16 B x odd -> conflict-free ds_read_b128
constexpr int NWMAX = 8;       // query waves per workgroup
constexpr float RESCALE_THR = 8.0f;

struct AttnP {
  TV q, k, v;
  float qscale;  // applied to q at load (1 = already scaled)
  bf16_t* out;   // [B, Lq, H*DH]
  float* lse;    // [B, H, Lq] (natural-log domain of the scaled scores) or null
  int B, H, Lq, Lk, causal;
  int dh;        // real head dim (= DH, or 72..128 in the DH = 128 instantiation)
  VL_PROF_FIELD
  int lq_main;   // queries handled by the per-wave tiles (Lq, or Lq-1 when the last row is shared)
};

// MULTI: more keys than one LDS chunk (the chunk loop restages inside the accumulation; kept out of the common
// single-chunk instantiations, where its 12 loads in flight would push the tile loop's registers to scratch)
// F16: q, k, v and the output are IEEE half (vl_attn_fwd_f16; the frozen text tower), P is rounded to half for the P.V product
// DMA (round 6; head dim 64, one chunk, bf16 - the ViT towers' and the Perceiver latents' self-attention): K and V rows go
// HBM -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`), both as ROW images; the V^T fragments of the P.V product come out of
// gfx950's transpose read (`ds_read_b64_tr_b16`, the fused backward's addressing) instead of a transposed image that every
// thread built with 2-byte LDS stores from 12 loads held in registers.  The phase timeline of round 5 had 11.4 k of a
// workgroup's 33.2 k cycles in "load + stage" (profiles/r05_attn_phase_timeline.log): with the DMA the staging is the
// memory round trip and nothing else.  The shared last row's per-wave share runs BEFORE the tile loop and the wave that
// merges the partials does so after its own tiles, without a workgroup barrier at the end of the kernel.
template <int DH, bool TAILQ, bool MULTI, bool F16 = false, bool DMA = false>
__global__ void __launch_bounds__(NWMAX * 64, DH == 128 ? 2 : 4) attn_fwd_kernel(const AttnP p) {
  static_assert(!F16 || (!TAILQ && DH == 64), "half operands: head dim 64, no shared last row (the text tower's shape)");
  static_assert(!DMA || (DH == 64 && !MULTI && !F16), "LDS-DMA staging: head dim 64, one key chunk, bf16");
  constexpr int RB = DH * 2;          // K row bytes in LDS
  constexpr int CH = RB / 16;         // 16-byte chunks per row
  constexpr int RSH = Rsh<DH>::v;     // rows per 256-B bank row = 2^RSH
  constexpr bool PAD = DH == 128;     // head dims 72..128 run zero-padded to 128
  constexpr int KS = DH / 16;         // MFMA k-steps for S
  constexpr int DT = DH / 32;         // 32-row tiles of O^T
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sK = smem;                       // [KC][RB] swizzled
  bf16_t* sV = (bf16_t*)(smem + KC * RB);         // [DH][VSP], key order permuted inside 16-key slices
  float* sP = (float*)(sV + DH * VSP);            // [NWMAX][32]   shared-row probabilities (TAILQ)
  float* sPart = sP + NWMAX * 32;                 // [KC/32][2+DH] shared-row partials      (TAILQ)
  bf16_t* sTailQ = (bf16_t*)(sPart + (KC / 32) * (2 + DH));   // [DH] the shared row of q, fetched with the chunk (TAILQ)

  const int b = blockIdx.z, h = blockIdx.y;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwq = nthr >> 6;
  const int fr = lane & 31, fg = lane >> 5;
  const size_t bh = (size_t)b * p.H + h;
  const bf16_t* Kg = p.k.p + b * p.k.sb + h * p.k.sh;
  const bf16_t* Vg = p.v.p + b * p.v.sb + h * p.v.sh;
  const bf16_t* Qb = p.q.p + b * p.q.sb + h * p.q.sh;

  const int q0 = (blockIdx.x * nwq + wid) * 32;
  const bool wave_active = q0 < p.lq_main;
  int qrow = q0 + fr; if (qrow >= p.lq_main) qrow = p.lq_main - 1;
  const int qidx = q0 + fr;
  const int dhr = PAD ? p.dh : DH;          // real head dim: output row stride and column count
  const int nch = dhr >> 3;                 // valid 16-byte chunks of an operand row

  VL_PROF_STAMP(p, 0);
  // raw q fragments: loaded first, scaled only after the chunk is staged (the loads share one memory round trip)
  u32x4 qraw[KS];
  {
    const bf16_t* Qg = Qb + (long)qrow * p.q.sr;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      qraw[ks] = u32x4{0u, 0u, 0u, 0u};
      if (!PAD || ks * 2 + fg < nch) qraw[ks] = *(const u32x4*)(Qg + ks * 16 + fg * 8);
    }
  }
  [[maybe_unused]] u32x4 tailraw = {0u, 0u, 0u, 0u};
  if constexpr (TAILQ) {
    // (!PAD: EVERY thread loads one of the row's CH chunks and stores it below - duplicates write identical bytes.  A load under
    //  a lane condition makes hipcc close the block with s_waitcnt vmcnt(0): a memory round trip of its own in front of the
    //  K / V requests.)
    if constexpr (!PAD) tailraw = *(const u32x4*)(Qb + (long)(p.Lq - 1) * p.q.sr + (tid & (CH - 1)) * 8);
    else if (tid < nch) tailraw = *(const u32x4*)(Qb + (long)(p.Lq - 1) * p.q.sr + tid * 8);
  }
  unsigned char* const sVr = (unsigned char*)sV;       // DMA: V as a row image [KC][128 B], chunks swizzled by vswz(row)
  auto vswz = [](int row) { const int x = (row >> 1) & 7; return ((x & 1) << 2) | (x >> 1); };      // (= the fused backward's fb_swz)
  if constexpr (DMA) {
    // unit u = rows 8u .. 8u+7 of K and of V: one 1 KB LDS-DMA instruction each (lane -> row 8u + lane/8, 16-byte slot lane%8,
    // which holds the row's chunk slot ^ swizzle(row)); rows beyond Lk are not fetched - K's stay whatever they were (their
    // scores are replaced by -inf), V's are zeroed (0 * garbage could be NaN)
    typedef __attribute__((address_space(3))) void* lds_p;
    const int nrows = ((p.Lk + 31) >> 5) << 5;
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)Kg, 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)Vg, 0, 0x7ffffff0, 0x00020000);
    for (int u = wid; u * 8 < p.Lk; u += nwq) {
      const int row = u * 8 + (lane >> 3), sl = lane & 7;
      if (row < p.Lk) {
        const unsigned offK = (unsigned)(row * (int)p.k.sr + ((sl ^ ((row >> 1) & 7)) << 3)) * 2u;
        const unsigned offV = (unsigned)(row * (int)p.v.sr + ((sl ^ vswz(row)) << 3)) * 2u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lds_p)(sK + u * 1024), 16, offK, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lds_p)(sVr + u * 1024), 16, offV, 0, 0, 0);
      }
    }
    for (int i = tid; i < (nrows - p.Lk) * 8; i += nthr)
      *(u32x4*)(sVr + (p.Lk + (i >> 3)) * 128 + ((i & 7) << 4)) = u32x4{0u, 0u, 0u, 0u};
    // (hipcc does not model the LDS write of the DMA builtin: wait for it by hand in front of the barrier)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
  // first chunk: staged before the accumulators exist (12 x 16-byte loads in flight per thread need the registers)
  stage2<DH, KC, true, false, false, true>(StageSrc{sK, nullptr, Kg, p.k.sr, 1.f, nch}, StageSrc{nullptr, sV, Vg, p.v.sr, 1.f, nch},
                                           0, p.Lk, tid, nthr);
  }
  if constexpr (TAILQ) {
    if constexpr (!PAD) *(u32x4*)(sTailQ + (tid & (CH - 1)) * 8) = tailraw;
    else if (tid < CH) *(u32x4*)(sTailQ + tid * 8) = tailraw;
  }
  VL_PROF_STAMP(p, 1);
  // V^T fragment in accumulator order out of the row image (DMA): this lane's column d = t*32 + (lane & 31), the 8 keys
  // tile*32 + 16c + 4fg + (e & 3) + 8(e >> 2): two transpose reads of 4 rows each (vl_attn_bwd_fused.hip: trQG)
  [[maybe_unused]] const int li = lane & 15, lb3 = (li >> 3) & 1;
  [[maybe_unused]] const unsigned aTr = (unsigned)((fg * 4 + (li >> 2)) * 128) +
                                        (((unsigned)(((lane >> 4) & 1) * 2 + ((li >> 1) & 1)) ^ (unsigned)((lb3 << 2) | fg)) << 4) + (unsigned)((li & 1) << 3);
  auto trV = [&](int tile, int c, int t) {
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
    struct { s16x4 lo, hi; } f;
    const unsigned a = (aTr + (unsigned)(tile * 4096)) ^ (unsigned)(t * 64);
    f.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(sVr + a + c * 2048));
    f.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(sVr + (a ^ 32u) + c * 2048 + 1024));
    return __builtin_bit_cast(bf16x8, f);
  };
  bf16x8 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
    qf[ks] = __builtin_bit_cast(bf16x8, p.qscale != 1.0f ? scale_x8<F16>(qraw[ks], p.qscale) : qraw[ks]);

  float m_run = 0.f;               // the running maximum (valid after the first tile)
  vl_f32x2 l2 = {0.f, 0.f};        // running sum, two partial accumulators
  f32x16 negm;                     // -m_run in every accumulator slot: the C operand of the score MFMA
  f32x16 o[DT];
#pragma unroll
  for (int r = 0; r < 16; ++r) negm[r] = 0.f;
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  bool first = true;

  // ---- the shared last row (query Lq-1, sees every key; host guarantees Lk <= KC and one workgroup per (b,h)) ----
  // Its per-wave share (wave w: the row's scores against key tile w, one more MFMA pair against V, a (max, sum, O) partial
  // in LDS) runs right behind the staging barrier, BEFORE the wave's own tiles; one workgroup barrier there; wave 0 merges
  // the partials after its own tiles.  (Until round 6 all of it ran after the tiles: every wave waited at a barrier for the
  // slowest one and then for wave 0's merge - 8.2 k of a workgroup's 33.2 k cycles, profiles/r05_attn_phase_timeline.log.)
  auto tail_partials = [&]() {
    if constexpr (TAILQ) {
    // ---- the shared last row (query Lq-1, sees every key; host guarantees Lk <= KC and one workgroup per (b,h)) ----
    const int qT = p.Lq - 1;
    const int ntile = (p.Lk + 31) >> 5;
    const int ksw = (fr >> RSH) & (CH - 1);
    bf16x8 qa[KS];                         // A operand: row 0 = the query, rows 1..31 zero
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int e = 0; e < 8; ++e) qa[ks][e] = (__bf16)0.f;
      if (fr == 0) {
        u32x4 raw = *(const u32x4*)(sTailQ + ks * 16 + fg * 8);
        if (p.qscale != 1.0f) raw = scale_bf16x8(raw, p.qscale);
        qa[ks] = __builtin_bit_cast(bf16x8, raw);
      }
    }
    for (int kt = wid; kt < ntile; kt += nwq) {
      f32x16 st;
#pragma unroll
      for (int r = 0; r < 16; ++r) st[r] = 0.f;
      const unsigned char* kbase = sK + (kt * 32 + fr) * RB;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[ks], *(const bf16x8*)(kbase + (((ks * 2 + fg) ^ ksw) * 16)), st, 0, 0, 0);
      // D[i = query row][j = key]: row 0 is accumulator slot 0 of the lanes with fg == 0; lane fr <-> key kt*32 + fr
      const bool valid = fg == 0 && kt * 32 + fr < p.Lk;
      const float sv = valid ? st[0] : -INFINITY;
      const float mw = wave_max_dpp(sv);
      const float pw = valid ? __builtin_amdgcn_exp2f(sv - mw) : 0.f;
      const float lw = wave_sum_dpp(pw);
      float* myP = sP + wid * 32;
      if (fg == 0) myP[fr] = pw;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      f32x16 ot[DT];
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[t][r] = 0.f;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        bf16x8 pa;
#pragma unroll
        for (int e = 0; e < 8; ++e) pa[e] = (__bf16)0.f;
        if (fr == 0) {
          const f32x4 lo = *(const f32x4*)(myP + c * 16 + fg * 4);
          const f32x4 hi = *(const f32x4*)(myP + c * 16 + 8 + fg * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { pa[e] = (__bf16)lo[e]; pa[4 + e] = (__bf16)hi[e]; }
        }
#pragma unroll
        for (int t = 0; t < DT; ++t)
          ot[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
              pa, DMA ? trV(kt, c, t) : *(const bf16x8*)(sV + (t * 32 + fr) * VSP + kt * 32 + c * 16 + fg * 8), ot[t], 0, 0, 0);
      }
      __builtin_amdgcn_wave_barrier();
      float* part = sPart + kt * (2 + DH);
      if (lane == 0) { part[0] = mw; part[1] = lw; }
      if (fg == 0) {
#pragma unroll
        for (int t = 0; t < DT; ++t) part[2 + t * 32 + fr] = ot[t][0];
      }
    }
    }
  };
  auto tail_merge = [&]() {
    if constexpr (TAILQ) {
      const int qT = p.Lq - 1;
      const int ntile = (p.Lk + 31) >> 5;
      if (wid == 0) {
        float M = -INFINITY;
        for (int kt = 0; kt < ntile; ++kt) M = fmaxf(M, sPart[kt * (2 + DH)]);
        for (int d = lane; d < dhr; d += 64) {           // (one pass for head dims up to 64)
          float L = 0.f, acc = 0.f;
          for (int kt = 0; kt < ntile; ++kt) {
            const float* part = sPart + kt * (2 + DH);
            const float w = __builtin_amdgcn_exp2f(part[0] - M);
            L = fmaf(part[1], w, L);
            acc = fmaf(part[2 + d], w, acc);
          }
          p.out[((size_t)b * p.Lq + qT) * (p.H * dhr) + h * dhr + d] = f2bf(acc / L);
          if (p.lse && d == 0) p.lse[bh * p.Lq + qT] = (M + __log2f(L)) * 0.6931471805599453f;
        }
      }
    }
  };
  const int q_hi = q0 + 31;  // last query row of this wave's tile
  const int blk_q_hi = min(p.lq_main - 1, (int)(blockIdx.x * nwq + nwq) * 32 - 1);

  for (int kc0 = 0; kc0 < (MULTI ? p.Lk : 1); kc0 += KC) {
    if (MULTI && p.causal && kc0 > blk_q_hi) break;   // uniform across the workgroup
    if constexpr (MULTI) {
      if (kc0 > 0) {
        __syncthreads();
        // (one item per round here: the accumulators are live and 12 loads in flight would spill them)
        stage2<DH, KC, true, false, false, true, 1>(StageSrc{sK, nullptr, Kg, p.k.sr, 1.f, nch},
                                                    StageSrc{nullptr, sV, Vg, p.v.sr, 1.f, nch}, kc0, p.Lk, tid, nthr);
      }
    }
    __syncthreads();
    VL_PROF_STAMP(p, 2);
    if constexpr (TAILQ) { tail_partials(); __syncthreads(); }      // (TAILQ: one chunk, every wave active)
    if (!wave_active) continue;

    int ntile = (min(p.Lk - kc0, KC) + 31) >> 5;
    if (p.causal) ntile = min(ntile, ((q_hi - kc0) >> 5) + 1);
    const int ksw = (fr >> RSH) & (CH - 1);
    for (int kt = 0; kt < ntile; ++kt) {
      // ---- S^T tile (rows = keys, cols = queries), relative to the running maximum ----
      const unsigned char* kbase = sK + (kt * 32 + fr) * RB;
      bf16x8 kf[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) kf[ks] = *(const bf16x8*)(kbase + (((ks * 2 + fg) ^ ksw) * 16));
      f32x16 s = mfma32<F16>(kf[0], qf[0], negm);
#pragma unroll
      for (int ks = 1; ks < KS; ++ks) s = mfma32<F16>(kf[ks], qf[ks], s);
      // V^T fragments of the first 16-key slice: issued before the softmax arithmetic so that they land under it
      // (the second slice is fetched after the exponentials, into the registers the scores vacate)
      bf16x8 vf0[DT];
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        if constexpr (DMA) vf0[t] = trV(kt, 0, t);
        else vf0[t] = *(const bf16x8*)(sV + (t * 32 + fr) * VSP + kt * 32 + fg * 8);
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
      float mx = fmaxf(fmaxf(s[0], s[1]), s[2]);      // v_max3_f32 chain (file is built with -fno-honor-nans)
#pragma unroll
      for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, s[r]), s[r + 1]);
      mx = fmaxf(mx, s[15]);
      mx = xhalf_max(mx);
      if (__builtin_amdgcn_ballot_w64(first || mx > RESCALE_THR) != 0) {
        // the maximum moved: shift this tile's scores, the C block and the accumulated sums (rare after the first tiles)
        const float d = first ? mx : fmaxf(mx, 0.f);
        const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(-d);
        m_run += d;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] -= d; negm[r] -= d; }
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
        l2 *= alpha;
        first = false;
      }
      float pv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) pv[r] = __builtin_amdgcn_exp2f(s[r]);
      {
        vl_f32x2 a0 = {pv[0], pv[1]}, a1 = {pv[2], pv[3]}, a2 = {pv[4], pv[5]}, a3 = {pv[6], pv[7]};
        const vl_f32x2 a4 = {pv[8], pv[9]}, a5 = {pv[10], pv[11]}, a6 = {pv[12], pv[13]}, a7 = {pv[14], pv[15]};
        a0 += a4; a1 += a5; a2 += a6; a3 += a7;
        a0 += a2; a1 += a3;
        l2 += a0 + a1;
      }
      // ---- O^T += V^T . P^T ----
      bf16x8 vf1[DT];
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        if constexpr (DMA) vf1[t] = trV(kt, 1, t);
        else vf1[t] = *(const bf16x8*)(sV + (t * 32 + fr) * VSP + kt * 32 + 16 + fg * 8);
      }
      {
        const bf16x8 pf = pack8x<F16>(pv);
#pragma unroll
        for (int t = 0; t < DT; ++t) o[t] = mfma32<F16>(vf0[t], pf, o[t]);
      }
      {
        const bf16x8 pf = pack8x<F16>(pv + 8);
#pragma unroll
        for (int t = 0; t < DT; ++t) o[t] = mfma32<F16>(vf1[t], pf, o[t]);
      }
    }
  }

  VL_PROF_STAMP(p, 3);
  if (wave_active) {
    const float l_tot = xhalf_sum(l2[0] + l2[1]);
    const float inv = 1.0f / l_tot;
    store_rows_t<DT, F16>(o, inv, p.out + ((size_t)b * p.Lq + qrow) * (p.H * dhr) + h * dhr, fg, qidx < p.lq_main, nch);
    if (p.lse && fg == 0 && qidx < p.lq_main)
      p.lse[bh * p.Lq + qidx] = (m_run + __log2f(l_tot)) * 0.6931471805599453f;
  }

  VL_PROF_STAMP(p, 4);
  tail_merge();
  VL_PROF_STAMP(p, 5);
}

}  // namespace

extern "C" int vl_set_error(const char* msg);

template <int DH, bool TAILQ, bool MULTI, bool F16 = false, bool DMA = false>
static int launch_fwd(const AttnP& p, int gx, int nwq, hipStream_t stream) {
  const size_t smem = (size_t)KC * DH * 2 + (size_t)DH * VSP * 2 + (size_t)NWMAX * 32 * 4 + (size_t)(KC / 32) * (2 + DH) * 4 +
                      (size_t)DH * 2;
  static const hipError_t attr = hipFuncSetAttribute((const void*)attn_fwd_kernel<DH, TAILQ, MULTI, F16, DMA>,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (attr != hipSuccess) return vl_set_error(hipGetErrorString(attr));
  hipLaunchKernelGGL((attn_fwd_kernel<DH, TAILQ, MULTI, F16, DMA>), dim3(gx, p.H, p.B), dim3(nwq * 64), smem, stream, p);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : vl_set_error(hipGetErrorString(e));
}

extern "C" int vl_attn_fwd_bf16(const void* q, const void* k, const void* v, const long* strides, void* out, float* lse,
                                int B, int H, int Lq, int Lk, int dh, float qscale, int causal, hipStream_t stream) {
  if (B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0) return vl_set_error("vl_attn_fwd_bf16: empty problem");
  if (dh != 32 && dh != 64 && !(dh > 64 && dh <= 128 && (dh & 7) == 0))
    return vl_set_error("vl_attn_fwd_bf16: head dim must be 32, 64, or a multiple of 8 in (64, 128] (run zero-padded to 128)");
  if (!strides) return vl_set_error("vl_attn_fwd_bf16: strides required");
  for (int i = 0; i < 9; ++i)
    if (strides[i] & 7) return vl_set_error("vl_attn_fwd_bf16: operand strides must be multiples of 8 elements (16-byte rows)");
  if ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v)) & 15) return vl_set_error("vl_attn_fwd_bf16: operands must be 16-byte aligned");
  AttnP p{TV{(const bf16_t*)q, strides[0], strides[1], strides[2]}, TV{(const bf16_t*)k, strides[3], strides[4], strides[5]},
          TV{(const bf16_t*)v, strides[6], strides[7], strides[8]}, qscale, (bf16_t*)out, lse, B, H, Lq, Lk, causal, dh,
#ifdef VL_ATTN_PROF
          vl_attn_prof_buf,
#endif
          Lq};
  // one row beyond whole tiles (257 tokens): shared by the waves of the single workgroup instead of a ninth wave
  const bool tailq = (Lq % 32 == 1) && Lq > 32 && Lq - 1 <= NWMAX * 32 && Lk <= KC && (!causal || Lk <= Lq);
  if (tailq) p.lq_main = Lq - 1;
  const int qtiles = (p.lq_main + 31) / 32;
  const int nwq = qtiles < NWMAX ? qtiles : NWMAX;
  const int gx = (qtiles + nwq - 1) / nwq;
  const bool multi = Lk > KC;
#define VL_FWD(DHV)                                                                  \
  (tailq ? launch_fwd<DHV, true, false>(p, gx, nwq, stream)                          \
         : (multi ? launch_fwd<DHV, false, true>(p, gx, nwq, stream) : launch_fwd<DHV, false, false>(p, gx, nwq, stream)))
  // head dim 64, one key chunk: K / V staged by LDS-DMA, V^T fragments by transpose reads (DMA = true)
  if (dh == 64 && !multi)
    return tailq ? launch_fwd<64, true, false, false, true>(p, gx, nwq, stream) : launch_fwd<64, false, false, false, true>(p, gx, nwq, stream);
  return dh == 64 ? VL_FWD(64) : (dh == 32 ? VL_FWD(32) : VL_FWD(128));
#undef VL_FWD
}

// The same kernel on IEEE-half operands (q, k, v, out fp16; lse fp32): head dim 64, at most one LDS chunk of keys (288), any
// mask mode - the frozen text tower (77 tokens, causal; open_clip/model.py:528-540, transformer.py:241-252, 870-876).
extern "C" int vl_attn_fwd_f16(const void* q, const void* k, const void* v, const long* strides, void* out, float* lse,
                               int B, int H, int Lq, int Lk, int dh, float qscale, int causal, hipStream_t stream) {
  if (B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0) return vl_set_error("vl_attn_fwd_f16: empty problem");
  if (dh != 64) return vl_set_error("vl_attn_fwd_f16: head dim must be 64");
  if (Lk > KC) return vl_set_error("vl_attn_fwd_f16: at most 288 keys");
  if (!strides) return vl_set_error("vl_attn_fwd_f16: strides required");
  for (int i = 0; i < 9; ++i)
    if (strides[i] & 7) return vl_set_error("vl_attn_fwd_f16: operand strides must be multiples of 8 elements (16-byte rows)");
  if ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v)) & 15) return vl_set_error("vl_attn_fwd_f16: operands must be 16-byte aligned");
  AttnP p{TV{(const bf16_t*)q, strides[0], strides[1], strides[2]}, TV{(const bf16_t*)k, strides[3], strides[4], strides[5]},
          TV{(const bf16_t*)v, strides[6], strides[7], strides[8]}, qscale, (bf16_t*)out, lse, B, H, Lq, Lk, causal, dh,
#ifdef VL_ATTN_PROF
          vl_attn_prof_buf,
#endif
          Lq};
  const int qtiles = (Lq + 31) / 32;
  const int nwq = qtiles < NWMAX ? qtiles : NWMAX;
  const int gx = (qtiles + nwq - 1) / nwq;
  return launch_fwd<64, false, false, true>(p, gx, nwq, stream);
}
