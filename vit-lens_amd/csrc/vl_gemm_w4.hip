// Persistent bf16 NT GEMM with ONE wave per SIMD and 128x128 wave tiles, for LONG reductions (K >= 2048: c_proj forward,
// the dX GEMM of c_fc - the dominant launches of the C3 step): C[M,N] = A[M,K] * W[N,K]^T + bf16 epilogue.
// 256x256 output tiles, BK = 64, 4 waves (2 x 2 of 128 x 128), two 64 KB LDS stages + four 4 KB transpose slabs = 144 KB.
//
// Why a second persistent kernel (round 6).  The vendor yardstick (tools/vendor_gemm_yardstick.py, profiles/
// r06_vendor_gemm_yardstick.log) put hipBLASLt's 256x256x64 kernel at 1 530 TF/s on (65 536, 1 024, 4 096) where the 8-wave
// kernel of vl_gemm_park.hip reaches 1 254 (sustained 6 s: 1 461 against 1 284), and at 1 650 against 1 464 on 8192^3 - while
// on the K = 1 024 shapes the 8-wave kernel is 2-3 % ahead.  Counters of both on the same launches (profiles/
// r06_vendor_vs_ours_pmc.txt): identical L2 hits / misses / fabric reads; matrix pipe busy 0.84 against 0.67 of the kernel's
// cycles at K = 4 096 (0.69 against 0.71 at K = 1 024), 8.5 M LDS instructions against 13.0 M.  The vendor kernel is one wave
// per SIMD on 128 x 128 wave tiles: a third fewer fragment reads per MFMA, and - what matters for rows streamed from HBM -
// the fragments of half a k-step sit in registers EARLY, so a stage buffer is free a third of a k-step after its k-step
// began and the LDS-DMA of the k-step after next has 1.1-1.4 k-steps to land; the 8-wave kernel (256 registers per wave: no
// room for that) frees a stage at 75 % of its k-step and gives the DMA 0.75-1.0 (2 830-3 030 cycles per k-step at K = 4 096
// against 2 470 with the same rows resident in the Infinity Cache, profiles/r05_kstep_probe.log).
// Round 5's one-wave-per-SIMD attempt (vl_gemm_pk4.hip, deleted: 8-10 % SLOWER than the 8-wave kernel) kept the 8-wave
// kernel's timing - one barrier in the middle of the k-step, DMA behind it in bursts of four instructions, fragment reads in
// pairs in front of rows of 8 MFMAs - and so had neither advantage.  This kernel takes its accumulator / epilogue code and
// changes the schedule of a k-step (128 MFMAs per wave, two halves of 64 on alternating fragment sets):
//     MFMA   0- 31   16 fragment reads of k half 1 of THIS stage, one behind every second MFMA
//     MFMA  32- 47   (the reads land)
//     ---- lgkmcnt(0) + barrier 1: every wave has read the last byte of this stage ----
//     MFMA  48- 79   the 16 LDS-DMA instructions of the k-step after next into this stage, one behind every second MFMA
//     MFMA  80- 95
//     ---- vmcnt(16) + barrier 2: everything older than those 16 has landed, i.e. the NEXT stage is complete ----
//     MFMA  96-127   16 fragment reads of k half 0 of the next stage, one behind every second MFMA
// The epilogues (bias, bf16, wave-private LDS transpose, 16-byte non-temporal stores, optional bf16 residual and the partial
// row statistics of the LayerNorm folding) run once per 64-column half of the wave's sub-tile, exposed: with one wave per
// SIMD nothing overlaps them, which is why this kernel is dispatched only where the k-loop is long (K >= 2048).
//
// Replaces: nn.Linear mlp.c_proj (+ residual) forward and the c_fc input-gradient GEMM of ResidualAttentionBlock
// (open_clip/transformer.py:226-234, 271) at ViT-L sizes.
#include <hip/hip_runtime.h>
#include <type_traits>

#include "vl_gemm_common.h"

namespace {

constexpr int W4_STAGE = 65536, W4_ABYTES = 32768;
constexpr int W4_LDS = 2 * W4_STAGE + 4 * 4096;           // 144 KB
constexpr int W4_GN = 8;                                  // N-tiles per group of the tile order (as the 8-wave kernel)

typedef __attribute__((address_space(3))) void* lds_ptr_w4;

template <int I>
using IC4 = std::integral_constant<int, I>;

// EPI_BF16 (ACT 0: plain), EPI_RES_BF16 (ACT 0: bf16 residual, in place allowed; ACT 20: + partial row statistics of what it
// stores, for the folded LayerNorm that follows - GemmP::row_part).
template <int EPI, int ACT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) gemm_nt_w4_kernel(const GemmP p) {
  constexpr bool HAS_AUX = (EPI == EPI_RES_BF16);
  constexpr bool STATS = (EPI == EPI_RES_BF16 && ACT == 20);
  static_assert((EPI == EPI_BF16 && ACT == 0) || (EPI == EPI_RES_BF16 && (ACT == 0 || ACT == 20)), "epilogues of the K >= 2048 launches only");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tiles_n = p.N >> 8, tiles_m = p.M >> 8;
  const int nk = p.K >> 6;
  const int ntiles = tiles_m * tiles_n;
  const int G = gridDim.x;
  const int slot = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);        // each XCD a contiguous run of the tile order
  if (slot >= ntiles) return;
  const int my_tiles = (ntiles - slot + G - 1) / G;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wid & 1, wave_n = wid >> 1;            // 2 x 2 waves of 128 x 128

  auto tile_origin = [&](int ti, int& m0, int& n0) {
    const int v = ti * G + slot;
    const int gsz = W4_GN * tiles_m;
    const int gid = v / gsz, rem = v - gid * gsz;
    const int first_n = gid * W4_GN;
    const int gn = min(tiles_n - first_n, W4_GN);
    const int tm = rem / gn;
    m0 = tm << 8; n0 = (first_n + (rem - tm * gn)) << 8;
  };

  // ---- LDS-DMA: unit i (0..7) of an operand = rows i*32 + wid*8 + (lane>>3), 16-byte chunk (lane&7) ^ swizzle(row) ----
  const int drow = wid * 8 + (lane >> 3);
  const int dsw = ((lane & 7) ^ ((drow >> 1) & 7)) * 16;   // (i*32 >> 1) is a multiple of 8: the swizzle does not depend on i
  const unsigned voffA = (unsigned)(drow * p.lda * 2 + dsw), voffW = (unsigned)(drow * p.ldw * 2 + dsw);
  const int a_unit = p.lda * 64, w_unit = p.ldw * 64;      // bytes between units (32 rows)
  __amdgpu_buffer_rsrc_t rsA, rsW;
  auto make_rsrc = [&](int m0, int n0, __amdgpu_buffer_rsrc_t& ra, __amdgpu_buffer_rsrc_t& rw) {
    ra = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (size_t)m0 * p.lda), 0, 0x7ffffff0, 0x00020000);
    rw = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (size_t)n0 * p.ldw), 0, 0x7ffffff0, 0x00020000);
  };

  // ---- fragments: 16 rows x 32 k (lane: row lane&15, 8-wide k chunk lane>>4); accumulator block [ia][jb]: the lane owns output
  // row ia*16 + (lane&15) and the four columns jb*16 + (lane>>4)*4 .. +3 (operands swapped, as the 8-wave kernel) ----
  const int fr16 = lane & 15, fq = lane >> 4;
  const int fsw16 = (fr16 >> 1) & 7;
  const int fa16 = (wave_m * 128 + fr16) * 128, fw16 = W4_ABYTES + (wave_n * 128 + fr16) * 128;
  bf16x8 af[2][8], wf[2][8];
  f32x4 acc[8][8];
  // fragment r (0..7: W block r, 8..15: A block r-8) of k half h of a stage into set c
  auto ldfrag = [&](const unsigned char* stage, int h, int c, int r) {
    const int off = ((h * 4 + fq) ^ fsw16) * 16;
    if (r < 8) wf[c][r] = *(const bf16x8*)(stage + fw16 + r * 2048 + off);
    else af[c][r - 8] = *(const bf16x8*)(stage + fa16 + (r - 8) * 2048 + off);
  };
  auto ldfrags = [&](const unsigned char* stage, int h, int c) {
#pragma unroll
    for (int r = 0; r < 16; ++r) ldfrag(stage, h, c, r);
  };
  // MFMA m (0..63) of a half on fragment set c: accumulator row ia = m / 8, column block jb = m % 8.  Inline assembly with the
  // accumulator TIED to an AGPR operand (through the builtin hipcc 7.2 rotated the 256 accumulators through VGPRs: hundreds of
  // v_accvgpr moves per k-step); every accumulator block is written once per 64 MFMAs, so there is no MFMA -> MFMA hazard to pad.
  // ZC: the first k half of a tile accumulates onto the inline constant 0 - no zeroing pass.
  auto mma1 = [&](int c, int m) {
    const int ia = m >> 3, jb = m & 7;
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[ia][jb]) : "v"(wf[c][jb]), "v"(af[c][ia]));
  };
  auto mma1_zc = [&](int c, int m) {
    const int ia = m >> 3, jb = m & 7;
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc[ia][jb]) : "v"(wf[c][jb]), "v"(af[c][ia]));
  };

  int cur_m0, cur_n0;
  tile_origin(0, cur_m0, cur_n0);
  make_rsrc(cur_m0, cur_n0, rsA, rsW);
  __amdgpu_buffer_rsrc_t rsA_n = rsA, rsW_n = rsW;
  int dti = 0, dkt = 0;                                     // DMA position: two k-steps ahead of the MFMAs, across tile boundaries
  // piece d (0..15) of the k-step at the DMA position: d < 8 -> unit d of A, else unit d - 8 of W
  auto dma_piece = [&](unsigned char* stage, int d) {
    const int kbyte = dkt << 7;
    if (d < 8) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_w4)(stage + (d * 4 + wid) * 1024), 16, voffA, kbyte + d * a_unit, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr_w4)(stage + W4_ABYTES + ((d - 8) * 4 + wid) * 1024), 16, voffW, kbyte + (d - 8) * w_unit, 0, 0);
  };
  auto dma_advance = [&]() {
    ++dkt;
    if (dkt == nk) { dkt = 0; ++dti; rsA = rsA_n; rsW = rsW_n; }
  };
  auto dma_step = [&](unsigned char* stage) {
#pragma unroll
    for (int d = 0; d < 16; ++d) dma_piece(stage, d);
    dma_advance();
  };
  // (hipcc does not model the LDS write of the DMA builtin: the vmcnt waits are written by hand; raw barriers, no fences - the
  //  only cross-wave LDS traffic is the DMA (vmcnt) and fragment READS (lgkmcnt), as in vl_gemm_park.hip)
  auto reads_done_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  auto dma_all_barrier = [&]() { asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); };      // (prologue)
  auto dma_older_barrier = [&]() { asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory"); };
  // barrier 2 of a tile's FIRST k-step behind an epilogue: in front of this k-step's 16 DMA instructions the in-order queue
  // holds the epilogue's NSTW output stores, and only what is OLDER than those (the DMA of this tile's second k-step, issued in
  // the previous tile's last k-step) has to have landed: the stores keep draining under the k-loop (vl_gemm_park.hip, NST)
  constexpr int NSTW = 2 * 16 * (STATS ? 2 : 1);            // stores per wave and tile: 2 halves x 16 chunks (x 2 with the partial sums)
  constexpr int FIRSTW = NSTW + 16 > 63 ? 63 : NSTW + 16;   // (vmcnt is a 6-bit counter)
  auto first_older_barrier = [&]() { asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(FIRSTW) : "memory"); };
  bool after_epi = false;                                   // (wave-uniform)

  dma_step(smem);
  dma_step(smem + W4_STAGE);                                // nk >= 32: still inside tile 0
  dma_all_barrier();
  ldfrags(smem, 0, 0);

  int par = 0;
  // (no branch inside: in the last two k-steps of a workgroup's last tile there is nothing left to fetch - the descriptors
  //  then have zero records and the 16 DMA instructions are out-of-range no-ops, see the tile loop - so the 128 MFMAs and what
  //  is interleaved with them stay ONE basic block and `vmcnt(16)` means the same thing in every k-step)
  auto kstep = [&](auto FIRST, auto LAST) {
    constexpr bool last = decltype(LAST)::value, first = decltype(FIRST)::value;
    unsigned char* cur = smem + par * W4_STAGE;
    unsigned char* oth = smem + (par ^ 1) * W4_STAGE;
    auto& m_acc = mma1; auto& m_new = mma1_zc;              // (named outside the discarded branches of this generic lambda)
    // ---- k half 0 on set 0 ----
#pragma unroll
    for (int g = 0; g < 16; ++g) {                          // MFMA 0-31: one fragment read of half 1 (set 1) behind every second MFMA
      if constexpr (first) { m_new(0, 2 * g); m_new(0, 2 * g + 1); } else { m_acc(0, 2 * g); m_acc(0, 2 * g + 1); }
      __builtin_amdgcn_sched_barrier(0);
      ldfrag(cur, 1, 1, g);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int m = 32; m < 48; ++m) { if constexpr (first) m_new(0, m); else m_acc(0, m); }
    __builtin_amdgcn_sched_barrier(0);
    reads_done_barrier();                                   // every wave holds all its fragments of `cur`
#pragma unroll
    for (int g = 0; g < 8; ++g) {                           // MFMA 48-63: DMA pieces 0-7 (A) of the k-step after next into `cur`
      if constexpr (first) { m_new(0, 48 + 2 * g); m_new(0, 49 + 2 * g); } else { m_acc(0, 48 + 2 * g); m_acc(0, 49 + 2 * g); }
      __builtin_amdgcn_sched_barrier(0);
      dma_piece(cur, g);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- k half 1 on set 1 ----
#pragma unroll
    for (int g = 0; g < 8; ++g) {                           // MFMA 64-79: DMA pieces 8-15 (W)
      m_acc(1, 2 * g); m_acc(1, 2 * g + 1);
      __builtin_amdgcn_sched_barrier(0);
      dma_piece(cur, 8 + g);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int m = 16; m < 32; ++m) m_acc(1, m);              // MFMA 80-95
    __builtin_amdgcn_sched_barrier(0);
    // the NEXT stage (issued one k-step ago) has landed for every wave; this k-step's own 16 pieces stay in flight
    if (after_epi) { first_older_barrier(); after_epi = false; } else dma_older_barrier();
#pragma unroll
    for (int g = 0; g < 16; ++g) {                          // MFMA 96-127: fragment reads of the next k-step's half 0 (set 0) from `oth`
      m_acc(1, 32 + 2 * g); m_acc(1, 33 + 2 * g);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!last) ldfrag(oth, 0, 0, g);            // (at a tile boundary the fragments would sit in registers through the epilogue)
      __builtin_amdgcn_sched_barrier(0);
    }
    dma_advance();
    par ^= 1;
  };

  for (int ti = 0; ti < my_tiles; ++ti) {
    if (ti + 1 < my_tiles) {
      int nm0, nn0;
      tile_origin(ti + 1, nm0, nn0);
      make_rsrc(nm0, nn0, rsA_n, rsW_n);
    } else {
      // behind the last tile: zero records - every address is out of range, the DMA instructions fetch nothing
      rsA_n = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0, 0x00020000);
      rsW_n = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0, 0x00020000);
    }
    after_epi = ti > 0;
    kstep(std::true_type{}, std::false_type{});
    for (int kt = 1; kt < nk - 1; ++kt) kstep(std::false_type{}, std::false_type{});
    kstep(std::false_type{}, std::true_type{});
    {
      // ---------------- tile finished: bias, bf16, wave-private LDS transpose, 16-byte non-temporal stores ----------------
      // (the epilogues of vl_gemm_park.hip, run once per 64-column half of the wave's 128 x 128 sub-tile)
      const GemmP pe = reload_params();
      mfma_results_settled();
      int el = lane;
      asm volatile("" : "+v"(el));                          // (the epilogue's lane constants must not be hoisted above the k-loop)
      const int er = el & 15, eq = el >> 4;                 // accumulator layout: row er of a 16-row block, columns eq*4 .. +3 of a 16-column block
      const int prow = el >> 3, pchunk = el & 7;            // store layout: row prow of an 8-row pass, 16-byte chunk pchunk of the 128-byte row
      const int mrow0 = cur_m0 + wave_m * 128, ncol0 = cur_n0 + wave_n * 128;
      const bool has_bias = pe.bias != nullptr;
      const float* const bsrc = has_bias ? pe.bias : (const float*)pe.W;     // branch-free optional bias: read something valid, select zero
      unsigned char* const slab = smem + 2 * W4_STAGE + wid * 4096;          // 32 rows x 64 columns bf16, 16-byte chunks XOR-swizzled by row & 7
      const size_t ldo2 = (size_t)pe.ldo * 2;
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) {
        f32x4 bvq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          bvq[q] = *(const f32x4*)(bsrc + ncol0 + ch * 64 + q * 16 + eq * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) bvq[q][e] = has_bias ? bvq[q][e] : 0.f;
        }
        const size_t lane_off = ((size_t)prow * pe.ldo + ncol0 + ch * 64 + pchunk * 8) * 2;       // this lane's 16 bytes of a pass
        unsigned char* const obase = (unsigned char*)pe.out + (size_t)mrow0 * ldo2 + lane_off;
        // EPI_RES_BF16 indexes its residual by the absolute row (m + m_off) with an un-offset pointer (vl_gemm.hip run_gemm)
        [[maybe_unused]] const unsigned char* const abase =
            (const unsigned char*)pe.res + (size_t)(mrow0 + (EPI == EPI_RES_BF16 ? pe.m_off : 0)) * ldo2 + lane_off;
        [[maybe_unused]] u32x4 aux[2][4];                   // second operand of row block i in aux[i & 1], requested one block ahead
        auto load_aux = [&](int i) {
          if constexpr (HAS_AUX) {
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) aux[i & 1][pass] = *(const u32x4*)(abase + (size_t)(i * 32 + pass * 8) * ldo2);
          }
        };
        load_aux(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {                       // 32-row blocks
          if (i < 3) load_aux(i + 1);
#pragma unroll
          for (int jh = 0; jh < 2; ++jh)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const f32x4 v = scale_bias(acc[i * 2 + jh][ch * 4 + q], pe.alpha, bvq[q]);
              u32x2 o; o[0] = pack2bf(v[0], v[1]); o[1] = pack2bf(v[2], v[3]);
              const int row = jh * 16 + er;
              *(u32x2*)(slab + row * 128 + (((q * 2 + (eq >> 1)) ^ (row & 7)) << 4) + (eq & 1) * 8) = o;
            }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          [[maybe_unused]] float st1[4], st2[4];            // STATS: this lane's partial row sums of the four passes
#pragma unroll
          for (int pass = 0; pass < 4; ++pass) {
            const int r = pass * 8 + prow;
            u32x4 w = *(const u32x4*)(slab + r * 128 + ((pchunk ^ (r & 7)) << 4));
            const size_t roff = (size_t)(i * 32 + pass * 8) * ldo2;
            if constexpr (EPI == EPI_RES_BF16) {
              const u32x4 rr = aux[i & 1][pass];
#pragma unroll
              for (int e = 0; e < 4; ++e)
                w[e] = pack2bf(bf2f((bf16_t)(w[e] & 0xffff)) + bf2f((bf16_t)(rr[e] & 0xffff)), bf2f((bf16_t)(w[e] >> 16)) + bf2f((bf16_t)(rr[e] >> 16)));
              if constexpr (STATS) {
                // (sum, sum of squares) of the 8 STORED bf16 values of this lane (vl_gemm_park.hip: inline assembly, hipcc 7.2
                //  fed all four builtin dot products the first dword)
                const unsigned one2 = 0x3f803f80u;
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const unsigned we = w[e];
                  asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(s1) : "v"(we), "v"(one2));
                  asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(s2) : "v"(we), "v"(we));
                }
                st1[pass] = s1; st2[pass] = s2;
              }
            }
            __builtin_nontemporal_store(w, (u32x4*)(obase + roff));
          }
          if constexpr (STATS) {
            // the 8 lanes that hold a row's 64 columns: three DPP steps, then one 8-byte store per row and 64-column slice
            asm volatile("s_nop 1" ::: "memory");           // the sums come out of inline assembly: the DPP read's wait states by hand
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) { st1[pass] += dpp_f<0xB1>(st1[pass]); st2[pass] += dpp_f<0xB1>(st2[pass]); }
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) { st1[pass] += dpp_f<0x4E>(st1[pass]); st2[pass] += dpp_f<0x4E>(st2[pass]); }
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) { st1[pass] += dpp_f<0x141>(st1[pass]); st2[pass] += dpp_f<0x141>(st2[pass]); }
            if (pchunk == 0) {
              float* dst = pe.row_part + ((size_t)(mrow0 + i * 32 + prow) * (pe.N >> 6) + ((ncol0 + ch * 64) >> 6)) * 2;
#pragma unroll
              for (int pass = 0; pass < 4; ++pass)
                *(vl_f32x2*)(dst + (size_t)pass * 8 * (pe.N >> 6) * 2) = vl_f32x2{st1[pass], st2[pass]};
            }
          }
        }
      }
      if (ti + 1 < my_tiles) { tile_origin(ti + 1, cur_m0, cur_n0); ldfrags(smem + par * W4_STAGE, 0, 0); }
    }
  }
}

template <int EPI, int ACT>
int launch_w4(const GemmP& p, int ncu, hipStream_t s) {
  auto kern = gemm_nt_w4_kernel<EPI, ACT>;
  static const hipError_t attr = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS);   // thread-safe one-time init
  if (attr != hipSuccess) return (int)attr;
  const int tiles = (p.M >> 8) * (p.N >> 8);
  int G = ncu & ~7;
  if (tiles < G) G = (tiles + 7) & ~7;
  hipLaunchKernelGGL(kern, dim3(G), dim3(256), W4_LDS, s, p);
  return (int)hipGetLastError();
}

}  // namespace

// Internal entries used by vl_gemm.hip's dispatcher (not part of the public C ABI): cfg = 14 explicitly, cfg = -1 for K >= 2048.
bool vl_gemm_w4_supported(int epi, const void* params) {
  const GemmP& p = *(const GemmP*)params;
  if (p.ksplit_len || p.res_div != 1 || p.ln_mean || p.f16 || p.out2 || p.act != 0) return false;
  if ((p.M & 255) || (p.N & 255) || (p.K & 63) || p.K < 2048 || p.M <= 0 || p.N <= 0 || (p.ldo & 7)) return false;
  if ((((uintptr_t)p.A | (uintptr_t)p.W | (uintptr_t)p.out | (uintptr_t)p.res) & 15) || (p.bias && (((uintptr_t)p.bias) & 15))) return false;
  if (epi == EPI_BF16) return !p.res && !p.row_part;
  if (epi == EPI_RES_BF16) return p.res != nullptr;
  return false;
}

int vl_gemm_w4_launch(int epi, const void* params, int ncu, hipStream_t s) {
  const GemmP& p = *(const GemmP*)params;
  if (epi == EPI_BF16) return launch_w4<EPI_BF16, 0>(p, ncu, s);
  if (epi == EPI_RES_BF16) return p.row_part ? launch_w4<EPI_RES_BF16, 20>(p, ncu, s) : launch_w4<EPI_RES_BF16, 0>(p, ncu, s);
  return (int)hipErrorInvalidValue;
}
