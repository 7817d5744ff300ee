// EXPERIMENTAL (round 4, built without a GPU in reach - NOT dispatched by cfg -1, NOT used by any product path): the persistent
// bf16 NT GEMM with ONE wave per SIMD and a 128x128 wave tile.
//
// Why: the 8-wave kernel of vl_gemm_park.hip (2 x 4 waves of 128 x 64) reads 12 fragments per 32 MFMAs from LDS; an LDS-fed
// loop of that shape with nothing else in it tops out at 1 529 TF/s on this power-limited part against 2 024 TF/s from
// registers: the LDS reads are a quarter of the energy.  A 128 x 128 wave tile needs 16 fragments per 64 MFMAs
// (8 per 32): a third fewer reads per flop.  It takes four waves per 256 x 256 tile, i.e. one wave per SIMD with 256
// accumulator registers (AGPRs) + up to 256 others, and a software-pipelined stream because no second wave hides latency.
//
// What is here: the main loop (LDS-DMA two k-steps ahead and across tile boundaries, one barrier per k-step, fragments of
// the next half k-step in flight under the 64 MFMAs of the current one, hand-ordered) and the epilogues of the 8-wave kernel
// that carry the C3 step (plain bf16, GELU, GELU + gelu', bf16 residual, dGELU from the saved gelu': bias, wave-private LDS
// transpose, 16-byte non-temporal stores), run once per 64-column half of the wave's sub-tile.  Reachable only through
// vl_gemm_bf16(..., cfg = 14) on whole 256x256 tiles with K >= 512; tools/pk4_probe.py checks every variant against the
// shipped kernel and times both.  First thing to do with it on a GPU: that probe; then the phase counters (k-loop cycles per
// step against the 8-wave kernel's 2 460).  If it wins: launch_best_persist() in vl_gemm.hip is where to prefer it.
#include <hip/hip_runtime.h>
#include <type_traits>

#include "vl_gemm_common.h"

namespace {

constexpr int P4_STAGE = 65536, P4_ABYTES = 32768;
constexpr int P4_LDS = 2 * P4_STAGE + 4 * 4096;           // 144 KB: two operand stages (A 256 x 64, W 256 x 64, bf16) + four 4 KB transpose slabs
constexpr int P4_GN = 8;                                  // N-tiles per group of the tile order (as the 8-wave kernel)

typedef __attribute__((address_space(3))) void* lds_ptr4_t;

// EPI / ACT as in vl_gemm_park.hip, the subset that carries the C3 step: EPI_BF16 with ACT 0 (plain), 1 (GELU), 4 (GELU, gelu' of
// the bf16-rounded pre-activation to out2); EPI_RES_BF16 (bf16 residual, in place allowed); EPI_DGELU with ACT 4 (times the
// saved gelu').
template <int EPI, int ACT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) gemm_nt_pk4_kernel(const GemmP p) {
  constexpr bool HAS_AUX = (EPI == EPI_RES_BF16 || EPI == EPI_DGELU);       // second operand of the output's shape
  constexpr bool OUT2 = (EPI == EPI_BF16 && ACT == 4);
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
    const int GN = p.pk_gn > 0 ? p.pk_gn : P4_GN;
    const int gsz = GN * tiles_m;
    const int gid = v / gsz, rem = v - gid * gsz;
    const int first_n = gid * GN;
    const int gn = min(tiles_n - first_n, GN);
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
  const int fa16 = (wave_m * 128 + fr16) * 128, fw16 = P4_ABYTES + (wave_n * 128 + fr16) * 128;
  bf16x8 af[2][8], wf[2][8];
  f32x4 acc[8][8];
  // one fragment (W: jb, A: ia) of k half h of a stage into set c
  auto ldW = [&](const unsigned char* stage, int h, int c, int jb) {
    wf[c][jb] = *(const bf16x8*)(stage + fw16 + jb * 2048 + ((h * 4 + fq) ^ fsw16) * 16);
  };
  auto ldA = [&](const unsigned char* stage, int h, int c, int ia) {
    af[c][ia] = *(const bf16x8*)(stage + fa16 + ia * 2048 + ((h * 4 + fq) ^ fsw16) * 16);
  };
  auto ldfrags = [&](const unsigned char* stage, int h, int c) {
#pragma unroll
    for (int jb = 0; jb < 8; ++jb) ldW(stage, h, c, jb);
#pragma unroll
    for (int ia = 0; ia < 8; ++ia) ldA(stage, h, c, ia);
  };
  // The 8 MFMAs of accumulator row ia from fragment set c.  Inline assembly with the accumulator TIED to an AGPR operand:
  // through the builtin hipcc 7.2 allocated C and D of many of these MFMAs to different registers and, with all 256 AGPRs
  // holding the tile, rotated them through VGPRs - 328 v_accvgpr moves per k-step (ISA inspection).  Every accumulator block
  // is written once per 64 MFMAs: no MFMA -> MFMA hazard to pad by hand; the fragments are ordinary operands (the compiler
  // places the lgkmcnt waits).  ZC: the first k half of a tile accumulates onto the inline constant 0 - no zeroing pass.
  auto mma_row = [&](int c, int ia) {
#pragma unroll
    for (int jb = 0; jb < 8; ++jb)
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[ia][jb]) : "v"(wf[c][jb]), "v"(af[c][ia]));
  };
  auto mma_row_zc = [&](int c, int ia) {
#pragma unroll
    for (int jb = 0; jb < 8; ++jb)
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc[ia][jb]) : "v"(wf[c][jb]), "v"(af[c][ia]));
  };

  int cur_m0, cur_n0;
  tile_origin(0, cur_m0, cur_n0);
  make_rsrc(cur_m0, cur_n0, rsA, rsW);
  __amdgpu_buffer_rsrc_t rsA_n = rsA, rsW_n = rsW;
  int dti = 0, dkt = 0;                                     // DMA position: runs two k-steps ahead of the MFMAs, across tile boundaries
  // unit i (0..7) of both operands of the k-step at the DMA position; 8 pieces = one k-step, then dma_advance()
  auto dma_piece = [&](unsigned char* stage, int i) {
    const int kbyte = dkt << 7;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr4_t)(stage + (i * 4 + wid) * 1024), 16, voffA, kbyte + i * a_unit, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr4_t)(stage + P4_ABYTES + (i * 4 + wid) * 1024), 16, voffW, kbyte + i * w_unit, 0, 0);
  };
  auto dma_advance = [&]() {
    ++dkt;
    if (dkt == nk) { dkt = 0; ++dti; rsA = rsA_n; rsW = rsW_n; }
  };
  auto dma_step = [&](unsigned char* stage) {
#pragma unroll
    for (int i = 0; i < 8; ++i) dma_piece(stage, i);
    dma_advance();
  };
  // (hipcc does not model the LDS write of the DMA builtin: wait by hand; raw barrier, no fence - vl_gemm_park.hip)
  auto dma_wait_and_barrier = [&]() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  // the FIRST barrier of a tile behind an epilogue: the DMA it needs was issued in front of the epilogue, i.e. it is older than
  // the epilogue's NST output stores in the in-order vmcnt queue - those keep draining under the first k-step (vl_gemm_park.hip)
  constexpr int NST = OUT2 ? 63 : 32;                       // (vmcnt is 6 bits on gfx9: 63 is the most that can be left outstanding)
  auto first_wait_and_barrier = [&]() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NST) : "memory"); };
  bool after_epi = false;                                   // (wave-uniform) the next barrier is the first one behind an epilogue

  dma_step(smem);
  dma_step(smem + P4_STAGE);                                // nk >= 8: still inside tile 0
  dma_wait_and_barrier();
  ldfrags(smem, 0, 0);

  // One k-step = two phases of 64 MFMAs on ONE wave per SIMD: nothing else hides latency, so every phase carries its share of the
  // other work BETWEEN its MFMA rows (sched_barrier keeps the hand order):
  //   phase A (k half 0, set 0): per MFMA row (8 MFMAs, ~136 cycles) two of the 16 fragment reads of half 1 (set 1) in front of
  //           it.  (All 16 reads up front cannot be waited for without a stall: lgkmcnt is 4 bits, hipcc emitted lgkmcnt(14),
  //           i.e. a wait on two of the NEW reads; and eight reads in a row leave the matrix pipe idle behind the last MFMA);
  //   barrier (every wave holds its last fragments of `cur`; the DMA of the next k-step, issued one k-step ago, has landed);
  //   phase B (k half 1, set 1): per MFMA row two fragment reads of the next k-step's half 0 (set 0, from `oth`) in front of it
  //           and, behind each of the first four rows, two DMA pieces of the k-step after next (into `cur`).
  int par = 0;
  auto kstep = [&](auto FIRST, auto LAST) {
    constexpr bool last = decltype(LAST)::value, first = decltype(FIRST)::value;
    unsigned char* cur = smem + par * P4_STAGE;
    unsigned char* oth = smem + (par ^ 1) * P4_STAGE;
    auto& row_acc = mma_row; auto& row_new = mma_row_zc;    // (named outside the discarded branches of this generic lambda)
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      ldW(cur, 1, 1, g); ldA(cur, 1, 1, g);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (first) row_new(0, g); else row_acc(0, g);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (after_epi) { first_wait_and_barrier(); after_epi = false; } else dma_wait_and_barrier();
    const bool more = dti < my_tiles;                       // (wave-uniform) operands left to fetch
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      if constexpr (!last) { ldW(oth, 0, 0, g); ldA(oth, 0, 0, g); }     // (at a tile boundary the fragments would sit in registers through the epilogue)
      __builtin_amdgcn_sched_barrier(0);
      row_acc(1, g);                                        // MFMAs first behind the barrier: the matrix pipe drained while the wave waited
      __builtin_amdgcn_sched_barrier(0);
      // the eight DMA pieces go behind the FIRST four rows, two each (M0 write, wait state, 2 x 2 KB issued while the row
      // executes): the data is read one k-step later, and the last piece should have more than one phase (~1 100 cycles,
      // about one loaded HBM round trip) to land before that barrier's vmcnt(0)
      if (g < 4 && more) { dma_piece(cur, 2 * g); dma_piece(cur, 2 * g + 1); }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (more) dma_advance();
    par ^= 1;
  };

  for (int ti = 0; ti < my_tiles; ++ti) {
    if (ti + 1 < my_tiles) {
      int nm0, nn0;
      tile_origin(ti + 1, nm0, nn0);
      make_rsrc(nm0, nn0, rsA_n, rsW_n);
    }
    after_epi = ti > 0;
    kstep(std::true_type{}, std::false_type{});
    for (int kt = 1; kt < nk - 1; ++kt) kstep(std::false_type{}, std::false_type{});
    kstep(std::false_type{}, std::true_type{});
    {
      // ---------------- tile finished: bias, bf16, wave-private LDS transpose, 16-byte non-temporal stores ----------------
      // (the plain-bf16 epilogue of vl_gemm_park.hip, run once per 64-column half of the wave's 128 x 128 sub-tile)
      const GemmP pe = reload_params();
      mfma_results_settled();
      int el = lane;
      asm volatile("" : "+v"(el));                          // (the epilogue's lane constants must not be hoisted above the k-loop)
      const int er = el & 15, eq = el >> 4;                 // accumulator layout: row er of a 16-row block, columns eq*4 .. +3 of a 16-column block
      const int prow = el >> 3, pchunk = el & 7;            // store layout: row prow of an 8-row pass, 16-byte chunk pchunk of the 128-byte row
      const int mrow0 = cur_m0 + wave_m * 128, ncol0 = cur_n0 + wave_n * 128;
      const bool has_bias = pe.bias != nullptr;
      const float* const bsrc = has_bias ? pe.bias : (const float*)pe.W;     // branch-free optional bias: read something valid, select zero
      unsigned char* const slab = smem + 2 * P4_STAGE + wid * 4096;          // 32 rows x 64 columns bf16, 16-byte chunks XOR-swizzled by row & 7
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
        [[maybe_unused]] unsigned char* const o2base = (unsigned char*)pe.out2 + (size_t)mrow0 * ldo2 + lane_off;
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
              f32x4 v = scale_bias(acc[i * 2 + jh][ch * 4 + q], pe.alpha, bvq[q]);
              if constexpr (EPI == EPI_BF16 && ACT == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
              }
              u32x2 o; o[0] = pack2bf(v[0], v[1]); o[1] = pack2bf(v[2], v[3]);
              const int row = jh * 16 + er;
              *(u32x2*)(slab + row * 128 + (((q * 2 + (eq >> 1)) ^ (row & 7)) << 4) + (eq & 1) * 8) = o;
            }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int pass = 0; pass < 4; ++pass) {
            const int r = pass * 8 + prow;
            u32x4 w = *(const u32x4*)(slab + r * 128 + ((pchunk ^ (r & 7)) << 4));
            const size_t roff = (size_t)(i * 32 + pass * 8) * ldo2;
            if constexpr (OUT2) {                           // out = gelu(pre), out2 = gelu'(pre), both of the bf16-rounded pre-activation
              u32x4 d;
#pragma unroll
              for (int e = 0; e < 4; ++e) { unsigned int y, g; gelu_and_grad_pk(w[e], y, g); w[e] = y; d[e] = g; }
              __builtin_nontemporal_store(d, (u32x4*)(o2base + roff));
            } else if constexpr (EPI == EPI_RES_BF16) {
              const u32x4 rr = aux[i & 1][pass];
#pragma unroll
              for (int e = 0; e < 4; ++e)
                w[e] = pack2bf(bf2f((bf16_t)(w[e] & 0xffff)) + bf2f((bf16_t)(rr[e] & 0xffff)), bf2f((bf16_t)(w[e] >> 16)) + bf2f((bf16_t)(rr[e] >> 16)));
            } else if constexpr (EPI == EPI_DGELU) {
              const u32x4 rr = aux[i & 1][pass];
#pragma unroll
              for (int e = 0; e < 4; ++e) w[e] = mul_pk_bf16(w[e], rr[e]);      // aux = gelu' saved by the forward
            }
            __builtin_nontemporal_store(w, (u32x4*)(obase + roff));
          }
        }
      }
      if (ti + 1 < my_tiles) { tile_origin(ti + 1, cur_m0, cur_n0); ldfrags(smem + par * P4_STAGE, 0, 0); }
    }
  }
}

}  // namespace

template <int EPI, int ACT>
static int launch_pk4(const GemmP& p, int ncu, hipStream_t s) {
  auto kern = gemm_nt_pk4_kernel<EPI, ACT>;
  static const hipError_t attr = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, P4_LDS);
  if (attr != hipSuccess) return (int)attr;
  const int tiles = (p.M >> 8) * (p.N >> 8);
  int G = ncu & ~7;
  if (tiles < G) G = (tiles + 7) & ~7;
  hipLaunchKernelGGL(kern, dim3(G), dim3(256), P4_LDS, s, p);
  return (int)hipGetLastError();
}

// Internal entries used by vl_gemm.hip's dispatcher for cfg = 14 (not part of the public C ABI).
bool vl_gemm_pk4_supported(int epi, const void* params) {
  const GemmP& p = *(const GemmP*)params;
  if (p.ksplit_len || p.res_div != 1 || p.ln_mean || p.row_part) return false;
  if ((p.M & 255) || (p.N & 255) || (p.K & 63) || p.K < 512 || p.M <= 0 || p.N <= 0 || (p.ldo & 7)) return false;
  if ((((uintptr_t)p.A | (uintptr_t)p.W | (uintptr_t)p.out | (uintptr_t)p.res | (uintptr_t)p.out2) & 15) || (p.bias && (((uintptr_t)p.bias) & 15))) return false;
  if (epi == EPI_BF16) return !p.res && ((p.act == 0 && !p.out2) || (p.act == 1 && !p.out2) || (p.act == 4 && p.out2));
  if (epi == EPI_RES_BF16) return p.res && p.act == 0 && !p.out2;
  if (epi == EPI_DGELU) return p.res && p.act == 4 && !p.out2 && !p.bias;
  return false;
}

int vl_gemm_pk4_launch(int epi, const void* params, int ncu, hipStream_t s) {
  const GemmP& p = *(const GemmP*)params;
  if (epi == EPI_BF16) return p.act == 1 ? launch_pk4<EPI_BF16, 1>(p, ncu, s) : (p.act == 4 ? launch_pk4<EPI_BF16, 4>(p, ncu, s) : launch_pk4<EPI_BF16, 0>(p, ncu, s));
  if (epi == EPI_RES_BF16) return launch_pk4<EPI_RES_BF16, 0>(p, ncu, s);
  if (epi == EPI_DGELU) return launch_pk4<EPI_DGELU, 4>(p, ncu, s);
  return (int)hipErrorInvalidValue;
}
