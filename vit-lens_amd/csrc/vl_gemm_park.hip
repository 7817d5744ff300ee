// Persistent bf16 NT GEMM for gfx950 (round-2 kernel): C[M,N] = A[M,K] * W[N,K]^T (+ fused bf16-output epilogue),
// 256x256 output tiles, BK = 64, 8 waves (2x4, 128x64 each, two per SIMD), M % 256 == N % 256 == 0, K >= 512.
//
// What changed against the round-1 persistent kernel of vl_gemm.hip, and why (measurements: DESIGN.md section 7,
// profiles/r02*_gemm_var_probe.log; every number below is the c_fc shape M = 65536, N = 4096, K = 1024):
//
//  * ONE barrier per k-step, placed in FRONT of the last 16-deep substep's MFMAs instead of at the end of the step:
//    by then every wave holds its last fragments of the current LDS stage and the next stage has landed, so the DMA of
//    step s+2 and the first fragment reads of step s+1 are issued UNDER those MFMAs, not behind the barrier.
//  * LDS-DMA through a buffer descriptor (`buffer_load_dwordx4 ... offen lds`): ONE voffset VGPR per operand, all
//    other addressing scalar (round 1: 16 address VGPRs).  hipcc does not model this builtin's LDS write in its waitcnt
//    insertion, so the kernel waits `vmcnt(0)` by hand in front of a raw `s_barrier`.
//  * The two waves of a SIMD (wave w and w+4) issue their 8 DMA instructions half a k-step apart (group A right after
//    the barrier, group B at the top of the next step): an LDS-DMA issue blocks its wave for tens of cycles and with
//    both partners blocked at once the matrix pipe idles.  Main loop alone: 1231 TF/s (round 1) -> 1300-1350 TF/s.
//  * Epilogue = arithmetic in the accumulator layout -> bf16 -> wave-private 4 KB LDS slab (XOR-swizzled, no barrier)
//    -> 16-byte NON-TEMPORAL stores, 8 lanes per 128-byte line, issued as ONE burst per wave; the next tile's main
//    loop starts while they drain.  Plain (L2-allocating) stores cost +0.10 ms here: the 0.5 GB output evicts the
//    operand slabs from the 4 MB L2s.  Trickling the stores through the next tile's k-loop from parked registers
//    (the obvious "hide the epilogue" design, built and measured: 0.53 ms) is SLOWER than the burst (0.48 ms): a store
//    instruction blocks its wave and the CU's vector-memory pipe wherever it is placed, so spreading them only spreads
//    the damage over more k-steps.
//  * Epilogues with a second operand of the output's shape (bf16 residual, saved pre-activation of dGELU) load it with
//    16 loads per wave at the START of the epilogue, under the arithmetic of the first row block.  Prefetching it inside
//    the k-loop (2 loads per k-step over the last 8 steps) was built and measured: 0.74 ms against 0.67 ms (c_fc shape,
//    bf16 residual) - same lesson as for the stores: keep the k-loop's vector-memory queue for the DMA.
//
//  * Round 6, measured and NOT kept (profiles/r06_kstagger_probe.log, r06_step_ab_kstagger_auxprefetch.log): (a) walking each
//    tile's reduction from a rotated start (tile row * 2 k-steps, the vendor kernel's "StaggerU") so that the 256 workgroups
//    do not ask for the same 128-byte column slab of rows 8 KB apart at the same instant: +3.6 % on c_proj + residual, +2.9 %
//    on the dX of c_fc, +3.2 % at 8192^3 in isolation, nothing on the K = 1024 shapes; (b) requesting row block 0's second
//    operand in the last k-step, in front of the next tile's first DMA batch.  Interleaved in the C3 step on one box, as
//    separately built libraries (tools/build_variant.sh, tools/lib_ab.sh): 592.1 / 592.5 ms without either, 592.8 / 592.5 with
//    both, 593.3 / 593.4 rotation only, 592.1 / 593.2 prefetch only - the board returns main-loop cycles as clock (DESIGN.md
//    7.1).  The rotation also ends the bit-equality with the round-1 kernel that the parity tests use; both were removed.
//  * Round 6, the vendor yardstick and a second one-wave-per-SIMD kernel (vl_gemm_w4.hip, in the history at commit "vl_gemm_w4.hip:
//    one wave per SIMD ..."): hipBLASLt's 256x256x64 kernel (4 waves of 128x128, 512 registers, fragments of half a k-step
//    resident) runs (65 536, 1 024, 4 096) at 1 491-1 530 TF/s and 8192^3 at 1 647 where this kernel reaches 1 228-1 254 and
//    1 464 - and on the K = 1 024 shapes this kernel is 2-3 % ahead, 16-18 % at M = 65 792 (profiles/r06_vendor_gemm_yardstick.log,
//    counters: r06_vendor_vs_ours_pmc.txt - same L2 traffic, matrix pipe busy 0.84 against 0.67 at K = 4 096).  The rebuilt
//    one-wave-per-SIMD kernel (one fragment read or one DMA instruction behind every second MFMA, two barriers per k-step,
//    bit-identical results) reached 1 113-1 147 on the K = 4 096 launches and cost the C3 step 8 ms (591.3 against 583.5 ms,
//    profiles/r06_step_ab_w4_longk_kernel.log).  Ablation builds of it say where its k-loop goes (r06_w4_ablation*.log, dX of
//    c_fc / 8192^3): as built 1 139 / 1 551; without the 16 DMA instructions per wave and k-step 1 603 / 1 893; without
//    barriers 1 256 / 1 593; without fragment reads 1 189 / 1 664; bare MFMAs 1 730 / 2 070; the XOR-permuted chunk order on
//    the global side of the DMA makes no difference (1 091 against 1 098).  With one wave per SIMD every LDS-DMA issue idles
//    the matrix pipe, here the partner wave covers it; how the vendor's hand-scheduled stream issues the same 16 pieces per
//    wave without that cost is not visible in its disassembly.  Removed; the 8-wave kernel stays the product for every shape.
//
// Replaces: nn.Linear / MultiheadAttention in/out projections of ResidualAttentionBlock
// (open_clip/transformer.py:215,226-234,252-272) in forward and dX-backward at ViT-L sizes.
#include <type_traits>

#include "vl_gemm_common.h"

namespace {

constexpr int PK_STAGE = 65536, PK_ABYTES = 32768;
constexpr int PK_LDS = 2 * PK_STAGE + 32768;            // 160 KB: two operand stages + 8 x 4 KB transpose slabs
constexpr int PK_GN = 8;                                 // N-tiles per group (tile order, see tile_origin; in-step A/B of 2 / 4 / 8 / 16: profiles/r04_gemm_tile_order_ab.log)

// Measurement build only (tools/build_gemm_prof.sh, -DVL_GEMM_PROF): wave 0 of every workgroup adds up the shader-clock cycles
// it spends in the k-loops and in the epilogues: prof[4 * workgroup + {0: k-loop, 1: epilogue, 2: tiles}].
#ifdef VL_GEMM_PROF
}  // namespace
extern "C" { long* vl_gemm_prof_buf = nullptr; }
namespace {
#define PK_PROF_ARG , long* prof
#define PK_PROF_PASS , vl_gemm_prof_buf
#define PK_PROF_T(v) const long v = __builtin_amdgcn_s_memtime()
#define PK_PROF_ADD(i, d) do { if (prof && threadIdx.x == 0) prof[4 * blockIdx.x + (i)] += (d); } while (0)
#else
#define PK_PROF_ARG
#define PK_PROF_PASS
#define PK_PROF_T(v) do { } while (0)
#define PK_PROF_ADD(i, d) do { } while (0)
#endif

template <int I>
using IC = std::integral_constant<int, I>;

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// ACT (EPI_BF16 only): 0 none, 1 GELU, 2 ReLU, 3 GELU with the pre-activation also written to out2 (saved for backward),
// 4 GELU with gelu'(pre-activation) written to out2.  EPI_DGELU: ACT 4 = the aux operand is that saved gelu'.
// The main loop runs on `v_mfma_f32_16x16x32_*` (32 MFMAs of 16 384 flop per 32-deep half k-step from 8 + 4 fragments): against
// `v_mfma_f32_32x32x16_*` (rounds 2-3a; removed in round 5 with its A/B switch) the same LDS traffic and accumulator registers,
// a quarter of the accumulator read-modify-write per flop.  The board is power-limited on real data (profiles/
// r05_power_trace.log: 1 400 W, 1.86 GHz through a GEMM loop): a register-only loop of the 16x16x32 instruction sustains
// 2087 TF/s on random-normal operands against 1862 TF/s for 32x32x16 (profiles/r03b_mfma_power_probe.log).
// F16: operands (A, W) and the 16-bit output are IEEE half instead of bf16 (`v_mfma_f32_16x16x32_f16`, same rate, same LDS
// image; fp32 accumulation and epilogue arithmetic unchanged, the 16-bit store saturates at +-65504).  Instantiated for the
// frozen text tower's three launches (plain, GELU, fp32 residual): three more mantissa bits on every operand bring its
// cosine matrix to 1-2e-4 of the fp32 CPU path at ONE product per weight (two-term bf16 weights: 6-8e-4 at two).
template <int EPI, int ACT, bool F16>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
    gemm_nt_pk_kernel(const GemmP p PK_PROF_ARG) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr bool HAS_AUX = (EPI == EPI_RES_BF16 || EPI == EPI_DGELU);   // second operand of the output's shape
  // LayerNorm folding (round 4; GemmP::ln_*): ACT 10 / 11 / 14 = ACT 0 / 1 / 4 with the row-statistics epilogue
  // rstd_m * (acc - mean_m * c_n) + bias_n on the raw residual rows; EPI_RES_BF16 with ACT 20 also emits the partial row
  // statistics (sum, sum of squares per 64-column slice) of what it stores, for the LayerNorm that follows it
  constexpr bool LNF = (EPI == EPI_BF16 && ACT >= 10 && ACT < 20);
  constexpr int ACTB = LNF ? ACT - 10 : ACT;
  constexpr bool STATS = (EPI == EPI_RES_BF16 && ACT == 20);
  static_assert(!F16 || ((EPI == EPI_BF16 && (ACT == 0 || ACT == 1)) || (EPI == EPI_RES_F32 && ACT == 0)),
                "fp16 operands: plain / GELU 16-bit output and the fp32 residual epilogue only");
  // GEGLU (Perceiver feed-forward, perceiver.py:85-102): rows of W interleaved (a_j, gate_j) -> out[M, N/2] = a * gelu(gate),
  // optionally the bf16 pre-activation [M, N] to out2 (row stride 2 * ldo).  DGEGLU (its backward): acc = dy[M, N], res =
  // the saved pre-activation h[M, 2N]: out[M, 2N] = (dy * gelu(g), dy * a * gelu'(g)) interleaved (ldo = row stride of h / out).
  constexpr bool IS_GEGLU = (EPI == EPI_GEGLU), IS_DGEGLU = (EPI == EPI_DGEGLU);
  constexpr int NW = 8, NTL = 2, WTN = 64;   // 2 (M) x 4 (N) waves of 128 x 64
  constexpr int NU = 4;                      // 1 KB DMA units per operand per wave per k-step
  constexpr int NCH = 16;                    // output chunks per wave: chunk = 8 rows x 128 bytes = one store instruction
  constexpr int SLAB = 4096;                 // wave-private transpose slab: 32 rows x 64 columns bf16

  const int tiles_n = p.N >> 8, tiles_m = p.M >> 8;
  // split-K (EPI_F32 only: weight gradients, few output tiles and a very long reduction): work item = (k-slice, tile),
  // slice sp reduces k-steps [sp*nk, (sp+1)*nk) and writes its partial product to out + sp*split_stride
  const int nk = (EPI == EPI_F32 && p.ksplit_len) ? p.ksplit_len : (p.K >> 6);
  const int ntiles_mn = tiles_m * tiles_n;
  const int ntiles = ntiles_mn * ((EPI == EPI_F32 && p.ksplit_len) ? (p.K >> 6) / p.ksplit_len : 1);
  const int G = gridDim.x;
  const int slot = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  if (slot >= ntiles) return;
  const int my_tiles = (ntiles - slot + G - 1) / G;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wid & 1, wave_n = (wid >> 1) & 3;
  const bool grpB = wid >= 4;                // the second wave of each SIMD (waves w and w+4 share one)

  // tile order: groups of PK_GN consecutive N-tiles, M fastest inside a group, each XCD owns a contiguous run per round
  auto tile_origin = [&](int ti, int& m0, int& n0, int& sp) {
    int v = ti * G + slot;
    sp = 0;
    if constexpr (EPI == EPI_F32) { sp = v / ntiles_mn; v -= sp * ntiles_mn; }
    constexpr int GN = PK_GN;
    const int gsz = GN * tiles_m;
    const int gid = v / gsz, rem = v - gid * gsz;
    const int first_n = gid * GN;
    const int gn = min(tiles_n - first_n, GN);
    const int tm = rem / gn;
    m0 = tm << 8; n0 = (first_n + (rem - tm * gn)) << 8;
  };

  // ---- LDS-DMA: unit i of an operand = rows i*64 + wid*8 + (lane>>3), 16-byte chunk (lane&7) ^ swizzle(row) ----
  const int drow = wid * 8 + (lane >> 3);
  const int dsw = ((lane & 7) ^ ((drow >> 1) & 7)) * 16;            // (i*64 >> 1) is a multiple of 8: the swizzle does not depend on i
  const unsigned voffA = (unsigned)(drow * p.lda * 2 + dsw), voffW = (unsigned)(drow * p.ldw * 2 + dsw);
  const int a_unit = p.lda * 128, w_unit = p.ldw * 128;            // bytes between units (64 rows)
  __amdgpu_buffer_rsrc_t rsA, rsW;
  auto make_rsrc = [&](int m0, int n0, int sp, __amdgpu_buffer_rsrc_t& ra, __amdgpu_buffer_rsrc_t& rw) {
    const size_t k0 = (size_t)sp * nk * 64;        // first reduction index of the slice
    ra = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (size_t)m0 * p.lda + k0), 0, 0x7ffffff0, 0x00020000);
    rw = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (size_t)n0 * p.ldw + k0), 0, 0x7ffffff0, 0x00020000);
  };

  // ---- fragments (two sets: the reads of substep s+1 are in flight under the MFMAs of substep s) ----
  bf16x8 af[2][4], wf[2][4];
  // ---- the wave tile in 16x16 blocks: fragment = 16 rows x 32 k (lane: row lane&15, 8-wide k chunk lane>>4), A rows
  // mh*64 + ia*16, W rows jb*16; accumulator block [ib][jb]: lane owns output row ib*16 + (lane&15) and the four columns
  // jb*16 + (lane>>4)*4 .. +3.  Same XOR swizzle ((row>>1)&7 on the 16-byte chunk index; block bases are multiples of 16
  // rows), conflict-free in the four 16-lane groups of ds_read_b128.
  const int fr16 = lane & 15, fq = lane >> 4;
  const int fsw16 = (fr16 >> 1) & 7;
  const int fa16 = (wave_m * 128 + fr16) * 128, fw16 = PK_ABYTES + (wave_n * WTN + fr16) * 128;
  f32x4 acc16[8][4];
  auto ldA16 = [&](const unsigned char* stage, int h, int mh, int c) {
    const int off = ((h * 4 + fq) ^ fsw16) * 16;
#pragma unroll
    for (int ia = 0; ia < 4; ++ia) af[c][ia] = *(const bf16x8*)(stage + fa16 + (mh * 4 + ia) * 2048 + off);
  };
  auto ldW16 = [&](const unsigned char* stage, int h, int c, auto J0, auto J1) {
    const int off = ((h * 4 + fq) ^ fsw16) * 16;
#pragma unroll
    for (int jb = decltype(J0)::value; jb < decltype(J1)::value; ++jb) wf[c][jb] = *(const bf16x8*)(stage + fw16 + jb * 2048 + off);
  };
  auto mma16 = [&](auto MH, int ca, int cw) {
    constexpr int mh = decltype(MH)::value;
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const int jb = o, ia = n;      // (W-fragment-outer order; A-outer and a static s_setprio for waves 4-7 measured equal, profiles/r03e_*)
        if constexpr (F16)
          acc16[mh * 4 + ia][jb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf[cw][jb]), __builtin_bit_cast(f16x8, af[ca][ia]),
                                                                          acc16[mh * 4 + ia][jb], 0, 0, 0);
        else
          acc16[mh * 4 + ia][jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[cw][jb], af[ca][ia], acc16[mh * 4 + ia][jb], 0, 0, 0);
      }
  };
  auto first_frags = [&](const unsigned char* stage) {       // fragments of a stage's first phase
    ldA16(stage, 0, 0, 0); ldW16(stage, 0, 0, IC<0>{}, IC<4>{});
  };
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  // ---- epilogue operands ----
  // chunk ci (0..15) = rows (ci>>2)*32 + (ci&3)*8 + (lane>>3) of the wave's sub-tile, columns (lane&7)*8 .. +7
  [[maybe_unused]] u32x4 aux[2][4];          // second operand of row block i in aux[i & 1] (double buffered)
  // (the epilogue's lane constants are derived at the START OF EVERY EPILOGUE from a laundered copy of the lane index: computed
  // here they would be hoisted above the k-loop, live through it beside 128 accumulators + 64 fragment registers, and spill)
  int prow = 0, pcol = 0;
  unsigned lo_out = 0;                                               // lane part of an output / aux address (row-major outputs)
  const int ldo2 = p.ldo * 2;
  auto chunk_row = [](int ci) { return (ci >> 2) * 32 + (ci & 3) * 8; };
  [[maybe_unused]] const unsigned char* aux_src = nullptr;      // &aux[tile row 0 of this wave][tile col 0 of this wave] (current tile)
  // the four 16-byte pieces of row block I (rows I*32 + pass*8 + (lane>>3)), requested one row block ahead: the lines were
  // touched at the start of the tile (L2 / Infinity Cache), 32 registers instead of the 64 of loading the whole tile up front
  auto load_block = [&](auto II) {
    constexpr int i = decltype(II)::value;
    if constexpr (HAS_AUX && i < 4) {
#pragma unroll
      for (int pass = 0; pass < 4; ++pass)
        aux[i & 1][pass] = *(const u32x4*)(aux_src + (size_t)(chunk_row(i * 4 + pass) * ldo2) + lo_out);
    }
  };

  int cur_m0, cur_n0, cur_sp;
  tile_origin(0, cur_m0, cur_n0, cur_sp);
  auto set_aux = [&]() {
    if constexpr (HAS_AUX) {
      // EPI_RES_BF16 indexes its residual by the absolute row (m + m_off) with an un-offset pointer (vl_gemm.hip run_gemm)
      const int moff = EPI == EPI_RES_BF16 ? p.m_off : 0;
      aux_src = (const unsigned char*)p.res + ((size_t)(cur_m0 + moff + wave_m * 128) * p.ldo + cur_n0 + wave_n * WTN) * 2;
    }
  };
  make_rsrc(cur_m0, cur_n0, cur_sp, rsA, rsW);
  // DMA position (dti, dkt) = the step whose operands the next DMA batch loads; it runs two k-steps ahead of the MFMAs and
  // therefore enters the next tile at kt = nk-2: that tile's descriptors are prepared once per tile, outside the k-loop
  __amdgpu_buffer_rsrc_t rsA_n = rsA, rsW_n = rsW;
  int dti = 0, dkt = 0;
  auto dma_step = [&](unsigned char* stage) {
    const int kbyte = dkt << 7;
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(stage + (i * NW + wid) * 1024), 16, voffA, kbyte + i * a_unit, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr_t)(stage + PK_ABYTES + (i * NW + wid) * 1024), 16, voffW, kbyte + i * w_unit, 0, 0);
    }
    ++dkt;
    if (dkt == nk) { dkt = 0; ++dti; rsA = rsA_n; rsW = rsW_n; }
  };
  // hipcc does not wait for this builtin's LDS writes in front of a barrier: wait by hand.  The barrier is the raw
  // instruction: __syncthreads() carries a release fence, i.e. a compiler vmcnt(0)/lgkmcnt(0) for everything else, which
  // is what we want to control here.  The only cross-wave LDS traffic is the DMA (vmcnt) and fragment READS (lgkmcnt).
  // vmcnt(0) also retires the previous tile's output stores (in-order counter): they have ~1.5 k-steps to drain before
  // they can delay a barrier.
#ifdef PK_X_NOBAR      // timing-only ablation builds (tools/pk_ablation.py; results are garbage): no s_barrier / no DMA / no fragment reads
#define PK_BAR ""
#else
#define PK_BAR "\n\ts_barrier"
#endif
  auto dma_wait_and_barrier = [&]() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" PK_BAR ::: "memory");
  };
  // The FIRST barrier of a tile behind an epilogue (round 4): the DMA it needs (k-step 1 of this tile) was issued in front of
  // the epilogue, i.e. it is OLDER than the epilogue's output stores in the in-order vmcnt queue - waiting for "at most
  // NST operations outstanding" retires exactly that DMA and leaves the NST stores of this wave draining under the first
  // k-step instead of in front of it (measured with vmcnt(0): 0.2-0.7 k cycles per k-step of store drain charged to the
  // k-loop, 11 k per tile behind the two-output GELU epilogue).  NST = the minimum number of stores a wave issues per tile.
  // Derived from the constants that generate the stores: every epilogue walks NCH = 16 chunks (4 row blocks x 4 passes) and
  // issues ST_PER_CHUNK vector-memory stores per chunk on EVERY lane-uniform path - fp32 outputs two 32-column halves, DGEGLU
  // two 16-byte halves, the two-output GELU variants out + out2, STATS the bf16 chunk + one 8-byte partial-sum store per pass
  // (issued by the wave even where only lanes 0, 8, .. are active), GEGLU one (its optional pre-activation output does not
  // count: NST is a MINIMUM).  An epilogue that issued fewer stores than NST would let this barrier pass before k-step 1 of
  // the tile has landed; -DVL_GEMM_SAFE_WAIT builds wait for everything (debug A/B of exactly that failure).
  constexpr int ST_PER_CHUNK = (EPI == EPI_F32 || EPI == EPI_RES_F32) ? NTL
                               : (IS_DGEGLU || STATS || (EPI == EPI_BF16 && (ACTB == 3 || ACTB == 4))) ? 2 : 1;
#ifdef VL_GEMM_SAFE_WAIT
  constexpr int NST = 0;
#else
  constexpr int NST = NCH * ST_PER_CHUNK;
#endif
  static_assert(NCH == 4 * 4 && NST <= 63, "NST = stores per wave and tile; vmcnt is a 6-bit counter");
  static_assert(!(EPI == EPI_BF16 && (ACTB == 3 || ACTB == 4)) || ST_PER_CHUNK == 2, "two-output epilogues store twice per chunk");
  auto first_wait_and_barrier = [&]() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" PK_BAR ::"n"(NST) : "memory");
  };

  zero_acc();
  dma_step(smem);
  dma_step(smem + PK_STAGE);                     // nk >= 8: still inside tile 0
  dma_wait_and_barrier();
  first_frags(smem);

  int par = 0;                                   // stage buffer of the k-step being computed
  bool pendB = false;                            // group B: a DMA batch is due at the top of the next k-step
  bool after_epi = false;                        // the next barrier is the first one behind an epilogue (wave-uniform)
  auto kstep = [&](auto LAST) {
    constexpr bool last = decltype(LAST)::value;
    unsigned char* cur = smem + par * PK_STAGE;
    unsigned char* oth = smem + (par ^ 1) * PK_STAGE;
#ifndef PK_X_NODMA
    if (grpB && pendB) { dma_step(oth); pendB = false; }
#endif
    __builtin_amdgcn_sched_barrier(0);
    // four phases of 16 MFMAs: (k half 0, rows 0-63) (0, 64-127) (1, 0-63) (1, 64-127); W fragments of a half stay for both
#ifdef PK_X_NOREAD
    mma16(IC<0>{}, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    mma16(IC<1>{}, 1, 0);
    __builtin_amdgcn_sched_barrier(0);
    mma16(IC<0>{}, 0, 1);
#else
    ldA16(cur, 0, 1, 1); ldW16(cur, 1, 1, IC<0>{}, IC<2>{});
    mma16(IC<0>{}, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    ldA16(cur, 1, 0, 0); ldW16(cur, 1, 1, IC<2>{}, IC<4>{});
    mma16(IC<1>{}, 1, 0);
    __builtin_amdgcn_sched_barrier(0);
    ldA16(cur, 1, 1, 1);
    mma16(IC<0>{}, 0, 1);
#endif
    __builtin_amdgcn_sched_barrier(0);
    if (after_epi) { first_wait_and_barrier(); after_epi = false; } else dma_wait_and_barrier();
    // (at the last k-step of a tile BOTH wave groups issue at once: the batch must sit in front of the epilogue's stores)
#ifndef PK_X_NODMA
    if (dti < my_tiles) { if (!grpB || last) dma_step(cur); else pendB = true; }
#endif
#ifndef PK_X_NOREAD
    if constexpr (!last) first_frags(oth);
#endif
    __builtin_amdgcn_sched_barrier(0);
    mma16(IC<1>{}, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    par ^= 1;
  };

  for (int ti = 0; ti < my_tiles; ++ti) {
    if (ti + 1 < my_tiles) {
      int nm0, nn0, nsp;
      tile_origin(ti + 1, nm0, nn0, nsp);
      make_rsrc(nm0, nn0, nsp, rsA_n, rsW_n);
    }
    set_aux();
    PK_PROF_T(t_a);
    after_epi = ti > 0;
    for (int kt = 0; kt < nk - 1; ++kt) kstep(std::false_type{});
    kstep(std::true_type{});
    PK_PROF_T(t_b);
    {
      // ---------------- tile finished: arithmetic, LDS transpose, one burst of 16-byte non-temporal stores ----------------
      const GemmP pe = reload_params();
      mfma_results_settled();
      const int mrow0 = cur_m0 + wave_m * 128, ncol0 = cur_n0 + wave_n * WTN;
      int el = lane;
      asm volatile("" : "+v"(el));
      prow = el >> 3; pcol = (el & 7) * 8;
      lo_out = (unsigned)((prow * pe.ldo + pcol) * 2);
      const int fr16 = el & 15, fq = el >> 4;      // shadow the main loop's copies
      if constexpr (HAS_AUX) {
        load_block(IC<0>{});
      }
      unsigned char* const slab = smem + 2 * PK_STAGE + wid * SLAB;
      // branch-free optional bias: read SOMETHING valid (the weight matrix) and select zero
      const bool has_bias = pe.bias != nullptr;
      const float* const bsrc = has_bias ? pe.bias : (const float*)pe.W;
      // The bias of this lane's columns, ONCE per tile (round 4).  It was loaded where it is used, inside the row-block loops:
      // 32 loads of 16 bytes per tile, each followed by the compiler's `s_waitcnt vmcnt(0)` - and on gfx9 that counter also
      // covers the output STORES issued just before, so every row block first waited for the previous block's stores to
      // reach memory and then paid eight serial L2 round trips.  tools/gemm_phase_prof.py: the plain epilogue took 10.4 k
      // cycles per 256x256 tile (21 % of a K = 1024 launch) with ~0.8 k cycles of VALU work in it.
      [[maybe_unused]] f32x4 bvq[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int n = ncol0 + c * 16 + fq * 4;
        if constexpr (IS_DGEGLU) {                 // (never has a bias: vl_gemm_park_supported)
          bvq[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
          bvq[c] = *(const f32x4*)(bsrc + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) bvq[c][e] = has_bias ? bvq[c][e] : 0.f;
        }
      }
      // LayerNorm folding: c_n of this lane's columns and (rstd, -mean * rstd) of its 8 rows, once per tile
      [[maybe_unused]] f32x4 cvq[4];
      [[maybe_unused]] float ln_r[8], ln_nm[8];
      if constexpr (LNF) {
#pragma unroll
        for (int c = 0; c < 4; ++c) cvq[c] = *(const f32x4*)(pe.ln_c + ncol0 + c * 16 + fq * 4);
#pragma unroll
        for (int rb = 0; rb < 8; ++rb) {                       // row block rb = 2 * i + j: rows mrow0 + rb * 16 + fr16
          const float mu = pe.ln_mean[mrow0 + rb * 16 + fr16];
          ln_r[rb] = pe.ln_rstd[mrow0 + rb * 16 + fr16];
          ln_nm[rb] = -mu * ln_r[rb];
        }
      }
      // destination of this lane's chunks
      unsigned char* const out_base = (unsigned char*)pe.out + ((size_t)mrow0 * pe.ldo + ncol0) * 2 + lo_out;
      // GEGLU / DGEGLU lane offsets (other strides than the plain outputs)
      [[maybe_unused]] const unsigned lo_half = (unsigned)((prow * pe.ldo + (el & 7) * 4) * 2);        // GEGLU out: 4 values per lane
      [[maybe_unused]] const unsigned lo_pre = (unsigned)((prow * 2 * pe.ldo + pcol) * 2);               // GEGLU out2: stride 2 * ldo
      [[maybe_unused]] const unsigned lo_h = (unsigned)((prow * pe.ldo + (el & 7) * 16) * 2);           // DGEGLU h / out: 16 values per lane
      [[maybe_unused]] u32x4 hx[2][8];                                                                  // DGEGLU: h of row block i (double buffered)
      [[maybe_unused]] const unsigned char* const hsrc = (const unsigned char*)pe.res + ((size_t)mrow0 * pe.ldo + 2 * ncol0) * 2 + lo_h;
      [[maybe_unused]] auto load_h = [&](auto BI, auto II) {
        constexpr int bi = decltype(BI)::value, ii = decltype(II)::value;
        if constexpr (IS_DGEGLU && ii < 4) {
#pragma unroll
          for (int pass = 0; pass < 4; ++pass) {
            const unsigned char* src = hsrc + (size_t)((ii * 32 + pass * 8) * ldo2);
            hx[bi][pass * 2] = *(const u32x4*)src;
            hx[bi][pass * 2 + 1] = *(const u32x4*)(src + 16);
          }
        }
      };
      if constexpr (IS_DGEGLU) load_h(IC<0>{}, IC<0>{});
      if constexpr (EPI == EPI_F32 || EPI == EPI_RES_F32) {
        // fp32 outputs: partial product of a k-slice (EPI_F32) or res + acc * alpha + bias into an fp32 residual stream
        // (EPI_RES_F32, in place allowed; round 4 - the Perceiver's residual projections and the two-term text tower ran on
        // the round-1 kernel until then): 32x32 blocks through the slab, 16-byte stores (8 lanes per 128-byte line)
        float* const fout = (float*)pe.out + (EPI == EPI_F32 ? (size_t)cur_sp * pe.split_stride : (size_t)0) + (size_t)mrow0 * pe.ldo + ncol0;
        [[maybe_unused]] const float* const rin = (const float*)pe.res + (size_t)mrow0 * pe.ldo + ncol0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int j = 0; j < NTL; ++j) {
            [[maybe_unused]] f32x4 rr[4];
            if constexpr (EPI == EPI_RES_F32) {        // the block's residual values, requested ahead of the slab round trip
#pragma unroll
              for (int pass = 0; pass < 4; ++pass)
                rr[pass] = *(const f32x4*)(rin + (size_t)(i * 32 + pass * 8 + prow) * pe.ldo + j * 32 + (el & 7) * 4);
            }
#pragma unroll
            for (int ibh = 0; ibh < 2; ++ibh)
#pragma unroll
              for (int jbh = 0; jbh < 2; ++jbh) {
                const int row = ibh * 16 + fr16;
                f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                if constexpr (EPI == EPI_RES_F32) bv = bvq[j * 2 + jbh];
                *(f32x4*)(slab + row * 128 + (((jbh * 4 + fq) ^ (row & 7)) << 4)) = scale_bias(acc16[i * 2 + ibh][j * 2 + jbh], pe.alpha, bv);
              }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
              const int r = pass * 8 + prow;
              f32x4 w = *(const f32x4*)(slab + r * 128 + (((el & 7) ^ (r & 7)) << 4));
              if constexpr (EPI == EPI_RES_F32) w = w + rr[pass];
              __builtin_nontemporal_store(w, (f32x4*)(fout + (size_t)(i * 32 + r) * pe.ldo + j * 32 + (el & 7) * 4));
            }
          }
        }
      } else {
      auto row_block = [&](auto II) {
        constexpr int i = decltype(II)::value;
        if constexpr (IS_DGEGLU) load_h(IC<(i + 1) & 1>{}, IC<i + 1>{});        // h of the next row block, under this block's work
        if constexpr (HAS_AUX) load_block(IC<i + 1>{});
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            // (j, q) = (16-row half, 16-column block) of the 32 x 64 row block
            f32x4 v = acc16[i * 2 + j][q];
            if constexpr (LNF) {
              // rstd * acc + (bias - mean * rstd * c): two FMAs per value (alpha is 1 on this path)
#pragma unroll
              for (int e = 0; e < 4; ++e)
                v[e] = __builtin_fmaf(ln_r[i * 2 + j], v[e], __builtin_fmaf(ln_nm[i * 2 + j], cvq[q][e], bvq[q][e]));
            } else {
              v = scale_bias(v, pe.alpha, bvq[q]);
            }
            if constexpr (EPI == EPI_BF16 && ACTB == 1) {
#pragma unroll
              for (int e = 0; e < 4; e += 2) {
                const vl_f32x2 y = gelu_erf2(vl_f32x2{v[e], v[e + 1]});
                v[e] = y[0]; v[e + 1] = y[1];
              }
            } else if constexpr (EPI == EPI_BF16 && ACTB == 2) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            u32x2 o;
            if constexpr (F16) { o[0] = pack2h(v[0], v[1]); o[1] = pack2h(v[2], v[3]); }
            else { o[0] = pack2bf(v[0], v[1]); o[1] = pack2bf(v[2], v[3]); }
            const int row = j * 16 + fr16;
            *(u32x2*)(slab + row * 128 + (((q * 2 + (fq >> 1)) ^ (row & 7)) << 4) + (fq & 1) * 8) = o;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        [[maybe_unused]] float st1[4], st2[4];                 // STATS: this lane's partial row sums of the four passes
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
          const int r = pass * 8 + prow;
          u32x4 w = *(const u32x4*)(slab + r * 128 + (((el & 7) ^ (r & 7)) << 4));
          [[maybe_unused]] u32x4 rr;
          if constexpr (HAS_AUX) rr = aux[i & 1][pass];
          if constexpr (IS_GEGLU) {
            // w = 4 (a, gate) pairs of the bf16 pre-activation (the reference's autocast multiplies the bf16 halves too)
            const size_t rowoff = (size_t)(mrow0 + i * 32 + pass * 8);
            if (pe.out2)
              __builtin_nontemporal_store(w, (u32x4*)((unsigned char*)pe.out2 + (rowoff * 2 * pe.ldo + ncol0) * 2 + lo_pre));
            u32x2 o;
#pragma unroll
            for (int e = 0; e < 2; ++e) {       // (a, gate) of two neighbouring hidden columns: the two gates as one packed pair
              const vl_f32x2 a = {bf2f((bf16_t)(w[2 * e] & 0xffff)), bf2f((bf16_t)(w[2 * e + 1] & 0xffff))};
              const vl_f32x2 gt = {bf2f((bf16_t)(w[2 * e] >> 16)), bf2f((bf16_t)(w[2 * e + 1] >> 16))};
              o[e] = pack2bf(a * gelu_erf2(gt));
            }
            __builtin_nontemporal_store(o, (u32x2*)((unsigned char*)pe.out + (rowoff * pe.ldo + (ncol0 >> 1)) * 2 + lo_half));
            continue;
          } else if constexpr (IS_DGEGLU) {
            // w = dy of 8 hidden columns (bf16); hx = their 8 (a, gate) pairs -> 8 (d a, d gate) pairs
            u32x4 o2[2];
#pragma unroll
            for (int c = 0; c < 8; c += 2) {      // two hidden columns per iteration: the arithmetic in packed-fp32 pairs
              const unsigned hv0 = hx[i & 1][pass * 2 + (c >> 2)][c & 3], hv1 = hx[i & 1][pass * 2 + (c >> 2)][(c + 1) & 3];
              const vl_f32x2 dy = unpack2bf(w[c >> 1]);
              const vl_f32x2 a = {bf2f((bf16_t)(hv0 & 0xffff)), bf2f((bf16_t)(hv1 & 0xffff))};
              const vl_f32x2 g = {bf2f((bf16_t)(hv0 >> 16)), bf2f((bf16_t)(hv1 >> 16))};
              const GeluParts2 gp = gelu_parts2(g);
              const vl_f32x2 da = dy * gelu_from_parts2(g, gp), dg = dy * a * gelu_grad_from_parts2(g, gp);
              o2[c >> 2][c & 3] = pack2bf(da[0], dg[0]);
              o2[c >> 2][(c + 1) & 3] = pack2bf(da[1], dg[1]);
            }
            unsigned char* dst = (unsigned char*)pe.out + ((size_t)(mrow0 + i * 32 + pass * 8) * pe.ldo + 2 * ncol0) * 2 + lo_h;
            __builtin_nontemporal_store(o2[0], (u32x4*)dst);
            __builtin_nontemporal_store(o2[1], (u32x4*)(dst + 16));
            continue;
          } else if constexpr (EPI == EPI_BF16 && ACTB == 3) {
            __builtin_nontemporal_store(w, (u32x4*)((unsigned char*)pe.out2 + ((size_t)(mrow0 + i * 32 + pass * 8) * pe.ldo + ncol0) * 2 + lo_out));
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = pack2bf(gelu_erf2(unpack2bf(w[e])));
          } else if constexpr (EPI == EPI_BF16 && ACTB == 4) {
            u32x4 d;                                              // out2 = gelu'(pre) for the dX GEMM of the backward
#pragma unroll
            for (int e = 0; e < 4; ++e) { unsigned int y, g; gelu_and_grad_pk(w[e], y, g); w[e] = y; d[e] = g; }
            __builtin_nontemporal_store(d, (u32x4*)((unsigned char*)pe.out2 + ((size_t)(mrow0 + i * 32 + pass * 8) * pe.ldo + ncol0) * 2 + lo_out));
          } else if constexpr (EPI == EPI_RES_BF16) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float lo = bf2f((bf16_t)(w[e] & 0xffff)) + bf2f((bf16_t)(rr[e] & 0xffff));
              const float hi = bf2f((bf16_t)(w[e] >> 16)) + bf2f((bf16_t)(rr[e] >> 16));
              w[e] = pack2bf(lo, hi);
            }
            if constexpr (STATS) {
              // (sum, sum of squares) of the 8 STORED values (bf16-rounded: what the next LayerNorm will read) of this lane:
              // two bf16 dot products per dword; the cross-lane part waits for the end of the row block (below)
              // (inline assembly: through __builtin_amdgcn_fdot2_f32_bf16 hipcc 7.2 fed all four dot products of this loop
              //  the FIRST dword of w - found by tests/test_hip_lnfold.py, visible in the ISA)
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
          } else if constexpr (EPI == EPI_DGELU && ACT == 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = mul_pk_bf16(w[e], rr[e]);     // aux = gelu' saved by the forward
          } else if constexpr (EPI == EPI_DGELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              w[e] = pack2bf(unpack2bf(w[e]) * gelu_erf_grad2(unpack2bf(rr[e])));
          }
          __builtin_nontemporal_store(w, (u32x4*)(out_base + (size_t)((i * 32 + pass * 8) * ldo2)));
        }
        if constexpr (STATS) {
          // the 8 lanes that hold a row's 64 columns of this wave: eight independent three-step reductions (lanes ^1, ^2,
          // mirror of the half row), then one 8-byte store per row and 64-column slice
          asm volatile("s_nop 1" ::: "memory");     // the sums come out of inline assembly: the DPP read's wait states by hand
#pragma unroll
          for (int pass = 0; pass < 4; ++pass) { st1[pass] += dpp_f<0xB1>(st1[pass]); st2[pass] += dpp_f<0xB1>(st2[pass]); }
#pragma unroll
          for (int pass = 0; pass < 4; ++pass) { st1[pass] += dpp_f<0x4E>(st1[pass]); st2[pass] += dpp_f<0x4E>(st2[pass]); }
#pragma unroll
          for (int pass = 0; pass < 4; ++pass) { st1[pass] += dpp_f<0x141>(st1[pass]); st2[pass] += dpp_f<0x141>(st2[pass]); }
          if ((el & 7) == 0) {
            float* dst = pe.row_part + ((size_t)(mrow0 + i * 32 + prow) * (pe.N >> 6) + (ncol0 >> 6)) * 2;
#pragma unroll
            for (int pass = 0; pass < 4; ++pass)
              *(vl_f32x2*)(dst + (size_t)pass * 8 * (pe.N >> 6) * 2) = vl_f32x2{st1[pass], st2[pass]};
          }
        }
      };
      row_block(IC<0>{}); row_block(IC<1>{}); row_block(IC<2>{}); row_block(IC<3>{});
      }
      zero_acc();
      if (ti + 1 < my_tiles) { tile_origin(ti + 1, cur_m0, cur_n0, cur_sp); first_frags(smem + par * PK_STAGE); }
#ifdef VL_GEMM_PROF
      { PK_PROF_T(t_c); PK_PROF_ADD(0, t_b - t_a); PK_PROF_ADD(1, t_c - t_b); PK_PROF_ADD(2, 1); }
#endif
    }
  }
}

template <int EPI, int ACT, bool F16 = false>
hipError_t launch_pk(const GemmP& p, int ncu, hipStream_t s) {
  auto kern = gemm_nt_pk_kernel<EPI, ACT, F16>;
  static const hipError_t attr = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, PK_LDS);   // thread-safe one-time init
  if (attr != hipSuccess) return attr;
  const int tiles = (p.M >> 8) * (p.N >> 8) * ((EPI == EPI_F32 && p.ksplit_len) ? (p.K >> 6) / p.ksplit_len : 1);
  int G = ncu & ~7;
  if (tiles < G) G = (tiles + 7) & ~7;
  hipLaunchKernelGGL(kern, dim3(G), dim3(512), PK_LDS, s, p PK_PROF_PASS);
  return hipGetLastError();
}

}  // namespace

// Internal entries used by vl_gemm.hip's dispatcher (not part of the public C ABI).
bool vl_gemm_park_supported(int epi, const void* params) {
  const GemmP& p = *(const GemmP*)params;
  if (epi == EPI_F32) {       // split-K partial products into a workspace (vl_gemm_splitk_accum_f32)
    if ((p.M & 255) || (p.N & 255) || (p.K & 63) || p.M <= 0 || p.N <= 0 || (p.ldo & 3) || p.bias) return false;
    const int nk = p.K >> 6;
    if (p.ksplit_len < 2 || nk % p.ksplit_len) return false;
    return !((((uintptr_t)p.A | (uintptr_t)p.W | (uintptr_t)p.out) & 15));
  }
  if (epi == EPI_RES_F32) {   // fp32 residual stream: out = res + acc * alpha + bias (in place allowed)
    if ((p.M & 255) || (p.N & 255) || (p.K & 63) || p.K < 512 || p.M <= 0 || p.N <= 0 || (p.ldo & 3) || !p.res || !p.out) return false;
    if (p.res_div != 1 || p.act != 0 || p.out2 || p.ksplit_len) return false;
    return !((((uintptr_t)p.A | (uintptr_t)p.W | (uintptr_t)p.out | (uintptr_t)p.res) & 15));
  }
  if (!(epi == EPI_BF16 || epi == EPI_RES_BF16 || epi == EPI_DGELU || epi == EPI_GEGLU || epi == EPI_DGEGLU)) return false;
  if (epi == EPI_GEGLU && p.act != 0) return false;
  if (epi == EPI_DGEGLU && (p.bias || !p.res)) return false;
  // whole tiles; the DMA prologue issues two k-steps of tile 0 up front
  if ((p.M & 255) || (p.N & 255) || (p.K & 63) || p.K < 512 || p.M <= 0 || p.N <= 0) return false;
  if (p.res_div != 1) return false;
  if (epi == EPI_RES_BF16 && p.act != 0) return false;
  if (epi == EPI_BF16 && p.out2 && p.act != 1 && p.act != 4) return false;
  if (epi == EPI_BF16 && p.act == 4 && !p.out2) return false;
  if (p.ldo & 7) return false;
  // 16-byte accesses on every operand
  if (((uintptr_t)p.A | (uintptr_t)p.W) & 15) return false;
  if (((uintptr_t)p.out | (uintptr_t)p.res | (uintptr_t)p.out2) & 15) return false;
  return true;
}

int vl_gemm_park_launch(int epi, const void* params, int ncu, hipStream_t s) {
  const GemmP& p = *(const GemmP*)params;
  if (p.f16) {                // IEEE-half operands (vl_gemm_f16): the frozen text tower's three launches
    if (p.ln_mean || p.row_part || p.out2) return (int)hipErrorInvalidValue;
    if (epi == EPI_BF16 && p.act == 0) return (int)launch_pk<EPI_BF16, 0, true>(p, ncu, s);
    if (epi == EPI_BF16 && p.act == 1) return (int)launch_pk<EPI_BF16, 1, true>(p, ncu, s);
    if (epi == EPI_RES_F32) return (int)launch_pk<EPI_RES_F32, 0, true>(p, ncu, s);
    return (int)hipErrorInvalidValue;
  }
  switch (epi) {
    case EPI_BF16:
      if (p.ln_mean) {        // LayerNorm folded into the epilogue (vl_gemm_lnfold_bf16 checked the rest)
        if (p.act == 1) return (int)launch_pk<EPI_BF16, 11>(p, ncu, s);
        if (p.act == 4) return (int)launch_pk<EPI_BF16, 14>(p, ncu, s);
        return p.act == 0 ? (int)launch_pk<EPI_BF16, 10>(p, ncu, s) : (int)hipErrorInvalidValue;
      }
      if (p.act == 1) return p.out2 ? (int)launch_pk<EPI_BF16, 3>(p, ncu, s) : (int)launch_pk<EPI_BF16, 1>(p, ncu, s);
      if (p.act == 4) return (int)launch_pk<EPI_BF16, 4>(p, ncu, s);
      return p.act == 2 ? (int)launch_pk<EPI_BF16, 2>(p, ncu, s) : (int)launch_pk<EPI_BF16, 0>(p, ncu, s);
    case EPI_RES_BF16:
      if (p.row_part) return (int)launch_pk<EPI_RES_BF16, 20>(p, ncu, s);
      return (int)launch_pk<EPI_RES_BF16, 0>(p, ncu, s);
    case EPI_DGELU: return p.act == 4 ? (int)launch_pk<EPI_DGELU, 4>(p, ncu, s) : (int)launch_pk<EPI_DGELU, 0>(p, ncu, s);
    case EPI_GEGLU: return (int)launch_pk<EPI_GEGLU, 0>(p, ncu, s);
    case EPI_DGEGLU: return (int)launch_pk<EPI_DGEGLU, 0>(p, ncu, s);
    case EPI_F32: return (int)launch_pk<EPI_F32, 0>(p, ncu, s);
    case EPI_RES_F32: return (int)launch_pk<EPI_RES_F32, 0>(p, ncu, s);
    default: return (int)hipErrorInvalidValue;
  }
}
