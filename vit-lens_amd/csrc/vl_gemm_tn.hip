// Weight-gradient GEMM on TOKEN-MAJOR operands (round 3b): P[s][M, N] = sum over the tokens t of slice s of
// At[t, m] * Bt[t, n], At = dY [T, M] and Bt = X [T, N] exactly as the backward holds them (row = token).  The round-1/2 path
// transposed both operands first (`transpose64_kernel`, 2.1 % of the C3 step: 396 launches per 3 steps reading and writing
// 0.13-0.54 GB each) so that the reduction index became the contiguous one the NT kernels want.  Here the k-slab of a tile is
// staged token-major by the same LDS-DMA (a token row of the tile = 512 contiguous bytes) and the MFMA fragments - 8
// consecutive TOKENS of one output row per lane - come out of gfx950's LDS transpose read: `ds_read_b64_tr_b16` hands lane i of
// a 16-lane group column i of a [4 tokens][16 columns] block whose rows the lanes address four at a time
// (tools/tr_probe.hip prints the mapping this relies on).
//
// Structure = the persistent kernel of vl_gemm_park.hip with the 16x16x32 main loop: 256x256 output tiles, 64 tokens per
// k-step, 8 waves (2 x 4, 128 x 64 each), two 64 KB stages, one barrier per k-step in front of the last phase, work item =
// (k-slice, tile), fp32 partial products through the wave-private LDS slab, summed in fixed order by splitk_reduce_kernel.
//
// LDS image of one operand stage: row t (token, 512 B) = 16 units of 32 B (unit u = columns 16u..16u+15 = one fragment's
// columns); unit u of row t sits at physical unit u ^ g(t), g(t) = (t & 3) | ((t >> 3) & 1) << 2: the 8 rows one 32-lane half
// of a transpose read touches (t0..t0+3 and t0+8..t0+11) land on 8 different 32-byte bank groups.
//
// Replaces: the autograd weight gradients of nn.Linear / in_proj / out_proj in the unlocked blocks
// (open_clip/transformer.py:215,226-234,252-272 under loss.backward(), training/train.py:212-216).
#include <type_traits>

#include "vl_gemm_common.h"

namespace {

constexpr int TN_STAGE = 65536, TN_ABYTES = 32768;
constexpr int TN_LDS = 2 * TN_STAGE + 32768;            // two operand stages + 8 x 4 KB transpose slabs

template <int I>
using IC = std::integral_constant<int, I>;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_t;

struct Frag2 { s16x4 lo, hi; };

__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
    gemm_tn_kernel(const GemmP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NW = 8, WTN = 64, NU = 4, SLAB = 4096;
  const int tiles_n = p.N >> 8, tiles_m = p.M >> 8;
  const int nk_len = p.ksplit_len, nk_tot = p.K >> 6;        // k-steps (64 tokens) per slice (the last slice may be shorter), in all
  const int ntiles_mn = tiles_m * tiles_n;
  const int ntiles = ntiles_mn * ((nk_tot + nk_len - 1) / nk_len);
  auto steps_of = [&](int sp) { return min(nk_len, nk_tot - sp * nk_len); };
  const int G = gridDim.x;
  const int slot = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  if (slot >= ntiles) return;
  const int my_tiles = (ntiles - slot + G - 1) / G;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wid & 1, wave_n = (wid >> 1) & 3;
  const bool grpB = wid >= 4;

  // work item -> (k-slice, tile): the tiles of one slice are consecutive (its token range streams through L2 once per column tile)
  auto tile_origin = [&](int ti, int& m0, int& n0, int& sp) {
    int v = ti * G + slot;
    sp = v / ntiles_mn; v -= sp * ntiles_mn;
    const int tm = v / tiles_n;
    m0 = tm << 8; n0 = (v - tm * tiles_n) << 8;
  };

  // ---- LDS-DMA: unit i of an operand = token rows i*16 + wid*2 + (lane>>5), 16-byte slot lane&31 of the row ----
  const int drow = wid * 2 + (lane >> 5);
  const int dg = (drow & 3) | (((drow >> 3) & 1) << 2);
  const int dslot = lane & 31;
  const int dcol = (((dslot >> 1) ^ dg) << 4) + ((dslot & 1) << 3);          // first of the 8 columns this lane fetches
  const unsigned voffA = (unsigned)((drow * p.lda + dcol) * 2), voffB = (unsigned)((drow * p.ldw + dcol) * 2);
  const int a_unit = p.lda * 32, b_unit = p.ldw * 32;                         // bytes between units (16 token rows)
  const int a_kstep = p.lda * 128, b_kstep = p.ldw * 128;                     // bytes between k-steps (64 token rows)
  __amdgpu_buffer_rsrc_t rsA, rsB;
  auto make_rsrc = [&](int m0, int n0, int sp, __amdgpu_buffer_rsrc_t& ra, __amdgpu_buffer_rsrc_t& rb) {
    const size_t t0 = (size_t)sp * nk_len * 64;
    ra = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + t0 * p.lda + m0), 0, 0x7ffffff0, 0x00020000);
    rb = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + t0 * p.ldw + n0), 0, 0x7ffffff0, 0x00020000);
  };

  // ---- fragments through the transpose read: lane (i = lane&15, q = lane>>4) addresses token row q*8 + (i>>2) (+ h*32 + r*4),
  // columns (i&3)*4..+3 of the fragment's unit, and receives column i for four consecutive tokens ----
  const int fi = lane & 15, fq = lane >> 4;
  const int fg = (fi >> 2) | ((fq & 1) << 2);                                 // g(t) of every row this lane addresses
  const int fbase = (fq * 8 + (fi >> 2)) * 512 + (fi & 3) * 8;
  const int fx = fg << 5;                                                      // unit u of these rows sits at byte (u << 5) ^ fx of the row
  const int ua0 = (wave_m * 8) << 5, ub0 = (wave_n * 4) << 5;                  // (wave-uniform) first unit of this wave's rows / columns
  // The reads are inline assembly with hand-placed waits: hipcc puts an `s_waitcnt vmcnt(0)` in front of the transpose-read
  // BUILTIN whenever an LDS-DMA is in flight (it cannot tell that the DMA writes the other stage) - that wait sat at the top
  // of every k-step, serialising the DMA round trip with the MFMAs (1 047 vs 1 217 TF/s for the NT kernel on the same problem).
  // Protocol: the reads of phase p+1 are issued at the start of phase p; phase p+1 begins with `s_waitcnt lgkmcnt(0)` tied
  // ("+v") to the registers it is about to consume, so that no consumer (and no compiler-made copy) can move above the wait.
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;
  s16x4 afr[2][4][2], wfr[2][4][2];                       // [buffer][fragment][tokens 0-3 | 4-7 of the lane's 8-token chunk]
  auto tr_issue = [&](s16x4& lo, s16x4& hi, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:2048" : "=&v"(lo), "=&v"(hi) : "v"(addr));
  };
  auto ldA = [&](unsigned stage, int h, auto MH, int c) {          // stage = LDS byte offset of the operand stage
    constexpr int mh = decltype(MH)::value;
    int fxx = fx;
    asm volatile("" : "+v"(fxx));        // keeps the 12 (x 2 stages) fragment addresses from being hoisted into registers (they spilled)
#pragma unroll
    for (int ia = 0; ia < 4; ++ia)
      tr_issue(afr[c][ia][0], afr[c][ia][1], lds0 + stage + h * 16384 + fbase + ((ua0 + ((mh * 4 + ia) << 5)) ^ fxx));
  };
  auto ldB = [&](unsigned stage, int h, int c, auto J0, auto J1) {
    int fxx = fx;
    asm volatile("" : "+v"(fxx));
#pragma unroll
    for (int jb = decltype(J0)::value; jb < decltype(J1)::value; ++jb)
      tr_issue(wfr[c][jb][0], wfr[c][jb][1], lds0 + stage + TN_ABYTES + h * 16384 + fbase + ((ub0 + (jb << 5)) ^ fxx));
  };
  auto waitA = [&](int c) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(afr[c][0][0]), "+v"(afr[c][0][1]), "+v"(afr[c][1][0]), "+v"(afr[c][1][1]),
                 "+v"(afr[c][2][0]), "+v"(afr[c][2][1]), "+v"(afr[c][3][0]), "+v"(afr[c][3][1]));
  };
  auto waitB = [&](int c) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wfr[c][0][0]), "+v"(wfr[c][0][1]), "+v"(wfr[c][1][0]), "+v"(wfr[c][1][1]),
                 "+v"(wfr[c][2][0]), "+v"(wfr[c][2][1]), "+v"(wfr[c][3][0]), "+v"(wfr[c][3][1]));
  };
  auto frag = [&](const s16x4 (&f)[2]) { return __builtin_bit_cast(bf16x8, Frag2{f[0], f[1]}); };
  f32x4 acc[8][4];
  auto mma = [&](auto MH, int ca, int cw) {
    constexpr int mh = decltype(MH)::value;
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
      for (int ia = 0; ia < 4; ++ia)
        acc[mh * 4 + ia][jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag(wfr[cw][jb]), frag(afr[ca][ia]), acc[mh * 4 + ia][jb], 0, 0, 0);
  };
  auto first_frags = [&](unsigned stage) { ldA(stage, 0, IC<0>{}, 0); ldB(stage, 0, 0, IC<0>{}, IC<4>{}); };
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  int cur_m0, cur_n0, cur_sp;
  tile_origin(0, cur_m0, cur_n0, cur_sp);
  make_rsrc(cur_m0, cur_n0, cur_sp, rsA, rsB);
  __amdgpu_buffer_rsrc_t rsA_n = rsA, rsB_n = rsB;
  int nk = steps_of(cur_sp);                     // steps of the work item being computed
  int nk_dma = nk, nk_dma_n = nk;                // ... of the work item the DMA is in / of the one after it
  int dti = 0, dkt = 0;
  auto dma_step = [&](unsigned char* stage) {
    const int ka = dkt * a_kstep, kb = dkt * b_kstep;
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(stage + (i * NW + wid) * 1024), 16, voffA, ka + i * a_unit, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(stage + TN_ABYTES + (i * NW + wid) * 1024), 16, voffB, kb + i * b_unit, 0, 0);
    }
    ++dkt;
    if (dkt == nk_dma) { dkt = 0; ++dti; rsA = rsA_n; rsB = rsB_n; nk_dma = nk_dma_n; }
  };
  auto dma_wait_and_barrier = [&]() {          // see vl_gemm_park.hip: hipcc does not wait for the builtin's LDS writes
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };

  zero_acc();
  dma_step(smem);
  dma_step(smem + TN_STAGE);                   // nk >= 4 (vl_gemm_tn_supported): still inside work item 0
  dma_wait_and_barrier();
  first_frags(0u);

  int par = 0;
  bool pendB = false;
  auto kstep = [&](auto LAST) {
    constexpr bool last = decltype(LAST)::value;
    const unsigned cur = par * TN_STAGE, oth = (par ^ 1) * TN_STAGE;
    if (grpB && pendB) { dma_step(smem + oth); pendB = false; }
    __builtin_amdgcn_sched_barrier(0);
    waitA(0); waitB(0);                                   // first fragments of this stage (issued behind the previous barrier)
    ldA(cur, 0, IC<1>{}, 1); ldB(cur, 1, 1, IC<0>{}, IC<2>{});
    mma(IC<0>{}, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    waitA(1);
    ldA(cur, 1, IC<0>{}, 0); ldB(cur, 1, 1, IC<2>{}, IC<4>{});
    mma(IC<1>{}, 1, 0);
    __builtin_amdgcn_sched_barrier(0);
    waitA(0); waitB(1);
    ldA(cur, 1, IC<1>{}, 1);
    mma(IC<0>{}, 0, 1);
    __builtin_amdgcn_sched_barrier(0);
    waitA(1);
    dma_wait_and_barrier();
    if (dti < my_tiles) { if (!grpB) dma_step(smem + cur); else pendB = true; }
    if constexpr (!last) first_frags(oth);
    __builtin_amdgcn_sched_barrier(0);
    mma(IC<1>{}, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    par ^= 1;
  };

  const int prow = lane >> 3;
  for (int ti = 0; ti < my_tiles; ++ti) {
    if (ti + 1 < my_tiles) {
      int nm0, nn0, nsp;
      tile_origin(ti + 1, nm0, nn0, nsp);
      make_rsrc(nm0, nn0, nsp, rsA_n, rsB_n);
      nk_dma_n = steps_of(nsp);
    }
    for (int kt = 0; kt < nk - 1; ++kt) kstep(std::false_type{});
    kstep(std::true_type{});
    {
      // fp32 partial product of the slice: 32x32 blocks through the wave's slab, 16-byte non-temporal stores (8 lanes per line)
      const GemmP pe = reload_params();
      mfma_results_settled();
      const int mrow0 = cur_m0 + wave_m * 128, ncol0 = cur_n0 + wave_n * WTN;
      unsigned char* const slab = smem + 2 * TN_STAGE + wid * SLAB;
      float* const fout = (float*)pe.out + (size_t)cur_sp * pe.split_stride + (size_t)mrow0 * pe.ldo + ncol0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
          for (int ibh = 0; ibh < 2; ++ibh)
#pragma unroll
            for (int jbh = 0; jbh < 2; ++jbh) {
              const int row = ibh * 16 + fi;
              *(f32x4*)(slab + row * 128 + (((jbh * 4 + fq) ^ (row & 7)) << 4)) =
                  scale_bias(acc[i * 2 + ibh][j * 2 + jbh], pe.alpha, f32x4{0.f, 0.f, 0.f, 0.f});
            }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int pass = 0; pass < 4; ++pass) {
            const int r = pass * 8 + prow;
            const f32x4 w = *(const f32x4*)(slab + r * 128 + (((lane & 7) ^ (r & 7)) << 4));
            __builtin_nontemporal_store(w, (f32x4*)(fout + (size_t)(i * 32 + r) * pe.ldo + j * 32 + (lane & 7) * 4));
          }
        }
      }
      zero_acc();
      if (ti + 1 < my_tiles) { tile_origin(ti + 1, cur_m0, cur_n0, cur_sp); nk = steps_of(cur_sp); first_frags((unsigned)(par * TN_STAGE)); }
    }
  }
}

}  // namespace

// Internal entry used by vl_gemm.hip (vl_gemm_tn_splitk_accum_f32): p.A = At [K, M] (row stride lda), p.W = Bt [K, N] (row stride
// ldw), p.out = workspace of (K/64/ksplit_len) planes of M x N floats (row stride ldo = N), plane stride split_stride.
bool vl_gemm_tn_supported(const void* params) {
  const GemmP& p = *(const GemmP*)params;
  if ((p.M & 255) || (p.N & 255) || (p.K & 63) || p.M <= 0 || p.N <= 0 || p.K <= 0) return false;
  if ((p.lda & 7) || (p.ldw & 7) || (p.ldo & 3)) return false;
  const int nk = p.K >> 6;
  if (p.ksplit_len < 4 || p.ksplit_len > nk) return false;
  const int last = nk - ((nk + p.ksplit_len - 1) / p.ksplit_len - 1) * p.ksplit_len;
  if (last < 4) return false;                                      // the DMA runs two k-steps ahead: it must not leave a work item in its prologue
  if ((long)p.ksplit_len * 128 * (p.lda > p.ldw ? p.lda : p.ldw) >= (1L << 31)) return false;      // scalar byte offsets of a slice
  return !((((uintptr_t)p.A | (uintptr_t)p.W | (uintptr_t)p.out) & 15));
}

int vl_gemm_tn_launch(const void* params, int ncu, hipStream_t s) {
  const GemmP& p = *(const GemmP*)params;
  auto kern = gemm_tn_kernel;
  static const hipError_t attr = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS);   // thread-safe one-time init
  if (attr != hipSuccess) return (int)attr;
  const int tiles = (p.M >> 8) * (p.N >> 8) * (((p.K >> 6) + p.ksplit_len - 1) / p.ksplit_len);
  int G = ncu & ~7;
  if (tiles < G) G = (tiles + 7) & ~7;
  hipLaunchKernelGGL(kern, dim3(G), dim3(512), TN_LDS, s, p);
  return (int)hipGetLastError();
}
