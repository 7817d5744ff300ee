// Ping-pong persistent bf16 NT GEMM for gfx950 (round-3 kernel): C[M,N] = A[M,K] * W[N,K]^T (+ fused epilogue).
// TWO independent 4-wave workgroups per CU (one wave of each per SIMD), each on its own 256 x 128 output tiles with its
// own 72 KB LDS ring and its own barriers, started half an epilogue apart: while one workgroup is in its epilogue
// (bias / GELU arithmetic on the VALU, LDS transpose, the store burst) the other one's main loop owns the matrix pipe.
//
// Why (DESIGN.md section 7, round 2 -> 3): the round-2 kernel (vl_gemm_park.hip: one 8-wave workgroup, 256 x 256 tiles) runs
// its main loop at 1300-1350 TF/s but the K = 1024 family at 920-1130, because all eight waves reach the epilogue
// together and the MFMAs idle for ~11 us of every 37 us tile (c_fc + GELU + saved pre-activation); everything tried INSIDE one
// workgroup lost (trickled stores, prefetched residual, 4 waves x 512 registers).  Two workgroups need no choreography: the
// hardware interleaves two instruction streams per SIMD, and a DMA-latency stall of one workgroup is the other's gain.
//
// Geometry: tile 256 (M) x 128 (N), 4 waves as 2 x 2 of 128 x 64 (the same per-wave tile, fragment reads and epilogue as
// round 2), k-step = 32, LDS ring of THREE 24 KB stages (A 256 rows x 64 B + W 128 rows x 64 B) = 72 KB per workgroup.
// 64-byte LDS rows: 16-byte chunk c of row r sits at slot c ^ ((r >> 2) & 3) -> the 16 lanes of a ds_read_b128 group hit
// 16 distinct (r & 3, slot) pairs = all 64 banks once.  The LDS-DMA writes lane-linear, so the swizzle is applied to the
// global source address (a 64-byte row segment is 4 lanes, permuted inside the segment: coalescing is unchanged).
//
// Pipeline per 32-deep step s (stage s % 3):  top: DMA of step s+2 into the stage step s-1 used (every wave finished reading
// it before the previous barrier) | fragment reads of the second 16-deep half | 8 MFMAs | s_waitcnt vmcnt(6) lgkmcnt(0) +
// s_barrier (own batch of step s+1 has landed - the 6 newest DMA instructions, step s+2, stay in flight - and everybody's
// has after the barrier) | fragment reads of step s+1's first half | 8 MFMAs.  The DMA position runs two steps ahead of
// the MFMAs across tile boundaries.
//
// Epilogue: as round 2 (arithmetic in the accumulator layout -> bf16 -> wave-private 4 KB LDS slab -> 16-byte non-temporal
// stores, 8 lanes per 128-byte line).  The slab needs no LDS of its own: it is the four 1 KB pieces of the just-consumed
// stage that THIS wave's next DMA batch overwrites (program order inside one wave), so no other wave ever touches it.
//
// Replaces: nn.Linear / MultiheadAttention in/out projections of ResidualAttentionBlock
// (open_clip/transformer.py:215,226-234,252-272) in forward and dX-backward at ViT-L sizes, and the split-K partial
// products of the weight gradients.
#include <type_traits>

#include "vl_gemm_common.h"

namespace {

constexpr int PP_ABYTES = 16384, PP_WBYTES = 8192, PP_STAGE = PP_ABYTES + PP_WBYTES, PP_NST = 3;
constexpr int PP_LDS = PP_NST * PP_STAGE;               // 72 KB: two workgroups per CU
constexpr int PP_GN = 8;                                 // N-tiles (128 columns) per group of the tile order

template <int I>
using IC = std::integral_constant<int, I>;

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// ACT (EPI_BF16 only): 0 none, 1 GELU, 2 ReLU, 3 GELU with the pre-activation also written to out2 (saved for backward)
template <int EPI, int ACT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
    gemm_nt_pp_kernel(const GemmP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr bool HAS_AUX = (EPI == EPI_RES_BF16 || EPI == EPI_DGELU);   // second operand of the output's shape
  constexpr int NW = 4, NTL = 2, WTN = 64;   // 2 (M) x 2 (N) waves of 128 x 64
  constexpr int NCH = 16;                    // output chunks per wave: chunk = 8 rows x 128 bytes = one store instruction

  const int tiles_n = p.N >> 7, tiles_m = p.M >> 8;
  // split-K (EPI_F32 only): work item = (k-slice, tile); ksplit_len counts 64-deep units (the host API's k-steps)
  const bool splitk = (EPI == EPI_F32 && p.ksplit_len);
  const int nk = splitk ? p.ksplit_len * 2 : (p.K >> 5);          // 32-deep steps per work item (>= 4)
  const int ntiles_mn = tiles_m * tiles_n;
  const int ntiles = ntiles_mn * (splitk ? (p.K >> 6) / p.ksplit_len : 1);
  const int G = gridDim.x;
  const int slot = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  if (slot >= ntiles) return;
  const int my_tiles = (ntiles - slot + G - 1) / G;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wid & 1, wave_n = wid >> 1;
  const int fr = lane & 31, fg = lane >> 5;
  const int fsw = (fr >> 2) & 3;

  // Start-up phase shift: the workgroup whose waves sit in the odd wave slot of their SIMD (the second one dispatched to
  // this CU) waits ~one epilogue before its first tile, so the two workgroups' epilogues never coincide.  Speed only.
  {
    constexpr int PP_DELAY = 4;      // x 4096 cycles (0 / 2 / 4 / 8 measured: 0.695 / 0.684 / 0.677 / 0.705 ms on the c_fc shape, round 3)
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 0, 4)" : "=s"(hw));
    if (hw & 1)
      for (int i = 0; i < PP_DELAY; ++i) __builtin_amdgcn_s_sleep(64);      // 64 x 64 cycles
  }

  // tile order: groups of PP_GN consecutive N-tiles, each XCD owns a contiguous run per round
  auto tile_origin = [&](int ti, int& m0, int& n0, int& sp) {
    int v = ti * G + slot;
    sp = 0;
    if constexpr (EPI == EPI_F32) { sp = v / ntiles_mn; v -= sp * ntiles_mn; }
    const int gsz = PP_GN * tiles_m;
    const int gid = v / gsz, rem = v - gid * gsz;
    const int first_n = gid * PP_GN;
    const int gn = min(tiles_n - first_n, PP_GN);
    const int tm = rem / gn;
    m0 = tm << 8; n0 = (first_n + (rem - tm * gn)) << 7;
  };

  // ---- LDS-DMA: unit u of an operand = 16 rows x 64 bytes = 1 KB; wave wid loads units i*4 + wid ----
  // lane -> row (lane >> 2) of the unit, physical 16-byte slot lane & 3, which holds logical chunk (lane & 3) ^ ((row >> 2) & 3)
  const int drow = wid * 16 + (lane >> 2);
  const int dch = ((lane & 3) ^ ((lane >> 4) & 3)) * 16;
  const unsigned voffA = (unsigned)(drow * p.lda * 2 + dch), voffW = (unsigned)(drow * p.ldw * 2 + dch);
  const int a_unit = p.lda * 128, w_unit = p.ldw * 128;            // bytes between a wave's units (64 rows)
  __amdgpu_buffer_rsrc_t rsA, rsW;
  auto make_rsrc = [&](int m0, int n0, int sp, __amdgpu_buffer_rsrc_t& ra, __amdgpu_buffer_rsrc_t& rw) {
    const size_t k0 = (size_t)sp * nk * 32;        // first reduction index of the slice
    ra = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (size_t)m0 * p.lda + k0), 0, 0x7ffffff0, 0x00020000);
    rw = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (size_t)n0 * p.ldw + k0), 0, 0x7ffffff0, 0x00020000);
  };

  // ---- fragments (two sets: substep parity) ----
  const int fa_base = (wave_m * 128 + fr) * 64, fw_base = PP_ABYTES + (wave_n * WTN + fr) * 64;
  bf16x8 af[2][4], wf[2][NTL];
  auto ldfrag = [&](const unsigned char* stage, int kk, int c) {
    const int off = ((kk * 2 + fg) ^ fsw) * 16;
#pragma unroll
    for (int j = 0; j < NTL; ++j) wf[c][j] = *(const bf16x8*)(stage + fw_base + j * 2048 + off);
#pragma unroll
    for (int i = 0; i < 4; ++i) af[c][i] = *(const bf16x8*)(stage + fa_base + i * 2048 + off);
  };
  f32x16 acc[4][NTL];
  auto mma = [&](int c) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NTL; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[c][j], af[c][i], acc[i][j], 0, 0, 0);
  };
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NTL; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };

  // ---- epilogue operands ----
  // chunk ci (0..15) = rows (ci>>2)*32 + (ci&3)*8 + (lane>>3) of the wave's sub-tile, columns (lane&7)*8 .. +7
  [[maybe_unused]] u32x4 aux[NCH];
  const int prow = lane >> 3, pcol = (lane & 7) * 8;
  const unsigned lo_out = (unsigned)((prow * p.ldo + pcol) * 2);     // lane part of an output / aux address (row-major outputs)
  const int ldo2 = p.ldo * 2;
  auto chunk_row = [](int ci) { return (ci >> 2) * 32 + (ci & 3) * 8; };
  [[maybe_unused]] const unsigned char* aux_src = nullptr;
  auto load_pair = [&](auto GI) {
    constexpr int g = decltype(GI)::value;
    if constexpr (HAS_AUX) {
      aux[g * 2] = *(const u32x4*)(aux_src + (size_t)(chunk_row(g * 2) * ldo2) + lo_out);
      aux[g * 2 + 1] = *(const u32x4*)(aux_src + (size_t)(chunk_row(g * 2 + 1) * ldo2) + lo_out);
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
  __amdgpu_buffer_rsrc_t rsA_n = rsA, rsW_n = rsW;
  int dti = 0, dkt = 0;                          // DMA position: (tile, step) the next batch loads
  int rdma = 0;                                  // ring stage the next batch fills
  auto dma_step = [&]() {
    unsigned char* stage = smem + rdma * PP_STAGE;
    const int kbyte = dkt << 6;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(stage + (i * NW + wid) * 1024), 16, voffA, kbyte + i * a_unit, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr_t)(stage + PP_ABYTES + (i * NW + wid) * 1024), 16, voffW, kbyte + i * w_unit, 0, 0);
    rdma = rdma == PP_NST - 1 ? 0 : rdma + 1;
    ++dkt;
    if (dkt == nk) { dkt = 0; ++dti; rsA = rsA_n; rsW = rsW_n; }
  };

  zero_acc();
  dma_step();
  dma_step();                                    // nk >= 4: still inside tile 0
  // hipcc does not wait for this builtin's LDS writes in front of a barrier: wait by hand, raw barrier (see vl_gemm_park.hip)
  asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
  ldfrag(smem, 0, 0);

  int rcur = 0;                                  // ring stage of the step being computed
  auto kstep = [&](auto LAST) {
    constexpr bool last = decltype(LAST)::value;
    unsigned char* cur = smem + rcur * PP_STAGE;
    const bool more = dti < my_tiles;
    if (more) dma_step();                        // step s+2 into the stage of step s-1
    __builtin_amdgcn_sched_barrier(0);
    ldfrag(cur, 1, 1);
    mma(0);
    __builtin_amdgcn_sched_barrier(0);
    // every wave holds its last fragments of `cur`; the batch of step s+1 (issued one step ago) must have landed
    if (more) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    rcur = rcur == PP_NST - 1 ? 0 : rcur + 1;
    if constexpr (!last) ldfrag(smem + rcur * PP_STAGE, 0, 0);      // (at a tile boundary the fragments would sit in registers through the epilogue)
    __builtin_amdgcn_sched_barrier(0);
    mma(1);
    __builtin_amdgcn_sched_barrier(0);
  };

  for (int ti = 0; ti < my_tiles; ++ti) {
    if (ti + 1 < my_tiles) {
      int nm0, nn0, nsp;
      tile_origin(ti + 1, nm0, nn0, nsp);
      make_rsrc(nm0, nn0, nsp, rsA_n, rsW_n);
    }
    set_aux();
    for (int kt = 0; kt < nk - 1; ++kt) kstep(std::false_type{});
    kstep(std::true_type{});
    {
      // ---------------- tile finished: arithmetic, LDS transpose, one burst of 16-byte non-temporal stores ----------------
      const GemmP pe = reload_params();
      mfma_results_settled();
      const int mrow0 = cur_m0 + wave_m * 128, ncol0 = cur_n0 + wave_n * WTN;
      if constexpr (HAS_AUX) {
        load_pair(IC<0>{}); load_pair(IC<1>{}); load_pair(IC<2>{}); load_pair(IC<3>{});
        load_pair(IC<4>{}); load_pair(IC<5>{}); load_pair(IC<6>{}); load_pair(IC<7>{});
      }
      // slab = this wave's four 1 KB DMA pieces of the stage consumed last (= the stage its next DMA batch fills):
      // row r (0..31) at (r >> 3) * 4096 + (r & 7) * 128
      const int rfree = rcur == 0 ? PP_NST - 1 : rcur - 1;
      unsigned char* const slab = smem + rfree * PP_STAGE + wid * 1024;
      unsigned char* const wr = slab + (fr >> 3) * 4096 + (fr & 7) * 128 + fg * 8;
      const int wsw = fr & 7;
      // branch-free optional bias: read SOMETHING valid (the weight matrix) and select zero
      const bool has_bias = pe.bias != nullptr;
      const float* const bsrc = has_bias ? pe.bias : (const float*)pe.W;
      unsigned char* const out_base = (unsigned char*)pe.out + ((size_t)mrow0 * pe.ldo + ncol0) * 2 + lo_out;
      if constexpr (EPI == EPI_F32) {
        // fp32 partial product of a k-slice: 32x32 blocks through the slab, 16-byte stores (8 lanes per 128-byte line)
        float* const fout = (float*)pe.out + (size_t)cur_sp * pe.split_stride + (size_t)mrow0 * pe.ldo + ncol0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int j = 0; j < NTL; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              f32x4 v = {acc[i][j][q * 4 + 0], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]};
              *(f32x4*)(slab + (fr >> 3) * 4096 + (fr & 7) * 128 + (((q * 2 + fg) ^ wsw) << 4)) = scale_bias(v, pe.alpha, f32x4{0.f, 0.f, 0.f, 0.f});
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
              const f32x4 w = *(const f32x4*)(slab + pass * 4096 + prow * 128 + (((lane & 7) ^ prow) << 4));
              __builtin_nontemporal_store(w, (f32x4*)(fout + (size_t)(i * 32 + pass * 8 + prow) * pe.ldo + j * 32 + (lane & 7) * 4));
            }
          }
        }
      } else
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int n = ncol0 + j * 32 + q * 8 + fg * 4;
            f32x4 v = {acc[i][j][q * 4 + 0], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]};
            f32x4 bv = *(const f32x4*)(bsrc + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[e] = has_bias ? bv[e] : 0.f;
            v = scale_bias(v, pe.alpha, bv);
            if constexpr (EPI == EPI_BF16 && ACT == 1) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
            } else if constexpr (EPI == EPI_BF16 && ACT == 2) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            u32x2 o; o[0] = pack2bf(v[0], v[1]); o[1] = pack2bf(v[2], v[3]);
            *(u32x2*)(wr + (((j * 4 + q) ^ wsw) << 4)) = o;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
          u32x4 w = *(const u32x4*)(slab + pass * 4096 + prow * 128 + (((lane & 7) ^ prow) << 4));
          [[maybe_unused]] u32x4 rr;
          if constexpr (HAS_AUX) rr = aux[i * 4 + pass];
          if constexpr (EPI == EPI_BF16 && ACT == 3) {
            __builtin_nontemporal_store(w, (u32x4*)((unsigned char*)pe.out2 + ((size_t)(mrow0 + i * 32 + pass * 8) * pe.ldo + ncol0) * 2 + lo_out));
#pragma unroll
            for (int e = 0; e < 4; ++e)
              w[e] = pack2bf(gelu_erf(bf2f((bf16_t)(w[e] & 0xffff))), gelu_erf(bf2f((bf16_t)(w[e] >> 16))));
          } else if constexpr (EPI == EPI_BF16 && ACT == 4) {
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
          } else if constexpr (EPI == EPI_DGELU && ACT == 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = mul_pk_bf16(w[e], rr[e]);     // aux = gelu' saved by the forward
          } else if constexpr (EPI == EPI_DGELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              w[e] = pack2bf(bf2f((bf16_t)(w[e] & 0xffff)) * gelu_erf_grad(bf2f((bf16_t)(rr[e] & 0xffff))),
                             bf2f((bf16_t)(w[e] >> 16)) * gelu_erf_grad(bf2f((bf16_t)(rr[e] >> 16))));
          }
          __builtin_nontemporal_store(w, (u32x4*)(out_base + (size_t)((i * 32 + pass * 8) * ldo2)));
        }
      }
      zero_acc();
      // all slab reads have returned (their data fed the stores above) before this wave's next DMA batch reuses the pieces
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (ti + 1 < my_tiles) { tile_origin(ti + 1, cur_m0, cur_n0, cur_sp); ldfrag(smem + rcur * PP_STAGE, 0, 0); }
    }
  }
}

template <int EPI, int ACT>
hipError_t launch_pp(const GemmP& p, int ncu, hipStream_t s) {
  auto kern = gemm_nt_pp_kernel<EPI, ACT>;
  static const hipError_t attr = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS);   // thread-safe one-time init
  if (attr != hipSuccess) return attr;
  const bool splitk = (EPI == EPI_F32 && p.ksplit_len);
  const int tiles = (p.M >> 8) * (p.N >> 7) * (splitk ? (p.K >> 6) / p.ksplit_len : 1);
  int G = (2 * ncu) & ~7;                        // two workgroups per CU
  if (tiles < G) G = (tiles + 7) & ~7;
  hipLaunchKernelGGL(kern, dim3(G), dim3(256), PP_LDS, s, p);
  return hipGetLastError();
}

}  // namespace

// Internal entries used by vl_gemm.hip's dispatcher (not part of the public C ABI).
bool vl_gemm_pp_supported(int epi, const void* params) {
  const GemmP& p = *(const GemmP*)params;
  if ((p.M & 255) || (p.N & 127) || (p.K & 63) || p.M <= 0 || p.N <= 0) return false;
  if (epi == EPI_F32) {       // split-K partial products into a workspace (vl_gemm_splitk_accum_f32)
    if ((p.ldo & 3) || p.bias) return false;
    const int nk = p.K >> 6;
    if (p.ksplit_len < 2 || nk % p.ksplit_len) return false;
    return !((((uintptr_t)p.A | (uintptr_t)p.W | (uintptr_t)p.out) & 15));
  }
  if (!(epi == EPI_BF16 || epi == EPI_RES_BF16 || epi == EPI_DGELU)) return false;
  if (p.K < 128) return false;                  // the DMA prologue issues two steps of tile 0 up front
  if (p.res_div != 1) return false;
  if (epi == EPI_RES_BF16 && p.act != 0) return false;
  if (epi == EPI_BF16 && p.out2 && p.act != 1 && p.act != 4) return false;
  if (epi == EPI_BF16 && p.act == 4 && !p.out2) return false;
  if (p.ldo & 7) return false;
  if (((uintptr_t)p.A | (uintptr_t)p.W) & 15) return false;
  if (((uintptr_t)p.out | (uintptr_t)p.res | (uintptr_t)p.out2) & 15) return false;
  return true;
}

int vl_gemm_pp_launch(int epi, const void* params, int ncu, hipStream_t s) {
  const GemmP& p = *(const GemmP*)params;
  switch (epi) {
    case EPI_BF16:
      if (p.act == 1) return p.out2 ? (int)launch_pp<EPI_BF16, 3>(p, ncu, s) : (int)launch_pp<EPI_BF16, 1>(p, ncu, s);
      if (p.act == 4) return (int)launch_pp<EPI_BF16, 4>(p, ncu, s);
      return p.act == 2 ? (int)launch_pp<EPI_BF16, 2>(p, ncu, s) : (int)launch_pp<EPI_BF16, 0>(p, ncu, s);
    case EPI_RES_BF16: return (int)launch_pp<EPI_RES_BF16, 0>(p, ncu, s);
    case EPI_DGELU: return p.act == 4 ? (int)launch_pp<EPI_DGELU, 4>(p, ncu, s) : (int)launch_pp<EPI_DGELU, 0>(p, ncu, s);
    case EPI_F32: return (int)launch_pp<EPI_F32, 0>(p, ncu, s);
    default: return (int)hipErrorInvalidValue;
  }
}
