// bf16 NT GEMM for gfx950 (MI355X):  C[M,N] = A[M,K] * W[N,K]^T  (+ fused epilogues)
//
// Replaces the nn.Linear / nn.MultiheadAttention in/out projections / conv1-as-GEMM calls of
// the reference hot path (open_clip/transformer.py:215,226-234,252,464-470; perceiver.py:85-123;
// loss.py:131-133).  Design (MI355X-first, not a translation of anything):
//   * 64-wide waves, v_mfma_f32_32x32x16_bf16, fp32 accumulation in the unified VGPR/AGPR file.
//   * operands are "swapped": the weight tile feeds the MFMA row operand and the activation tile
//     the column operand, so each lane ends up owning 4 CONSECUTIVE output columns of one output
//     row -> 8-byte bf16 / 16-byte fp32 epilogue stores and float4 residual reads.
//   * BK = 64 K-tiles (128-byte rows) double-buffered in LDS; rows are XOR-swizzled at 16-byte
//     granularity (chunk ^= (row>>1)&7) so every ds_read_b128 lane group hits 16 distinct slots.
//   * staging is either LDS-DMA (global_load_lds_dwordx4, swizzle applied on the per-lane SOURCE
//     address, LDS image lane-linear) or plain register staging; selected by template flag.
//   * XCD-aware bijective block remap keeps consecutive N-tiles of one M-tile on one XCD's L2.
#include "vl_gemm_common.h"

namespace {

// ---- epilogue: lane owns row m = mrow0 + 32*i + fr, columns n = ncol0 + 32*j + 8*q + 4*fg + {0..3} ----
template <int EPI, int MT, int NTL>
__device__ __forceinline__ void store_tile(const GemmP& p, f32x16 (&acc)[MT][NTL], int mrow0, int ncol0, int fr, int fg) {
  // rows OUTER: a lane writes the 8 column groups of one row back to back, so the 128-byte lines of that row
  // are completed while still in the write-combining window (columns-outer order cost the fc GEMM 40 %).
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = mrow0 + i * 32 + fr;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < NTL; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = ncol0 + j * 32 + q * 8 + fg * 4;
        if (n >= p.N) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][q * 4 + e] * p.alpha;
        if (p.bias) {
          const f32x4 bv = *(const f32x4*)(p.bias + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += bv[e];
        }
        if constexpr (EPI == EPI_BF16) {
          if (p.act == 4) {          // GELU of the bf16-rounded pre-activation, out2 = gelu' of the same (bf16)
            unsigned int y0, g0, y1, g1;
            gelu_and_grad_pk(pack2bf(v[0], v[1]), y0, g0);
            gelu_and_grad_pk(pack2bf(v[2], v[3]), y1, g1);
            const u32x2 o = {y0, y1}, d = {g0, g1};
            *(u32x2*)((bf16_t*)p.out2 + (size_t)m * p.ldo + n) = d;
            *(u32x2*)((bf16_t*)p.out + (size_t)m * p.ldo + n) = o;
            continue;
          }
          if (p.act == 1) {
            if (p.out2) {
              u32x2 o2; o2[0] = pack2bf(v[0], v[1]); o2[1] = pack2bf(v[2], v[3]);
              *(u32x2*)((bf16_t*)p.out2 + (size_t)m * p.ldo + n) = o2;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
          } else if (p.act == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          u32x2 o; o[0] = pack2bf(v[0], v[1]); o[1] = pack2bf(v[2], v[3]);
          *(u32x2*)((bf16_t*)p.out + (size_t)m * p.ldo + n) = o;
        } else if constexpr (EPI == EPI_F32) {
          f32x4 o = {v[0], v[1], v[2], v[3]};
          *(f32x4*)((float*)p.out + (size_t)m * p.ldo + n) = o;
        } else if constexpr (EPI == EPI_RES_F32) {
          const f32x4 r = *(const f32x4*)((const float*)p.res + (size_t)m * p.ldo + n);
          f32x4 o = {v[0] + r[0], v[1] + r[1], v[2] + r[2], v[3] + r[3]};
          *(f32x4*)((float*)p.out + (size_t)m * p.ldo + n) = o;
        } else if constexpr (EPI == EPI_RES_BF16) {
          const u32x2 r = *(const u32x2*)((const bf16_t*)p.res + (size_t)((m + p.m_off) / p.res_div) * p.ldo + n);
          v[0] += bf2f((bf16_t)(r[0] & 0xffff)); v[1] += bf2f((bf16_t)(r[0] >> 16));
          v[2] += bf2f((bf16_t)(r[1] & 0xffff)); v[3] += bf2f((bf16_t)(r[1] >> 16));
          if (p.act == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          u32x2 o; o[0] = pack2bf(v[0], v[1]); o[1] = pack2bf(v[2], v[3]);
          *(u32x2*)((bf16_t*)p.out + (size_t)m * p.ldo + n) = o;
        } else if constexpr (EPI == EPI_DGELU) {
          const u32x2 r = *(const u32x2*)((const bf16_t*)p.res + (size_t)m * p.ldo + n);
          const bool saved = p.act == 4;          // res = gelu' itself (left by a VL_ACT_GELU_DSAVE forward)
          const float r0 = bf2f((bf16_t)(r[0] & 0xffff)), r1 = bf2f((bf16_t)(r[0] >> 16));
          const float r2 = bf2f((bf16_t)(r[1] & 0xffff)), r3 = bf2f((bf16_t)(r[1] >> 16));
          v[0] *= saved ? r0 : gelu_erf_grad(r0); v[1] *= saved ? r1 : gelu_erf_grad(r1);
          v[2] *= saved ? r2 : gelu_erf_grad(r2); v[3] *= saved ? r3 : gelu_erf_grad(r3);
          u32x2 o; o[0] = pack2bf(v[0], v[1]); o[1] = pack2bf(v[2], v[3]);
          *(u32x2*)((bf16_t*)p.out + (size_t)m * p.ldo + n) = o;
        } else if constexpr (EPI == EPI_GEGLU) {
          // interleaved rows: (a_j, gate_j, a_j+1, gate_j+1)
          if (p.out2) {   // pre-activation kept for the backward: bf16 [M, N], row stride 2*ldo
            u32x2 o2; o2[0] = pack2bf(v[0], v[1]); o2[1] = pack2bf(v[2], v[3]);
            *(u32x2*)((bf16_t*)p.out2 + (size_t)m * (2 * p.ldo) + n) = o2;
          }
          const float o0 = v[0] * gelu_erf(v[1]);
          const float o1 = v[2] * gelu_erf(v[3]);
          *(unsigned int*)((bf16_t*)p.out + (size_t)m * p.ldo + (n >> 1)) = pack2bf(o0, o1);
        } else if constexpr (EPI == EPI_DGEGLU) {
          const u32x4 hv = *(const u32x4*)((const bf16_t*)p.res + (size_t)m * p.ldo + 2 * n);
          u32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float a = bf2f((bf16_t)(hv[e] & 0xffff)), g = bf2f((bf16_t)(hv[e] >> 16));
            o[e] = pack2bf(v[e] * gelu_erf(g), v[e] * a * gelu_erf_grad(g));
          }
          *(u32x4*)((bf16_t*)p.out + (size_t)m * p.ldo + 2 * n) = o;
        }
      }
    }
  }
}

// bf16-output epilogues through the LDS transpose with 16-byte global accesses.  A wave-store instruction costs the
// store path ~73 clk whatever its width (measured: 128 KB of dwordx2 stores per tile drain at ~7 B/clk/CU), so the
// number of store instructions is what the epilogue pays for.  The values are therefore packed to bf16 BEFORE the
// transpose (slab = 32 rows x 64 columns bf16 = the same 4 KB) and come back as 8 consecutive columns per lane:
// 8 lanes write one whole 128-byte line with a single dwordx4 each - half the store instructions of the fp32
// transpose.  Arithmetic that needs another tensor of the output's shape (bf16 residual, saved pre-activation for
// dGELU, GELU when the pre-activation is also kept) runs after the read-back on the bf16-rounded values, with
// coalesced 16-byte loads - the same rounding points as the reference's autocast (bf16 linear output, then the add /
// activation); bias, alpha, the q scale and GELU-without-save are applied in fp32 before packing.
template <int EPI, int MT, int NTL>
__device__ __forceinline__ void store_tile_lds16(const GemmP& p, f32x16 (&acc)[MT][NTL], int j0, int mrow0, int ncol0, int fr, int fg,
                                                 int lane, unsigned char* wl) {
  // handles the 64 columns of column blocks j0, j0+1 (ncol0 = first column of block j0)
  const int rsub = lane >> 3, c = lane & 7;
  unsigned char* const wr = wl + fr * 128 + fg * 8;
  const int wsw = (fr >> 1) & 7;
  const int n8 = ncol0 + c * 8;
  const bool col_ok = n8 < p.N;
  const bool gelu_pre = (EPI == EPI_BF16) && p.act == 1 && !p.out2;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = ncol0 + j * 32 + q * 8 + fg * 4;
        f32x4 v = {acc[i][j0 + j][q * 4 + 0], acc[i][j0 + j][q * 4 + 1], acc[i][j0 + j][q * 4 + 2], acc[i][j0 + j][q * 4 + 3]};
        v = v * p.alpha;
        if (p.bias && n < p.N) v = v + *(const f32x4*)(p.bias + n);
        if constexpr (EPI == EPI_BF16) {
          if (gelu_pre) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
          } else if (p.act == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          }
        }
        u32x2 o; o[0] = pack2bf(v[0], v[1]); o[1] = pack2bf(v[2], v[3]);
        *(u32x2*)(wr + (((j * 4 + q) ^ wsw) << 4)) = o;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int mb = mrow0 + i * 32;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int r = pass * 8 + rsub;
      const int m = mb + r;
      u32x4 w = *(const u32x4*)(wl + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
      if (m >= p.M || !col_ok) continue;
      if constexpr (EPI == EPI_BF16) {
        if (p.act == 1 && p.out2) {
          *(u32x4*)((bf16_t*)p.out2 + (size_t)m * p.ldo + n8) = w;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            w[e] = pack2bf(gelu_erf(bf2f((bf16_t)(w[e] & 0xffff))), gelu_erf(bf2f((bf16_t)(w[e] >> 16))));
        } else if (p.act == 4) {
          u32x4 d;
#pragma unroll
          for (int e = 0; e < 4; ++e) { unsigned int y, g; gelu_and_grad_pk(w[e], y, g); w[e] = y; d[e] = g; }
          *(u32x4*)((bf16_t*)p.out2 + (size_t)m * p.ldo + n8) = d;
        }
        *(u32x4*)((bf16_t*)p.out + (size_t)m * p.ldo + n8) = w;
      } else if constexpr (EPI == EPI_RES_BF16) {
        const u32x4 rr = *(const u32x4*)((const bf16_t*)p.res + (size_t)((m + p.m_off) / p.res_div) * p.ldo + n8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float lo = bf2f((bf16_t)(w[e] & 0xffff)) + bf2f((bf16_t)(rr[e] & 0xffff));
          float hi = bf2f((bf16_t)(w[e] >> 16)) + bf2f((bf16_t)(rr[e] >> 16));
          if (p.act == 2) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }
          w[e] = pack2bf(lo, hi);
        }
        *(u32x4*)((bf16_t*)p.out + (size_t)m * p.ldo + n8) = w;
      } else if constexpr (EPI == EPI_DGELU) {
        const u32x4 rr = *(const u32x4*)((const bf16_t*)p.res + (size_t)m * p.ldo + n8);
        if (p.act == 4) {
#pragma unroll
          for (int e = 0; e < 4; ++e) w[e] = mul_pk_bf16(w[e], rr[e]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            w[e] = pack2bf(bf2f((bf16_t)(w[e] & 0xffff)) * gelu_erf_grad(bf2f((bf16_t)(rr[e] & 0xffff))),
                           bf2f((bf16_t)(w[e] >> 16)) * gelu_erf_grad(bf2f((bf16_t)(rr[e] >> 16))));
        }
        *(u32x4*)((bf16_t*)p.out + (size_t)m * p.ldo + n8) = w;
      }
    }
    asm volatile("" ::: "memory");
  }
}

// ------------------------------------------------------------------------------------------------
// Epilogue through a wave-private LDS transpose (persistent kernel).  In the accumulator layout a lane owns one
// output ROW, so a direct store instruction touches 32 different 128-byte lines with 16 bytes each: the store
// path then issues ~7 B/clk/CU and the epilogue of a 256x256 tile takes ~20k cycles - as long as two thirds of
// the K=1024 main loop (measured: 0.66 ms with, 0.45 ms without epilogue; folding the stores onto L2-resident
// rows changed little, so it is issue, not HBM).  Here each 32x32 fp32 block of the wave's sub-tile goes through
// a 4 KB LDS slab (XOR-swizzled 16-byte chunks, conflict-free both ways) and comes back with 8 consecutive lanes
// on one row: every global access of the epilogue - output, residual, saved pre-activation - is a run of whole
// 64/128-byte segments, and the bias is loaded once per column block instead of once per row.
// The transposed attention layouts (q^T, k^T, v^T: token index contiguous) are row-per-lane friendly and keep the
// direct path.  GEGLU epilogues (Perceiver feed-forward only) fall back to store_tile.
template <int EPI, int MT, int NTL>
__device__ __forceinline__ void store_tile_lds(const GemmP& p, f32x16 (&acc)[MT][NTL], int mrow0, int ncol0, int fr, int fg,
                                               int lane, unsigned char* wl) {
  if constexpr (EPI == EPI_GEGLU || EPI == EPI_DGEGLU) {
    store_tile<EPI, MT, NTL>(p, acc, mrow0, ncol0, fr, fg);
    return;
  } else {
    if constexpr ((EPI == EPI_BF16 || EPI == EPI_RES_BF16 || EPI == EPI_DGELU) && (NTL & 1) == 0) {
      // bf16 outputs: 16-byte path when rows stay 16-byte aligned
      if ((p.N & 7) == 0 && (p.ldo & 7) == 0) {
#pragma unroll
        for (int j0 = 0; j0 < NTL; j0 += 2)
          store_tile_lds16<EPI, MT, NTL>(p, acc, j0, mrow0, ncol0 + j0 * 32, fr, fg, lane, wl);
        return;
      }
    }
    const int rsub = lane >> 3, c = lane & 7;
    unsigned char* const wr = wl + fr * 128;
    const int wsw = (fr >> 1) & 7;
    // per-column-block constants (bias) first; then ROW blocks outermost so that the two 64-byte
    // halves of a bf16 output line (j = 0, 1) are written back to back - with the column block outermost they were
    // four row blocks apart and WRITE_SIZE rose to 1.5x the output bytes (partial lines evicted before completion).
    int nj[NTL]; bool okj[NTL]; f32x4 bvj[NTL];
#pragma unroll
    for (int j = 0; j < NTL; ++j) {
      nj[j] = ncol0 + j * 32 + c * 4;
      okj[j] = nj[j] < p.N;
      bvj[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p.bias && okj[j]) bvj[j] = *(const f32x4*)(p.bias + nj[j]);
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int mb = mrow0 + i * 32;
#pragma unroll
      for (int j = 0; j < NTL; ++j) {
        const int n = nj[j];
        const bool col_ok = okj[j];
        const f32x4 bv = bvj[j];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 t = {acc[i][j][q * 4 + 0], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]};
          *(f32x4*)(wr + (((q * 2 + fg) ^ wsw) << 4)) = t;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
          const int r = pass * 8 + rsub;
          const int m = mb + r;
          f32x4 v = *(const f32x4*)(wl + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
          if (m >= p.M || !col_ok) continue;
          v = v * p.alpha + bv;
          if constexpr (EPI == EPI_BF16) {
            if (p.act == 4) {
              unsigned int y0, g0, y1, g1;
              gelu_and_grad_pk(pack2bf(v[0], v[1]), y0, g0);
              gelu_and_grad_pk(pack2bf(v[2], v[3]), y1, g1);
              const u32x2 o = {y0, y1}, d = {g0, g1};
              *(u32x2*)((bf16_t*)p.out2 + (size_t)m * p.ldo + n) = d;
              *(u32x2*)((bf16_t*)p.out + (size_t)m * p.ldo + n) = o;
              continue;
            }
            if (p.act == 1) {
              if (p.out2) {
                u32x2 o2; o2[0] = pack2bf(v[0], v[1]); o2[1] = pack2bf(v[2], v[3]);
                *(u32x2*)((bf16_t*)p.out2 + (size_t)m * p.ldo + n) = o2;
              }
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
            } else if (p.act == 2) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            u32x2 o; o[0] = pack2bf(v[0], v[1]); o[1] = pack2bf(v[2], v[3]);
            *(u32x2*)((bf16_t*)p.out + (size_t)m * p.ldo + n) = o;
          } else if constexpr (EPI == EPI_F32) {
            *(f32x4*)((float*)p.out + (size_t)m * p.ldo + n) = v;
          } else if constexpr (EPI == EPI_RES_F32) {
            const f32x4 rr = *(const f32x4*)((const float*)p.res + (size_t)m * p.ldo + n);
            *(f32x4*)((float*)p.out + (size_t)m * p.ldo + n) = v + rr;
          } else if constexpr (EPI == EPI_RES_BF16) {
            const u32x2 rr = *(const u32x2*)((const bf16_t*)p.res + (size_t)((m + p.m_off) / p.res_div) * p.ldo + n);
            v[0] += bf2f((bf16_t)(rr[0] & 0xffff)); v[1] += bf2f((bf16_t)(rr[0] >> 16));
            v[2] += bf2f((bf16_t)(rr[1] & 0xffff)); v[3] += bf2f((bf16_t)(rr[1] >> 16));
            if (p.act == 2) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            u32x2 o; o[0] = pack2bf(v[0], v[1]); o[1] = pack2bf(v[2], v[3]);
            *(u32x2*)((bf16_t*)p.out + (size_t)m * p.ldo + n) = o;
          } else if constexpr (EPI == EPI_DGELU) {
            const u32x2 rr = *(const u32x2*)((const bf16_t*)p.res + (size_t)m * p.ldo + n);
            const bool saved = p.act == 4;
            const float r0 = bf2f((bf16_t)(rr[0] & 0xffff)), r1 = bf2f((bf16_t)(rr[0] >> 16));
            const float r2 = bf2f((bf16_t)(rr[1] & 0xffff)), r3 = bf2f((bf16_t)(rr[1] >> 16));
            v[0] *= saved ? r0 : gelu_erf_grad(r0); v[1] *= saved ? r1 : gelu_erf_grad(r1);
            v[2] *= saved ? r2 : gelu_erf_grad(r2); v[3] *= saved ? r3 : gelu_erf_grad(r3);
            u32x2 o; o[0] = pack2bf(v[0], v[1]); o[1] = pack2bf(v[2], v[3]);
            *(u32x2*)((bf16_t*)p.out + (size_t)m * p.ldo + n) = o;
          }
        }
        asm volatile("" ::: "memory");
      }
    }
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI, bool DMA>
__global__ void __launch_bounds__(WAVES_M* WAVES_N * 64)
    gemm_nt_kernel(const GemmP p) {
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int NT = NW * 64;
  constexpr int WTM = BM / WAVES_M;  // wave tile rows of C (activation rows)
  constexpr int WTN = BN / WAVES_N;  // wave tile cols of C (weight rows)
  constexpr int MT = WTM / 32, NTL = WTN / 32;
  using S = Smem<BM, BN>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int nwg = tiles_m * tiles_n;
  const int pid = xcd_remap(blockIdx.x, nwg);
  const int tm = pid / tiles_n, tn = pid % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wid % WAVES_M, wave_n = wid / WAVES_M;

  int nk = p.K >> 6;
  size_t kbeg = 0;
  if (p.ksplit_len) {
    const int s0 = blockIdx.y * p.ksplit_len;
    nk = min(p.ksplit_len, nk - s0);
    kbeg = (size_t)s0 << 6;
  }

  // ---- staging helpers -------------------------------------------------------------------
  // one wave-instruction moves 8 rows x 128 B.  lane -> (row in group = lane>>3, phys chunk = lane&7)
  const int srow = lane >> 3, pch = lane & 7;
  constexpr int AI = BM / (8 * NW), BI = BN / (8 * NW);
  // per-lane global source pointers (K offset added per tile) and LDS destinations
  const bf16_t* gA[AI]; const bf16_t* gB[BI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int row = (i * NW + wid) * 8 + srow;
    int gm = m0 + row; gm = gm < p.M ? gm : p.M - 1;
    gA[i] = p.A + (size_t)gm * p.lda + (pch ^ ((row >> 1) & 7)) * 8 + kbeg;
  }
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int row = (i * NW + wid) * 8 + srow;
    int gn = n0 + row; gn = gn < p.N ? gn : p.N - 1;
    gB[i] = p.W + (size_t)gn * p.ldw + (pch ^ ((row >> 1) & 7)) * 8 + kbeg;
  }
  [[maybe_unused]] u32x4 rA[AI], rB[BI];
  // LDS-DMA: one instruction per 8-row group, destination = wave-uniform base + lane*16
  auto stage_dma = [&](int kt, int buf) {
    unsigned char* sA = smem + buf * S::STAGE;
    unsigned char* sB = sA + S::A_BYTES;
    const int k0 = kt << 6;
#pragma unroll
    for (int i = 0; i < AI; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gA[i] + k0),
                                       (__attribute__((address_space(3))) void*)(sA + (i * NW + wid) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < BI; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gB[i] + k0),
                                       (__attribute__((address_space(3))) void*)(sB + (i * NW + wid) * 1024), 16, 0, 0);
  };
  auto stage_load = [&](int kt) {
    const int k0 = kt << 6;
#pragma unroll
    for (int i = 0; i < AI; ++i) rA[i] = *(const u32x4*)(gA[i] + k0);
#pragma unroll
    for (int i = 0; i < BI; ++i) rB[i] = *(const u32x4*)(gB[i] + k0);
  };
  auto stage_write = [&](int buf) {
    unsigned char* sA = smem + buf * S::STAGE;
    unsigned char* sB = sA + S::A_BYTES;
#pragma unroll
    for (int i = 0; i < AI; ++i) *(u32x4*)(sA + (i * NW + wid) * 1024 + lane * 16) = rA[i];
#pragma unroll
    for (int i = 0; i < BI; ++i) *(u32x4*)(sB + (i * NW + wid) * 1024 + lane * 16) = rB[i];
  };

  f32x16 acc[MT][NTL];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTL; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int fr = lane & 31, fg = lane >> 5;
  const int fsw = (fr >> 1) & 7;  // swizzle term of this lane's fragment rows (tile bases are x32)

  if constexpr (DMA) { stage_dma(0, 0); } else { stage_load(0); stage_write(0); }
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) { if constexpr (DMA) stage_dma(kt + 1, buf ^ 1); else stage_load(kt + 1); }
    const unsigned char* sA = smem + buf * S::STAGE + (wave_m * WTM + fr) * 128;
    const unsigned char* sB = smem + buf * S::STAGE + S::A_BYTES + (wave_n * WTN + fr) * 128;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int off = ((kk * 2 + fg) ^ fsw) * 16;
      bf16x8 af[MT], wf[NTL];
#pragma unroll
      for (int i = 0; i < MT; ++i) af[i] = *(const bf16x8*)(sA + i * 32 * 128 + off);
#pragma unroll
      for (int j = 0; j < NTL; ++j) wf[j] = *(const bf16x8*)(sB + j * 32 * 128 + off);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
    if constexpr (!DMA) { if (kt + 1 < nk) stage_write(buf ^ 1); }
    __syncthreads();
  }

  {
    GemmP pe = reload_params();
    if constexpr (EPI == EPI_F32) { if (pe.ksplit_len) pe.out = (float*)pe.out + (size_t)blockIdx.y * pe.split_stride; }
    store_tile<EPI, MT, NTL>(pe, acc, m0 + wave_m * WTM, n0 + wave_n * WTN, fr, fg);
  }
}

// out[m, n] += sum_z ws[z][m, n]   (deterministic second stage of the split-K GEMM)
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const float* __restrict__ ws, int splits, int M, int N, float* __restrict__ out,
                                                            long ldo) {
  const int n4 = N >> 2;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)M * n4) return;
  const long m = i / n4; const int n = (int)(i - m * n4) * 4;
  f32x4 a = *(const f32x4*)(out + m * ldo + n);
  const size_t plane = (size_t)M * N;
  for (int z = 0; z < splits; ++z) a += *(const f32x4*)(ws + z * plane + m * N + n);
  *(f32x4*)(out + m * ldo + n) = a;
}

template <int BM, int BN, int WM, int WN, int EPI, bool DMA>
hipError_t launch(const GemmP& p, hipStream_t s) {
  using S = Smem<BM, BN>;
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  auto kern = gemm_nt_kernel<BM, BN, WM, WN, EPI, DMA>;
  constexpr int smem = 2 * S::STAGE;
  static const hipError_t attr = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);   // C++11 magic static: thread-safe, once
  if (attr != hipSuccess) return attr;
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(WM * WN * 64), smem, s, p);
  return hipGetLastError();
}


// ------------------------------------------------------------------------------------------------
// Tail kernel for the few rows the persistent launch leaves over (the cls token makes M = B*257, one 256-row tile
// more than a whole number of rounds).  On 128x128 tiles those rows are 8-32 workgroups that each walk the full K
// alone (44-136 us with 90 % of the chip idle); here every 32x32 output tile gets its own workgroup and the
// reduction is split 8 ways over its waves, each streaming its k-slices straight from L2 into MFMA operand
// registers (no LDS staging: a slice is touched once), partial sums meet in LDS, wave 0 runs the epilogue.
template <int EPI>
__global__ void __launch_bounds__(512) gemm_tail_kernel(const GemmP p) {
  __shared__ float red[7][16][64];
  const int tiles_n = (p.N + 31) >> 5;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
  const int m0 = tm << 5, n0 = tn << 5;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 31, fg = lane >> 5;
  int gm = m0 + fr; gm = gm < p.M ? gm : p.M - 1;
  int gn = n0 + fr; gn = gn < p.N ? gn : p.N - 1;
  const bf16_t* ap = p.A + (size_t)gm * p.lda + fg * 8;
  const bf16_t* wp = p.W + (size_t)gn * p.ldw + fg * 8;
  const int steps = p.K >> 4;
  f32x16 acc[1][1];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
  int s = wid;
  for (; s + 24 < steps; s += 32) {
    bf16x8 af[4], wf[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { af[u] = *(const bf16x8*)(ap + (s + 8 * u) * 16); wf[u] = *(const bf16x8*)(wp + (s + 8 * u) * 16); }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[u], af[u], acc[0][0], 0, 0, 0);
  }
  for (; s < steps; s += 8) {
    const bf16x8 af = *(const bf16x8*)(ap + s * 16), wf = *(const bf16x8*)(wp + s * 16);
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, af, acc[0][0], 0, 0, 0);
  }
  if (wid) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wid - 1][r][lane] = acc[0][0][r];
  }
  __syncthreads();
  if (wid == 0) {
#pragma unroll
    for (int w = 0; w < 7; ++w)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][0][r] += red[w][r][lane];
    const GemmP pe = reload_params();
    store_tile<EPI, 1, 1>(pe, acc, m0, n0, fr, fg);
  }
}

template <int EPI>
hipError_t launch_tail(const GemmP& p, hipStream_t s) {
  const int tiles = ((p.M + 31) / 32) * ((p.N + 31) / 32);
  hipLaunchKernelGGL(gemm_tail_kernel<EPI>, dim3(tiles), dim3(512), 0, s, p);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Persistent variant: one workgroup per CU walks its list of output tiles; the (tile, k-step)
// sequence is flattened so the LDS-DMA of step s+1 -- including the FIRST k-slab of the next tile
// -- is always in flight while step s computes and while a finished tile is being stored.
// MFMA operand fragments are double-buffered in registers across the four 16-wide k-substeps.
// Tile order: groups of GN consecutive N-tiles, M fastest inside a group, and each XCD owns a
// contiguous run of G/8 tiles per round -> the 4 weight slabs of a group stay resident in that
// XCD's L2 across rounds while activation tiles stream through once per group.
// Hand-scheduled k-step: the 8 LDS-DMA instructions of the next step and the 6
// fragment reads of the next 16-wide k-substep are issued ahead of the MFMAs of the current
// substep (sched_barrier fences pin the phases), so neither DMA issue cost nor ds_read latency sits
// in front of the matrix pipe.  The k-step body is one branch-free basic block: source pointers of
// the step after next are prepared at the end of the previous step, and the last step of a
// workgroup re-issues a harmless reload instead of branching around the DMA.
template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI>
__global__ void __launch_bounds__(WAVES_M* WAVES_N * 64)
    gemm_nt_persist2_kernel(const GemmP p) {
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int MT = WTM / 32, NTL = WTN / 32;
  constexpr int AI = BM / (8 * NW), BI = BN / (8 * NW);
  static_assert((NW == 8 && AI == 4 && (BI == 4 || BI == 2) && NTL == 2 && (MT == 4 || MT == 2)) ||
                    (NW == 4 && AI == 8 && BI == 8 && MT == 4 && NTL == 4),
                "schedule written for 8 waves on a 256x256 (2x4) / 256x128 (4x2) tile, or 4 waves (2x2, one per SIMD) on 256x256");
  constexpr int NU = AI + BI;          // 1 KB LDS-DMA units per wave per k-step
  constexpr int GN = 4;
  using S = Smem<BM, BN>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int ntiles = tiles_m * tiles_n;
  const int G = gridDim.x;
  const int slot = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  if (slot >= ntiles) return;
  const int my_tiles = (ntiles - slot + G - 1) / G;
  const int nk = p.K >> 6;
  const int total = my_tiles * nk;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wid % WAVES_M, wave_n = wid / WAVES_M;
  const int srow = lane >> 3, pch = lane & 7;
  const int fr = lane & 31, fg = lane >> 5;
  const int fsw = (fr >> 1) & 7;

  auto tile_origin = [&](int ti, int& m0, int& n0) {
    const int v = ti * G + slot;
    const int gsz = GN * tiles_m;
    const int gid = v / gsz, rem = v - gid * gsz;
    const int first_n = gid * GN;
    const int gn = min(tiles_n - first_n, GN);
    const int tm = rem / gn;
    m0 = tm * BM; n0 = (first_n + (rem - tm * gn)) * BN;
  };
  // src[0..AI) = A row groups, src[AI..NU) = W row groups; pointers already include the k offset
  const bf16_t* src[NU];
  auto set_sources = [&](int m0, int n0, int k0) {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int row = (i * NW + wid) * 8 + srow;
      const int sw = (pch ^ ((row >> 1) & 7)) * 8;
      int gm = m0 + row; gm = gm < p.M ? gm : p.M - 1;
      src[i] = p.A + (size_t)gm * p.lda + sw + k0;
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int row = (i * NW + wid) * 8 + srow;
      const int sw = (pch ^ ((row >> 1) & 7)) * 8;
      int gn_ = n0 + row; gn_ = gn_ < p.N ? gn_ : p.N - 1;
      src[AI + i] = p.W + (size_t)gn_ * p.ldw + sw + k0;
    }
  };
  auto dma = [&](int i, unsigned char* stage_base) {
    unsigned char* dst = stage_base + (i < AI ? 0 : S::A_BYTES) + ((i < AI ? i : i - AI) * NW + wid) * 1024;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src[i],
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };

  f32x16 acc[MT][NTL];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTL; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int cur_m0, cur_n0, nxt_m0, nxt_n0;
  tile_origin(0, cur_m0, cur_n0);
  nxt_m0 = cur_m0; nxt_n0 = cur_n0;
  set_sources(cur_m0, cur_n0, 0);
#pragma unroll
  for (int i = 0; i < NU; ++i) dma(i, smem);
  // sources for the DMA issued during step 0 (= data of step 1)
  int kt = 0, ti = 0;          // position of the step being computed
  int lkt = 0, lti = 0;        // position of the step whose data the in-loop DMA loads (s+1)
  auto advance_load = [&]() {
    ++lkt;
    if (lkt == nk) { lkt = 0; ++lti; if (lti < my_tiles) tile_origin(lti, nxt_m0, nxt_n0); else { lti = my_tiles - 1; } }
    set_sources(nxt_m0, nxt_n0, lkt << 6);
  };
  advance_load();
  __syncthreads();

  for (int s = 0; s < total; ++s) {
    unsigned char* cur = smem + (s & 1) * S::STAGE;
    unsigned char* oth = smem + ((s & 1) ^ 1) * S::STAGE;
    const unsigned char* sA = cur + (wave_m * WTM + fr) * 128;
    const unsigned char* sB = cur + S::A_BYTES + (wave_n * WTN + fr) * 128;
    bf16x8 af[2][MT], wf[2][NTL];
    auto ldfrag = [&](int kk, int c) {
      const int off = ((kk * 2 + fg) ^ fsw) * 16;
#pragma unroll
      for (int j = 0; j < NTL; ++j) wf[c][j] = *(const bf16x8*)(sB + j * 32 * 128 + off);
#pragma unroll
      for (int i = 0; i < MT; ++i) af[c][i] = *(const bf16x8*)(sA + i * 32 * 128 + off);
    };
    auto mma = [&](int c) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[c][j], af[c][i], acc[i][j], 0, 0, 0);
    };
    // phase fences: nothing moves across, so each phase ISSUES the next phase's fragment reads (and
    // half of the next step's DMA) ahead of its own MFMAs and the waits land one phase later.
    ldfrag(0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NU / 2; ++i) dma(i, oth);
    ldfrag(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    mma(0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = NU / 2; i < NU; ++i) dma(i, oth);
    ldfrag(2, 0);
    __builtin_amdgcn_sched_barrier(0);
    mma(1);
    __builtin_amdgcn_sched_barrier(0);
    ldfrag(3, 1);
    __builtin_amdgcn_sched_barrier(0);
    mma(0);
    __builtin_amdgcn_sched_barrier(0);
    mma(1);
    __builtin_amdgcn_sched_barrier(0);

    if (kt + 1 == nk) {
      {
        const GemmP pe = reload_params();
        store_tile_lds<EPI, MT, NTL>(pe, acc, cur_m0 + wave_m * WTM, cur_n0 + wave_n * WTN, fr, fg, lane, smem + 2 * S::STAGE + wid * (32768 / NW));
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      kt = 0; ++ti;
      if (ti < my_tiles) tile_origin(ti, cur_m0, cur_n0);
    } else {
      ++kt;
    }
    advance_load();     // pointers for the DMA of the next iteration (data of step s+2)
    __syncthreads();
  }
}

static int query_cus() {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  return n;
}
static int num_cus() {
  static const int n = query_cus();      // thread-safe one-time query (all devices of a node are the same part)
  return n;
}

template <int EPI, int BN = 256, int NWAVES = 8>
hipError_t launch_persist(const GemmP& p, hipStream_t s) {
  // BN = 256, 8 waves: 2x4 waves of 128x64 (default).  BN = 128: 4x2 waves of 64x64 (cfg 6, measured for the round-2
  // epilogue design).  NWAVES = 4 (128x128 per wave, one wave per SIMD, accumulators in AGPRs) compiles but spills
  // 100-490 VGPRs in the epilogues as written: not instantiated until its register budget is reworked.
  using S = Smem<256, BN>;
  constexpr int WM = NWAVES == 4 ? 2 : (BN == 256 ? 2 : 4), WN = NWAVES / WM;
  const int tiles = ((p.M + 255) / 256) * ((p.N + BN - 1) / BN);
  auto kern = gemm_nt_persist2_kernel<256, BN, WM, WN, EPI>;
  constexpr int smem = 2 * S::STAGE + 8 * 4096;      // + 32 KB of epilogue-transpose slabs (160 KB at BN = 256)
  static const hipError_t attr = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);   // C++11 magic static: thread-safe, once
  if (attr != hipSuccess) return attr;
  int G = num_cus() & ~7;
  if (tiles < G) G = (tiles + 7) & ~7;
  hipLaunchKernelGGL(kern, dim3(G), dim3(NWAVES * 64), smem, s, p);
  return hipGetLastError();
}

}  // namespace
// vl_gemm_park.hip: the round-2 persistent kernel (bf16-output epilogues, whole 256x256 tiles, K >= 512)
bool vl_gemm_park_supported(int epi, const void* params);
int vl_gemm_park_launch(int epi, const void* params, int ncu, hipStream_t s);
// vl_gemm_pp.hip: the round-3 ping-pong kernel (two 4-wave workgroups per CU, 256x128 tiles, K >= 128)
bool vl_gemm_pp_supported(int epi, const void* params);
int vl_gemm_pp_launch(int epi, const void* params, int ncu, hipStream_t s);
namespace {

// the persistent kernel for a problem whose M is a whole number of tiles: the round-3 / round-2 kernels where they apply
template <int EPI>
hipError_t launch_best_persist(const GemmP& p0, hipStream_t s) {
  const GemmP& p = p0;
  // 256x256 tiles where N allows them (fewer operand bytes per flop: the chip is power-bound on these GEMMs, DESIGN.md
  // section 7); the ping-pong kernel's 256x128 tiles take N % 256 == 128 (ViT-bigG: 1664 = 13 x 128)
  if (vl_gemm_park_supported(EPI, &p)) return (hipError_t)vl_gemm_park_launch(EPI, &p, num_cus(), s);
  if (vl_gemm_pp_supported(EPI, &p)) return (hipError_t)vl_gemm_pp_launch(EPI, &p, num_cus(), s);
  return launch_persist<EPI>(p, s);
}

template <int EPI>
hipError_t dispatch(const GemmP& p, int cfg, hipStream_t s) {
  if (cfg == 8) {
    if (!vl_gemm_park_supported(EPI, &p)) return hipErrorInvalidValue;
    return (hipError_t)vl_gemm_park_launch(EPI, &p, num_cus(), s);
  }
  if (cfg == 10) {
    if (!vl_gemm_pp_supported(EPI, &p)) return hipErrorInvalidValue;
    return (hipError_t)vl_gemm_pp_launch(EPI, &p, num_cus(), s);
  }
  // cfg bit0: 0 = 256x256 tile (8 waves), 1 = 128x128 tile (4 waves); bit1: 1 = register staging
  if (cfg == 4 || cfg == 5) return launch_persist<EPI>(p, s);       // 4: historical alias
  if (cfg == 6) return launch_persist<EPI, 128>(p, s);              // 256x128 tiles (experiment)
  if (cfg == 9) return launch_tail<EPI>(p, s);
  if (cfg == 11) return launch<64, 64, 2, 2, EPI, true>(p, s);      // few-row problems: one 64x64 tile per workgroup
  if (cfg == 12) return launch<128, 64, 2, 2, EPI, true>(p, s);
  switch (cfg & 3) {
    case 0: return launch<256, 256, 2, 4, EPI, true>(p, s);
    case 1: return launch<128, 128, 2, 2, EPI, true>(p, s);
    case 2: return launch<256, 256, 2, 4, EPI, false>(p, s);
    default: return launch<128, 128, 2, 2, EPI, false>(p, s);
  }
}

}  // namespace

extern "C" int vl_set_error(const char* msg);

static int gcd_i(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }

// cfg >= 0: run exactly that kernel configuration.
// cfg == -1 (auto): problems smaller than one wave of 256x256 tiles use 128x128 tiles; larger ones run
// the persistent 256x256 kernel on the largest row range whose tile count is a multiple of the CU count
// (every CU gets the same number of tiles -> no tail round) and the few remaining rows on 128x128 tiles.
template <int EPI>
static hipError_t run_gemm(const GemmP& p, int cfg, hipStream_t s) {
  if (cfg >= 0) return dispatch<EPI>(p, cfg, s);
  const int G = num_cus() & ~7;
  const int tiles_n = (p.N + 255) / 256, full_m = p.M / 256;
  // fewer 256x256 tiles than 3/4 of the CUs: 128x128 tiles.  (Between 3/4 and one full round the persistent kernel's single
  // uneven round wins: text tower out_proj / c_proj, 77 x 3 = 231 tiles, 30 / 86 us against 89 us on 128x128 tiles.)
  if ((long)full_m * tiles_n < (G * 3) / 4) return dispatch<EPI>(p, 1, s);
  int step = G / gcd_i(G, tiles_n);
  int main_m = (full_m / step) * step;
  if ((p.N & 255) == 128) {       // 256x128 tiles on the ping-pong kernel (two workgroups per CU): whole rounds where the row
    step = 2 * G / gcd_i(2 * G, p.N >> 7);      // count allows, otherwise every full row tile (an uneven last round beats 128x128 tiles)
    main_m = (full_m / step) * step;
    if (main_m == 0) main_m = full_m;
  }
  // no whole round fits (e.g. the text tower's 77 row tiles x 9 or 12 column tiles): every full row tile on the persistent
  // kernel with an uneven last round (launch_best_persist falls back to the round-1 kernel where the round-2 one does not apply)
  if (main_m == 0) main_m = full_m;
  if (main_m == 0) return launch_persist<EPI>(p, s);
  const int rows_main = main_m * 256;
  GemmP pm = p; pm.M = rows_main;
  hipError_t e = launch_best_persist<EPI>(pm, s);
  if (e != hipSuccess || rows_main == p.M) return e;
  GemmP pr = p;
  pr.M = p.M - rows_main; pr.m_off = p.m_off + rows_main;
  pr.A = p.A + (size_t)rows_main * p.lda;
  constexpr size_t osz = (EPI == EPI_F32 || EPI == EPI_RES_F32) ? 4 : 2;
  if (p.out) pr.out = (unsigned char*)p.out + (size_t)rows_main * p.ldo * osz;
  // EPI_RES_BF16 indexes its (possibly row-broadcast) residual with the ABSOLUTE row (m + m_off) / res_div: the
  // pointer must stay put there (offsetting it as well read rows_main rows past the end - found by the c3 bench);
  // every other epilogue indexes res with the local row.
  if (p.res && EPI != EPI_RES_BF16) pr.res = (const unsigned char*)p.res + (size_t)rows_main * p.ldo * osz;
  if (p.out2) pr.out2 = (bf16_t*)p.out2 + (size_t)rows_main * p.ldo * (EPI == EPI_GEGLU ? 2 : 1);
  if (pr.M > 512) return dispatch<EPI>(pr, 1, s);
  // leftover rows: 64x64 LDS-DMA tiles when they fill most of the chip (N >= 3072 at 256 rows: 10-12 us against 15-20 us,
  // profiles/r03b_tail_probe.log), otherwise the split-K 32x32 tail kernel (long K, few columns: 17 us against 28-32 us)
  const long t64 = (long)((pr.M + 63) / 64) * ((pr.N + 63) / 64);
  return t64 >= 192 ? dispatch<EPI>(pr, 11, s) : launch_tail<EPI>(pr, s);
}

#define VL_CHECK_ARG(c, msg) do { if (!(c)) return vl_set_error(msg); } while (0)

// ---- LayerNorm folded into the GEMMs either side of it (round 4; frozen pre-LN blocks on a bf16 residual stream) ----
// LN(x) W^T + b = rstd (x (W gamma)^T - mean c) + (b + W beta),  c_n = sum_k (W gamma)[n, k]: the consuming GEMM reads the raw
// residual rows and applies the row statistics in its epilogue; the GEMM that PRODUCED those rows (out-projection /
// c_proj + residual) leaves their partial sums behind.  The normalised activations are never written or read: one
// LayerNorm pass (2 x rows x D x 2 bytes at the HBM roofline) per LayerNorm disappears.  Persistent kernel only (whole
// 256x256 tiles): the host code runs the rows vl_gemm_main_rows() reports through these entries and the leftover rows
// through vl_layernorm_fwd + vl_gemm_bf16.
extern "C" int vl_gemm_main_rows(int M, int N) {
  const int G = num_cus() & ~7;
  if (M <= 0 || N <= 0 || (N & 255)) return 0;
  const int tiles_n = N / 256, full_m = M / 256;
  if ((long)full_m * tiles_n < (G * 3) / 4) return 0;
  const int step = G / gcd_i(G, tiles_n);
  int main_m = (full_m / step) * step;
  if (main_m == 0) main_m = full_m;
  return main_m * 256;
}

extern "C" int vl_gemm_lnfold_bf16(const void* A, const void* Wg, const float* bias_f, const float* ln_c, const float* ln_mean,
                                   const float* ln_rstd, void* out, void* out2, int M, int N, int K, int lda, int ldw, int ldo,
                                   int act, hipStream_t stream) {
  VL_CHECK_ARG(A && Wg && bias_f && ln_c && ln_mean && ln_rstd && out, "vl_gemm_lnfold_bf16: null operand");
  VL_CHECK_ARG(act == VL_ACT_NONE || act == VL_ACT_GELU || act == VL_ACT_GELU_DSAVE, "vl_gemm_lnfold_bf16: act must be none, GELU or GELU_DSAVE");
  VL_CHECK_ARG((act == VL_ACT_GELU_DSAVE) == (out2 != nullptr), "vl_gemm_lnfold_bf16: out2 goes with VL_ACT_GELU_DSAVE");
  VL_CHECK_ARG((((uintptr_t)ln_c | (uintptr_t)bias_f) & 15) == 0, "vl_gemm_lnfold_bf16: column vectors must be 16-byte aligned");
  GemmP p{};
  p.A = (const bf16_t*)A; p.W = (const bf16_t*)Wg; p.bias = bias_f; p.out = out; p.out2 = out2;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldo = ldo; p.alpha = 1.0f; p.act = act; p.res_div = 1;
  p.ln_mean = ln_mean; p.ln_rstd = ln_rstd; p.ln_c = ln_c;
  const GemmP& q = p;
  VL_CHECK_ARG(vl_gemm_park_supported(EPI_BF16, &q), "vl_gemm_lnfold_bf16: whole 256x256 tiles, K >= 512, 16-byte aligned operands required");
  const hipError_t e = (hipError_t)vl_gemm_park_launch(EPI_BF16, &q, num_cus(), stream);
  if (e != hipSuccess) return vl_set_error(hipGetErrorString(e));
  return 0;
}

extern "C" int vl_gemm_res_rowstats_bf16(const void* A, const void* W, const float* bias, void* out, const void* res,
                                         float* row_part, int M, int N, int K, int lda, int ldw, int ldo, hipStream_t stream) {
  VL_CHECK_ARG(A && W && out && res && row_part, "vl_gemm_res_rowstats_bf16: null operand");
  VL_CHECK_ARG((((uintptr_t)row_part) & 7) == 0, "vl_gemm_res_rowstats_bf16: row_part must be 8-byte aligned");
  GemmP p{};
  p.A = (const bf16_t*)A; p.W = (const bf16_t*)W; p.bias = bias; p.out = out; p.res = res;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldo = ldo; p.alpha = 1.0f; p.res_div = 1;
  p.row_part = row_part;
  const GemmP& q = p;
  VL_CHECK_ARG(vl_gemm_park_supported(EPI_RES_BF16, &q), "vl_gemm_res_rowstats_bf16: whole 256x256 tiles, K >= 512, 16-byte aligned operands required");
  const hipError_t e = (hipError_t)vl_gemm_park_launch(EPI_RES_BF16, &q, num_cus(), stream);
  if (e != hipSuccess) return vl_set_error(hipGetErrorString(e));
  return 0;
}

extern "C" int vl_gemm_bf16_ex(const void* A, const void* W, const float* bias, void* out, const void* res, void* out2,
                               int M, int N, int K, int lda, int ldw, int ldo, float alpha, int epi, int act,
                               int res_div, int cfg, hipStream_t stream);
extern "C" int vl_gemm_bf16(const void* A, const void* W, const float* bias, void* out, const void* res,
                            int M, int N, int K, int lda, int ldw, int ldo, float alpha, int epi, int act,
                            int cfg, hipStream_t stream) {
  return vl_gemm_bf16_ex(A, W, bias, out, res, nullptr, M, N, K, lda, ldw, ldo, alpha, epi, act, 1, cfg, stream);
}

extern "C" int vl_gemm_bf16_ex(const void* A, const void* W, const float* bias, void* out, const void* res, void* out2,
                               int M, int N, int K, int lda, int ldw, int ldo, float alpha, int epi, int act,
                               int res_div, int cfg, hipStream_t stream) {
  VL_CHECK_ARG(res_div >= 1, "vl_gemm_bf16: res_div must be >= 1");
  VL_CHECK_ARG(res_div == 1 || epi == VL_EPI_RES_BF16, "vl_gemm_bf16: row-broadcast residual needs VL_EPI_RES_BF16");
  VL_CHECK_ARG(M > 0 && N > 0 && K > 0, "vl_gemm_bf16: empty problem");
  VL_CHECK_ARG((K & 63) == 0, "vl_gemm_bf16: K must be a multiple of 64");
  VL_CHECK_ARG((N & 3) == 0, "vl_gemm_bf16: N must be a multiple of 4");
  VL_CHECK_ARG((lda & 7) == 0 && (ldw & 7) == 0, "vl_gemm_bf16: lda/ldw must be multiples of 8");
  VL_CHECK_ARG((ldo & 3) == 0, "vl_gemm_bf16: ldo must be a multiple of 4");
  GemmP p{};
  p.A = (const bf16_t*)A; p.W = (const bf16_t*)W; p.bias = bias; p.out = out; p.res = res; p.out2 = out2;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldo = ldo; p.alpha = alpha; p.act = act; p.res_div = res_div;
  hipError_t e;
  switch (epi) {
    case VL_EPI_BF16:
      VL_CHECK_ARG(act != VL_ACT_GELU_DSAVE || out2, "vl_gemm_bf16: VL_ACT_GELU_DSAVE needs out2 (the gelu' tensor)");
      e = run_gemm<EPI_BF16>(p, cfg, stream); break;
    case VL_EPI_F32: e = run_gemm<EPI_F32>(p, cfg, stream); break;
    case VL_EPI_RES_F32: VL_CHECK_ARG(res, "vl_gemm_bf16: residual missing"); e = run_gemm<EPI_RES_F32>(p, cfg, stream); break;
    case VL_EPI_RES_BF16: VL_CHECK_ARG(res, "vl_gemm_bf16: residual missing"); e = run_gemm<EPI_RES_BF16>(p, cfg, stream); break;
    case VL_EPI_GEGLU: e = run_gemm<EPI_GEGLU>(p, cfg, stream); break;
    case VL_EPI_DGELU: VL_CHECK_ARG(res, "vl_gemm_bf16: pre-activation tensor missing"); e = run_gemm<EPI_DGELU>(p, cfg, stream); break;
    case VL_EPI_DGEGLU: VL_CHECK_ARG(res, "vl_gemm_bf16: pre-activation tensor missing"); VL_CHECK_ARG((ldo & 7) == 0, "vl_gemm_bf16: DGEGLU needs ldo % 8 == 0"); e = run_gemm<EPI_DGEGLU>(p, cfg, stream); break;
    default: return vl_set_error("vl_gemm_bf16: unknown epilogue");
  }
  if (e != hipSuccess) return vl_set_error(hipGetErrorString(e));
  return 0;
}

// IEEE-half operands on the persistent 256x256 kernel (frozen text tower).  No other kernel family carries the type: the caller
// pads its row count to whole tiles (zero rows cost nothing downstream) instead of this entry growing a tail path.
extern "C" int vl_gemm_f16(const void* A, const void* W, const float* bias, void* out, const void* res, int M, int N, int K,
                           int lda, int ldw, int ldo, float alpha, int epi, int act, hipStream_t stream) {
  VL_CHECK_ARG(A && W && out, "vl_gemm_f16: null operand");
  VL_CHECK_ARG(epi == VL_EPI_BF16 || epi == VL_EPI_RES_F32, "vl_gemm_f16: VL_EPI_BF16 (16-bit output, here fp16) or VL_EPI_RES_F32");
  VL_CHECK_ARG(epi != VL_EPI_RES_F32 || (res && act == VL_ACT_NONE), "vl_gemm_f16: VL_EPI_RES_F32 needs res and takes no activation");
  VL_CHECK_ARG(act == VL_ACT_NONE || act == VL_ACT_GELU, "vl_gemm_f16: act must be none or GELU");
  GemmP p{};
  p.A = (const bf16_t*)A; p.W = (const bf16_t*)W; p.bias = bias; p.out = out; p.res = res;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldo = ldo; p.alpha = alpha; p.act = act; p.res_div = 1; p.f16 = 1;
  const int e_ = epi == VL_EPI_BF16 ? (int)EPI_BF16 : (int)EPI_RES_F32;
  VL_CHECK_ARG(vl_gemm_park_supported(e_, &p), "vl_gemm_f16: whole 256x256 tiles (M % 256 == N % 256 == 0), K % 64 == 0, K >= 512, 16-byte aligned operands required");
  const hipError_t e = (hipError_t)vl_gemm_park_launch(e_, &p, num_cus(), stream);
  if (e != hipSuccess) return vl_set_error(hipGetErrorString(e));
  return 0;
}

// vl_gemm_tn.hip: token-major operands (At [K, M], Bt [K, N]) through the LDS transpose read
bool vl_gemm_tn_supported(const void* params);
int vl_gemm_tn_launch(const void* params, int ncu, hipStream_t s);

// out[M,N] (f32, row stride ldo) += alpha * At[K,M]^T . Bt[K,N]: the weight-gradient GEMM on the operands as the backward
// holds them (row = token).  K is cut into `splits` equal slices of whole 64-token steps; partial products go to ws
// (splits * M * N floats) and are summed in a fixed order.
extern "C" int vl_gemm_tn_splitk_accum_f32(const void* At, const void* Bt, float* out, int M, int N, int K, int lda, int ldb,
                                           long ldo, float alpha, int splits, float* ws, hipStream_t stream) {
  VL_CHECK_ARG(M > 0 && N > 0 && K > 0, "vl_gemm_tn_splitk: empty problem");
  VL_CHECK_ARG((M & 255) == 0 && (N & 255) == 0 && (K & 63) == 0, "vl_gemm_tn_splitk: M % 256, N % 256, K % 64 required");
  VL_CHECK_ARG((lda & 7) == 0 && (ldb & 7) == 0 && (ldo & 3) == 0 && lda >= M && ldb >= N, "vl_gemm_tn_splitk: lda/ldb % 8, ldo % 4, lda >= M, ldb >= N required");
  VL_CHECK_ARG(splits >= 1 && splits <= 1024 && ws, "vl_gemm_tn_splitk: 1 <= splits <= 1024 and a workspace of splits*M*N floats");
  const int nk = K >> 6;
  const int len = (nk + splits - 1) / splits, eff = (nk + len - 1) / len;      // slices of `len` 64-token steps; the last one may be shorter
  VL_CHECK_ARG(len >= 4 && nk - (eff - 1) * len >= 4, "vl_gemm_tn_splitk: every K slice needs >= 4 steps of 64");
  GemmP p{};
  p.A = (const bf16_t*)At; p.W = (const bf16_t*)Bt; p.out = ws; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldb; p.ldo = N;
  p.alpha = alpha; p.res_div = 1; p.split_stride = (long)M * N; p.ksplit_len = len;
  VL_CHECK_ARG(vl_gemm_tn_supported(&p), "vl_gemm_tn_splitk: operands must be 16-byte aligned and a slice below 2 GB");
  hipError_t e = (hipError_t)vl_gemm_tn_launch(&p, num_cus(), stream);
  if (e != hipSuccess) return vl_set_error(hipGetErrorString(e));
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)(((long)M * (N >> 2) + 255) / 256)), dim3(256), 0, stream, ws, eff, M, N, out, ldo);
  e = hipGetLastError();
  if (e != hipSuccess) return vl_set_error(hipGetErrorString(e));
  return 0;
}

// out[M,N] (f32, row stride ldo) += alpha * A[M,K] . W[N,K]^T with the K range cut into `splits` slices computed by
// separate workgroups (weight-gradient GEMMs: tiny M x N, K = number of tokens) and summed in a fixed order.
extern "C" int vl_gemm_splitk_accum_f32(const void* A, const void* W, float* out, int M, int N, int K, int lda, int ldw,
                                        long ldo, float alpha, int splits, float* ws, hipStream_t stream) {
  VL_CHECK_ARG(M > 0 && N > 0 && K > 0, "vl_gemm_splitk: empty problem");
  VL_CHECK_ARG((K & 63) == 0 && (N & 3) == 0 && (lda & 7) == 0 && (ldw & 7) == 0 && (ldo & 3) == 0,
               "vl_gemm_splitk: K % 64, N % 4, lda/ldw % 8, ldo % 4 required");
  VL_CHECK_ARG(splits >= 1 && splits <= 1024 && ws, "vl_gemm_splitk: 1 <= splits <= 1024 and a workspace of splits*M*N floats");
  const int nk = K >> 6;
  GemmP p{};
  p.A = (const bf16_t*)A; p.W = (const bf16_t*)W; p.out = ws; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldo = N;
  p.alpha = alpha; p.res_div = 1;
  p.split_stride = (long)M * N;
  if (nk % splits == 0) {       // whole 256x256 tiles and equal slices: the persistent kernels write the partials
    p.ksplit_len = nk / splits;
    if (vl_gemm_park_supported(EPI_F32, &p)) {
      hipError_t e = (hipError_t)vl_gemm_park_launch(EPI_F32, &p, num_cus(), stream);
      if (e != hipSuccess) return vl_set_error(hipGetErrorString(e));
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)(((long)M * (N >> 2) + 255) / 256)), dim3(256), 0, stream, ws, splits, M, N, out, ldo);
      e = hipGetLastError();
      if (e != hipSuccess) return vl_set_error(hipGetErrorString(e));
      return 0;
    }
  }
  p.ksplit_len = (nk + splits - 1) / splits;
  const int eff = (nk + p.ksplit_len - 1) / p.ksplit_len;
  p.split_stride = (long)M * N;
  using S = Smem<128, 128>;
  auto kern = gemm_nt_kernel<128, 128, 2, 2, EPI_F32, true>;
  constexpr int smem = 2 * S::STAGE;
  static const hipError_t attr = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  if (attr != hipSuccess) return vl_set_error(hipGetErrorString(attr));
  const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
  hipLaunchKernelGGL(kern, dim3(tiles, eff), dim3(256), smem, stream, p);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)(((long)M * (N >> 2) + 255) / 256)), dim3(256), 0, stream, ws, eff, M, N, out, ldo);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return vl_set_error(hipGetErrorString(e));
  return 0;
}

