// Shared pieces of the gfx950 attention kernels (forward: vl_attn.hip, backward: vl_attn_bwd.hip).
//
// Operand addressing: every [batch, head, row, dh] operand is a strided view (struct TV), so the kernels read q / k / v
// straight out of the packed in-projection output [tokens, 3*width] and dO out of the out-projection's input gradient
// [tokens, width] -- no head-split copy, no transposed copy (round 1 made the GEMM epilogues write both with 2-byte
// scattered stores, which doubled those GEMMs' epilogue time).  Whatever an MFMA needs transposed is transposed while
// the chunk is staged into LDS.
//
// LDS images of a chunk of NR rows (keys or queries) of a [rows, DH] matrix:
//   row image  [NR][DH*2 bytes], 16-byte chunks XOR-swizzled by (row >> RSH)   -> fragment (row = lane, 8 consecutive d)
//   T image    [DH][NR+8] bf16, the row index permuted inside each 16-row slice (bits 2 and 3 swapped) so that the
//              8 rows an accumulator-layout lane owns are CONTIGUOUS          -> fragment (d = lane) is one ds_read_b128
// The permutation: a 32x32x16 MFMA accumulator lane (column j, half fg) holds rows (r&3) + 8*(r>>2) + 4*fg, r = 0..15.
// Used as the next MFMA's B operand (k = 8*fg + e, e = 0..7, slice c = r>>3) the lane's e-th value belongs to row
// 16c + 8(e>>2) + 4fg + (e&3); the A operand must present the same rows in the same slots.
#pragma once
#include "vl_common.h"

// Phase timeline for tools/attn_phase_prof.hip (built with -DVL_ATTN_PROF): wave 0 of every workgroup stamps the shader
// clock at the phase boundaries.  Compiles to nothing in the library.
#ifdef VL_ATTN_PROF
extern "C" long* vl_attn_prof_buf;     // [workgroups][8] on the device, set by the tool
#define VL_PROF_FIELD long* prof;
#define VL_PROF_STAMP(p, i)                                                                                    \
  do {                                                                                                         \
    if ((p).prof && threadIdx.x == 0)                                                                          \
      (p).prof[((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define VL_PROF_FIELD
#define VL_PROF_STAMP(p, i) do { } while (0)
#endif

namespace vlattn {

struct TV {            // element (b, h, l, d) at p[b*sb + h*sh + l*sr + d]
  const bf16_t* p;
  long sb, sh, sr;
};

__device__ __forceinline__ int vpos16(int j) { return (j & 3) | ((j & 4) << 1) | ((j & 8) >> 1); }

__device__ __forceinline__ u32x4 scale_bf16x8(u32x4 v, float s) {
#pragma unroll
  for (int e = 0; e < 4; ++e)
    v[e] = pack2bf(bf2f((bf16_t)(v[e] & 0xffffu)) * s, bf2f((bf16_t)(v[e] >> 16)) * s);
  return v;
}

// The forward kernel also runs on IEEE-half operands (F16: the frozen text tower, vl_attn_fwd_f16): fragments stay raw 16-byte
// values typed bf16x8, the three places that interpret the bits are below.
template <bool F16>
__device__ __forceinline__ u32x4 scale_x8(u32x4 v, float s) {
  if constexpr (F16) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = pack2h(h2f((uint16_t)(v[e] & 0xffffu)) * s, h2f((uint16_t)(v[e] >> 16)) * s);
    return v;
  } else {
    return scale_bf16x8(v, s);
  }
}
template <bool F16>
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// Staging of a chunk: rows [row0, row0+NR) of ONE or TWO strided [L, DH] matrices (rows >= L zero-filled) into their row
// image and / or transposed image.  One work item = one 16-byte chunk of TWO consecutive rows (the pair becomes one
// dword per d in the T image).  ALL the loads of a round (U items x 2 rows x up to 2 matrices = 12 x 16 bytes per
// thread with 512 threads and 288 rows) are issued before the first LDS write: a workgroup pays ONE memory round trip
// per chunk instead of one per matrix and per 1024 items (measured: 59 of 222 us of the forward were staging latency).
struct StageSrc {
  unsigned char* rows;   // row image or null
  bf16_t* t;             // T image or null
  const bf16_t* src;
  long sr;
  float scale;           // applied while copying when != 1 (q is scaled by softmax_scale*log2e)
  int nch = 0;           // padded head dims (DH = 128 instantiation, real dim 72..128): valid 16-byte chunks per row
};

// rows per 256-byte LDS bank row = 2^RSH (the XOR swizzle of the row image is keyed by row >> RSH)
template <int DH> struct Rsh { static constexpr int v = DH == 32 ? 2 : (DH == 64 ? 1 : 0); };

template <int DH, int NR, bool ROWS, bool TRANS>
__device__ __forceinline__ void stage_write(const StageSrc& m, int i, u32x4 a, u32x4 bq) {
  constexpr int RB = DH * 2, CH = RB / 16, RSH = Rsh<DH>::v, TS = NR + 8;
  const int rp = i / CH, c = i % CH;
  const int rl = 2 * rp;
  if (m.scale != 1.0f) { a = scale_bf16x8(a, m.scale); bq = scale_bf16x8(bq, m.scale); }
  if constexpr (ROWS) {
    *(u32x4*)(m.rows + rl * RB + ((c ^ ((rl >> RSH) & (CH - 1))) * 16)) = a;
    *(u32x4*)(m.rows + (rl + 1) * RB + ((c ^ (((rl + 1) >> RSH) & (CH - 1))) * 16)) = bq;
  }
  if constexpr (TRANS) {
    unsigned int* dst = (unsigned int*)(m.t + (c * 8) * TS + (rl & ~15) + vpos16(rl & 15));
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      dst[(2 * w) * (TS / 2)] = (a[w] & 0xffffu) | (bq[w] << 16);
      dst[(2 * w + 1) * (TS / 2)] = (a[w] >> 16) | (bq[w] & 0xffff0000u);
    }
  }
}

template <int DH, int NR, bool ROWS0, bool TRANS0, bool ROWS1, bool TRANS1, int U = 3>
__device__ __forceinline__ void stage2(const StageSrc& m0, const StageSrc& m1, int row0, int L, int tid, int nthr) {
  constexpr int CH = DH / 8, NP = (NR / 2) * CH;
  for (int base = 0; base < NP; base += U * nthr) {
    u32x4 a0[U], b0[U], a1[U], b1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = base + u * nthr + tid;
      const int rp = i / CH, c = i % CH;
      const int row = row0 + 2 * rp;
      const u32x4 z = {0u, 0u, 0u, 0u};
      a0[u] = z; b0[u] = z; a1[u] = z; b1[u] = z;
      // 32-bit byte offsets from the uniform base pointers: global_load ... saddr + voffset, one VGPR per address
      const unsigned o0 = ((unsigned)row * (unsigned)m0.sr + c * 8) * 2u, o1 = ((unsigned)row * (unsigned)m1.sr + c * 8) * 2u;
      const bool cok = DH != 128 || c < m0.nch;        // padded head dim: chunks beyond the real row stay zero
      if (i < NP && row < L && cok) {
        a0[u] = *(const u32x4*)((const unsigned char*)m0.src + o0);
        a1[u] = *(const u32x4*)((const unsigned char*)m1.src + o1);
      }
      if (i < NP && row + 1 < L && cok) {
        b0[u] = *(const u32x4*)((const unsigned char*)m0.src + (o0 + (unsigned)m0.sr * 2u));
        b1[u] = *(const u32x4*)((const unsigned char*)m1.src + (o1 + (unsigned)m1.sr * 2u));
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = base + u * nthr + tid;
      if (i >= NP) continue;
      stage_write<DH, NR, ROWS0, TRANS0>(m0, i, a0[u], b0[u]);
      stage_write<DH, NR, ROWS1, TRANS1>(m1, i, a1[u], b1[u]);
    }
  }
}

// ---- cross-lane helpers on the VALU (no LDS round trip) -------------------------------------------------------------
// v_permlane32_swap: r[0] = {x[0:31], y[0:31]}, r[1] = {x[32:63], y[32:63]}  (lanes 0-31 | lanes 32-63)
// (the two results are copied into scalars before any bit_cast: hipcc 7.2 folds __builtin_bit_cast(float, r[1]) on the
//  builtin's const vector result to element 0 - caught by the parity tests, the row sums came out as 2 x own half)
__device__ __forceinline__ float xhalf_max(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const unsigned lo = r[0], hi = r[1];
  return fmaxf(__builtin_bit_cast(float, lo), __builtin_bit_cast(float, hi));
}
__device__ __forceinline__ float xhalf_sum(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const unsigned lo = r[0], hi = r[1];
  return __builtin_bit_cast(float, lo) + __builtin_bit_cast(float, hi);
}
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_f(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
// whole-wave reductions: quad swaps, half-row / row mirrors (DPP), then the four rows through v_readlane
__device__ __forceinline__ float wave_max_dpp(float v) {
  v = fmaxf(v, dpp_f<0xB1>(v)); v = fmaxf(v, dpp_f<0x4E>(v)); v = fmaxf(v, dpp_f<0x141>(v)); v = fmaxf(v, dpp_f<0x140>(v));
  return fmaxf(fmaxf(lane_f(v, 0), lane_f(v, 16)), fmaxf(lane_f(v, 32), lane_f(v, 48)));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += dpp_f<0xB1>(v); v += dpp_f<0x4E>(v); v += dpp_f<0x141>(v); v += dpp_f<0x140>(v);
  return (lane_f(v, 0) + lane_f(v, 16)) + (lane_f(v, 32) + lane_f(v, 48));
}

// Store a wave's 32 x DH result held transposed in accumulator layout (lane = row fr, slots = d) as rows of a
// token-major matrix.  A lane owns 4-element d groups alternating with its partner (fg ^ 1); one v_permlane32_swap per
// dword gives every lane whole 8-element (16-byte) pieces: 4 stores of 16 bytes instead of 16 of 8 (the forward spent
// 50 of 222 us in its 8-byte epilogue stores).  dst = the row of lane fr (d = 0); `valid` masks padded rows.
// npiece: number of valid 16-byte pieces per row (padded head dims store only the real columns)
template <int DT, bool F16 = false>
__device__ __forceinline__ void store_rows_t(const f32x16* acc, float mul, bf16_t* dst, int fg, bool valid, int npiece = 4 * DT) {
  auto pk = [](float a, float b) { if constexpr (F16) return pack2h(a, b); else return pack2bf(a, b); };
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
      unsigned ev[2], od[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        ev[j] = pk(acc[t][(2 * pp) * 4 + 2 * j] * mul, acc[t][(2 * pp) * 4 + 2 * j + 1] * mul);
        od[j] = pk(acc[t][(2 * pp + 1) * 4 + 2 * j] * mul, acc[t][(2 * pp + 1) * 4 + 2 * j + 1] * mul);
      }
      auto r0 = __builtin_amdgcn_permlane32_swap(ev[0], od[0], false, false);
      auto r1 = __builtin_amdgcn_permlane32_swap(ev[1], od[1], false, false);
      const unsigned w0 = r0[0], w1 = r1[0], w2 = r0[1], w3 = r1[1];
      const u32x4 w = {w0, w1, w2, w3};
      if (valid && t * 4 + 2 * pp + fg < npiece) *(u32x4*)(dst + t * 32 + (2 * pp + fg) * 8) = w;
    }
}

// fragment of the row image: row `row`, d-slice (ks, fg) -> 8 consecutive d
template <int DH>
__device__ __forceinline__ bf16x8 frag_rows(const unsigned char* base, int row, int ks, int fg) {
  constexpr int RB = DH * 2, CH = RB / 16, RSH = Rsh<DH>::v;
  return *(const bf16x8*)(base + row * RB + (((ks * 2 + fg) ^ ((row >> RSH) & (CH - 1))) * 16));
}
// fragment of the T image: d = `d`, rows of tile `tile`, slice c, half fg (already in accumulator order)
template <int NR>
__device__ __forceinline__ bf16x8 frag_t(const bf16_t* base, int d, int tile, int c, int fg) {
  return *(const bf16x8*)(base + d * (NR + 8) + tile * 32 + c * 16 + fg * 8);
}

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}
__device__ __forceinline__ bf16x8 zero_bf8() {
  bf16x8 z;
#pragma unroll
  for (int e = 0; e < 8; ++e) z[e] = (__bf16)0.f;
  return z;
}
// pack accumulator slots [c*8, c*8+8) to a bf16 operand
__device__ __forceinline__ bf16x8 pack8(const float* v) {
  bf16x8 r;
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = (__bf16)v[e];
  return r;
}
template <bool F16>
__device__ __forceinline__ bf16x8 pack8x(const float* v) {
  if constexpr (F16) {
    f16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (_Float16)v[e];      // probabilities relative to a running maximum: <= 2^8, no clamp needed
    return __builtin_bit_cast(bf16x8, r);
  } else {
    return pack8(v);
  }
}
// dot product of two bf16x8 fragments in fp32
__device__ __forceinline__ float dot8(bf16x8 a, bf16x8 b, float acc) {
#pragma unroll
  for (int e = 0; e < 8; ++e) acc = fmaf((float)a[e], (float)b[e], acc);
  return acc;
}

}  // namespace vlattn
