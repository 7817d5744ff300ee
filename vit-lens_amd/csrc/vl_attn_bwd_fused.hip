// ONE-kernel attention backward for self-attention sequences that fit a workgroup (round 4): dQ, dK, dV and
// delta = rowsum(dO * O) of one (batch, head) from ONE pass over the score tiles.
//
// The two-kernel backward of vl_attn_bwd.hip computes every 32x32 tile of S and dP twice (once per wave that owns the
// queries for dQ, once per wave that owns the keys for dK / dV): 7 matrix products and 2 x 16 exponentials per lane per
// tile, and q, k, v, dO are staged by two launches.  Here wave w owns key tile w (dK, dV in registers, K / V rows as MFMA
// column operands in registers) AND query tile w (dQ in registers).  At step t it computes tile (queries (w + t) mod n,
// keys w): S, dP, P, dS once, feeds dV += dO^T P and dK += Q^T dS from registers, and hands the bf16 dS tile through a
// 2 KB LDS slot to the wave that owns those queries, which adds K^T dS^T to its dQ accumulators in the same step:
// 5 products, 16 exponentials, one barrier per step.  The rotation (w + t) mod n gives every query tile exactly one
// producer per step.
//
// No transposed LDS images: gfx950's transpose read (`ds_read_b64_tr_b16`: lane i of a 16-lane group receives column i of
// the 4 rows x 16 columns the group addresses) produces the dO^T / Q^T / K^T / dS^T fragments straight from row-major
// images - q, dO, k (32 KB each for 256 rows of 64) + 32 KB of dS slots = 128 KB + statistics, one workgroup per CU.
// The 16-byte-chunk swizzle of the row images is f(row) = rotl3((row >> 1) & 7): conflict-free for the 16-byte row
// fragments (as the forward's) AND for the 4-row transpose reads (rows r and r + 2 share a bank row; f moves them to
// different 64-byte halves).
//
// delta is computed while dO is staged (the thread that fetches a 16-byte chunk of a dO row also fetches O's, 8 lanes
// per row reduce by DPP) - the separate pass over O of the dQ kernel and the delta workspace round trip are gone.
//
// L = 257 (8 tiles + the class token's lone row / column): the lone key's column is evaluated by the VALU from the Q / dO
// fragments the tile MFMAs use anyway, the lone query's row from the K / V register fragments; their rank-1 updates of the
// tile accumulators are 32 FMAs per lane, and the three 64-vectors that need a sum over 256 rows (dQ, dK, dV of row 256)
// go through single-row MFMAs (operand row 0 = the probability / dS vector) and a fixed-order sum over the waves.
//
// Supported: head dim 64, Lq == Lk = L, no causal mask, L <= 256 or L = 32 m + 1 <= 257; everything else stays on the
// two-kernel path (vl_attn_bwd_bf16).  Replaces: autograd of F.multi_head_attention_forward
// (open_clip/transformer.py:241-252) and of the Perceiver's latent self-attention (open_clip/perceiver.py:128-145).
#include "vl_attn_common.h"
#include "vitlens_hip.h"

namespace {
using namespace vlattn;

constexpr int FB_RB = 128;                 // bytes per image row (64 x bf16)
constexpr int FB_IMG = 256 * FB_RB;        // one row image: 256 rows
constexpr int FB_ESLOT = 2048;             // one dS tile: 32 keys x 32 queries bf16
constexpr float FB_LOG2E = 1.4426950408889634f;

typedef __attribute__((ext_vector_type(4))) short fb_s16x4;
typedef __attribute__((address_space(3))) fb_s16x4* fb_lds_s16x4;
struct FbFrag2 { fb_s16x4 lo, hi; };

struct FusedBwdP {
  TV q, k, v, dO, o;
  const float* lse;
  bf16_t *dq, *dk, *dv;     // token-major destinations (already offset to the q / k / v column block)
  long ld_dq, ld_dkv;
  int B, H, L;
  float qscale, scale;
  int l_main;               // rows handled by tiles (L, or L - 1 when the last row is the lone one)
  int tail;                 // 1: row l_main is the lone row
  VL_PROF_FIELD
};

__device__ __forceinline__ int fb_swz(int row) {
  const int x = (row >> 1) & 7;
  return ((x & 1) << 2) | (x >> 1);
}
// 16-byte row fragment: row `row`, d-slice (ks, fg) -> 8 consecutive d
__device__ __forceinline__ bf16x8 fb_rows(const unsigned char* img, int row, int ks, int fg) {
  return *(const bf16x8*)(img + row * FB_RB + (((ks * 2 + fg) ^ fb_swz(row)) << 4));
}
// Transposed fragment of a row image: this lane's column = cb*16 + (lane & 15) of the 64, for the 8 rows
// row_lo .. row_lo + 3 and row_hi .. row_hi + 3 (each 4-aligned).
__device__ __forceinline__ bf16x8 fb_tr(const unsigned char* img, int row_lo, int row_hi, int cb, int lane) {
  const int i = lane & 15;
  const int chunk = cb * 2 + ((i >> 1) & 1), half = (i & 1) << 3;
  const int r0 = row_lo + (i >> 2), r1 = row_hi + (i >> 2);
  FbFrag2 f;
  f.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((fb_lds_s16x4)(img + r0 * FB_RB + ((chunk ^ fb_swz(r0)) << 4) + half));
  f.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((fb_lds_s16x4)(img + r1 * FB_RB + ((chunk ^ fb_swz(r1)) << 4) + half));
  return __builtin_bit_cast(bf16x8, f);
}
// dS slot [32 keys][32 queries] bf16, 64-byte rows, 16-byte chunks XOR-swizzled by (key >> 1) & 3
__device__ __forceinline__ bf16x8 fb_tr_e(const unsigned char* slot, int key_lo, int key_hi, int lane) {
  const int i = lane & 15;
  const int chunk = ((lane >> 4) & 1) * 2 + ((i >> 1) & 1), half = (i & 1) << 3;
  const int r0 = key_lo + (i >> 2), r1 = key_hi + (i >> 2);
  FbFrag2 f;
  f.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((fb_lds_s16x4)(slot + r0 * 64 + ((chunk ^ ((r0 >> 1) & 3)) << 4) + half));
  f.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((fb_lds_s16x4)(slot + r1 * 64 + ((chunk ^ ((r1 >> 1) & 3)) << 4) + half));
  return __builtin_bit_cast(bf16x8, f);
}
__device__ __forceinline__ bf16x8 fb_load_frag(const bf16_t* row, int ks, int fg) {
  return __builtin_bit_cast(bf16x8, *(const u32x4*)(row + ks * 16 + fg * 8));
}
// operand whose row 0 (lanes fr == 0) holds 8 values of an fp32 vector in LDS, every other row zero
__device__ __forceinline__ bf16x8 fb_row0_perm(const float* v, int c, int fr, int fg) {      // accumulator row order (dO^T / Q^T fragments)
  bf16x8 r = zero_bf8();
  if (fr == 0) {
    const f32x4 lo = *(const f32x4*)(v + c * 16 + fg * 4), hi = *(const f32x4*)(v + c * 16 + 8 + fg * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { r[e] = (__bf16)lo[e]; r[4 + e] = (__bf16)hi[e]; }
  }
  return r;
}
__device__ __forceinline__ bf16x8 fb_row0_nat(const float* v, int c, int fr, int fg) {       // natural order (K^T fragments)
  bf16x8 r = zero_bf8();
  if (fr == 0) {
    const f32x4 lo = *(const f32x4*)(v + c * 16 + fg * 8), hi = *(const f32x4*)(v + c * 16 + fg * 8 + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { r[e] = (__bf16)lo[e]; r[4 + e] = (__bf16)hi[e]; }
  }
  return r;
}
__device__ __forceinline__ void fb_wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// sum over the 8 lanes that share a row pair while staging (lane & 7 = chunk)
__device__ __forceinline__ float fb_sum8(float v) {
  v += dpp_f<0xB1>(v); v += dpp_f<0x4E>(v); v += dpp_f<0x141>(v);
  return v;
}
// dot of a bf16x8 fragment with 8 consecutive floats of an LDS vector (the address is wave-uniform per half: broadcast)
__device__ __forceinline__ float fb_dot8f(bf16x8 a, const float* v, float acc) {
  const f32x4 lo = *(const f32x4*)v, hi = *(const f32x4*)(v + 4);
#pragma unroll
  for (int e = 0; e < 4; ++e) { acc = fmaf((float)a[e], lo[e], acc); acc = fmaf((float)a[4 + e], hi[e], acc); }
  return acc;
}

#ifdef VL_ATTN_PROF
#define FB_STAMP(p, i) do { if ((p).prof && threadIdx.x == 0) (p).prof[(size_t)item * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FB_STAMP(p, i) do { } while (0)
#endif

// Persistent: one workgroup per CU walks the (batch, head) items blockIdx.x, blockIdx.x + gridDim.x, ...  The 128 KB of
// images leave no LDS for a second resident workgroup or a second stage, so the overlap of one item's memory round trip
// with another's arithmetic comes from L2: as soon as an item is staged the workgroup touches one dword of every 128-byte
// line of the NEXT item's q, k, v, dO, o rows (each row of a head is exactly one line) - those loads fill L2 / the
// Infinity Cache while the tiles are computed, when HBM would otherwise idle (measured: every CU staging at once is
// HBM-bound, 20 k of 53 k cycles per item).
// (kernel arguments re-read from the kernarg segment per item through an opaque pointer: 40 scalars of strides and base
//  pointers otherwise live across the whole item loop and spill)
typedef const __attribute__((address_space(4))) FusedBwdP* FbKernargP;
__device__ __forceinline__ FusedBwdP fb_reload_params() {
  FusedBwdP r;
#if defined(__HIP_DEVICE_COMPILE__)
  FbKernargP kp = (FbKernargP)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(kp));
  // (copied THROUGH the constant address space: cast to a generic pointer the copy became 14 vector loads whose
  //  s_waitcnt vmcnt(0) drained the previous item's stores - and the q / k prefetch - at the top of every item)
  __builtin_memcpy(&r, kp, sizeof(FusedBwdP));
#endif
  return r;
}

__global__ void __launch_bounds__(512, 2) attn_bwd_fused_kernel(const FusedBwdP p0) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sQ = smem;
  unsigned char* sG = smem + FB_IMG;                 // dO
  unsigned char* sK = smem + 2 * FB_IMG;
  unsigned char* sE = smem + 3 * FB_IMG;             // [2][8] dS slots
  float* sNegL = (float*)(sE + 2 * 8 * FB_ESLOT);    // [264]  -lse * log2e   (-inf: padded rows)
  float* sNegD = sNegL + 264;                        // [264]  -delta         (0: padded rows)
  float* sTail = sNegD + 264;                        // [4][64] the lone row of q * qscale (bf16-rounded), dO, k, v as floats + [4][64] bf16
  float* sVec = sTail + 4 * 64 + 128;                      // [8][2][32] per-wave scratch for the single-row MFMA operands
  float* sPart = sVec + 8 * 64;                      // [8][3][64] per-wave partial dq / dk / dv of the lone row
  float* sDummy = sPart + 8 * 192;                   // [64] landing area of the next item's prefetch loads (never read)

  const int nthr = blockDim.x, nt = nthr >> 6;
  const int nitems = p0.B * p0.H;
  // (Round 6: every workgroup's item takes the same time, so all 256 CUs stage at the same instant; starting every second
  //  workgroup 8-32 k cycles late to spread that demand changed nothing - 0.439-0.447 ms with and without at L = 257,
  //  profiles/r06_attn_bwd_stagger.log - and was removed.)
  // (Round 6, late: the staging phase had FIVE serial memory round trips per item - the kernel arguments re-read through vector
  //  loads, then four blocks of loads under lane conditions that hipcc closed with s_waitcnt vmcnt(0) each.  Now: arguments by
  //  scalar loads, every staging load branch-free in one batch.  Fetching the next item's q / k rows into registers under the
  //  tile loop - 32 more live VGPRs - shortened the staging by 2.4 k cycles and lengthened the tiles by 3.2 k: removed;
  //  profiles/r06_v23_attn_bwd_fused_variants.log.)
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
  if (item != (int)blockIdx.x) __syncthreads();      // the previous item's last LDS reads (lone-row finish) are done
  const FusedBwdP p = fb_reload_params();
  // (thread / wave indices re-derived from laundered copies per item: hoisted out of the item loop, the address arithmetic
  //  of every fragment read stays live across the whole body and spills - 119 VGPRs when first tried)
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  asm volatile("" : "+s"(wid));
  const int fr = lane & 31, fg = lane >> 5;
  const int b = item / p.H, h = item - b * p.H;
  const size_t bh = (size_t)item;
  const bf16_t* Qb = p.q.p + b * p.q.sb + h * p.q.sh;
  const bf16_t* Kb = p.k.p + b * p.k.sb + h * p.k.sh;
  const bf16_t* Vb = p.v.p + b * p.v.sb + h * p.v.sh;
  const bf16_t* Gb = p.dO.p + b * p.dO.sb + h * p.dO.sh;
  const bf16_t* Ob = p.o.p + b * p.o.sb + h * p.o.sh;
  const int lm = p.l_main;
  const int kidx = wid * 32 + fr;                    // this lane's key (dK / dV column) and query (dQ column)
  const int krow = kidx < lm ? kidx : lm - 1;

  FB_STAMP(p, 0);
  // ---------------------------------------------------------------- staging: one memory round trip for the workgroup
  // V rows of this lane as MFMA column operands (the K rows come out of the staged image behind the barrier: 32 KB less
  // to fetch per item)
  bf16x8 kf[4], vf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) vf[ks] = fb_load_frag(Vb + (long)krow * p.v.sr, ks, fg);
  // the lone row: lane d of wave 0 fetches element d of q, dO, o, k, v
  // (EVERY wave loads it, from a row that exists with or without a lone row, and uses only wave 0's copy: loads whose
  //  results are consumed under a condition leave hipcc's wait-count analysis with "pending" registers on the other path,
  //  and the first later write to one of them becomes an s_waitcnt that also drains the next item's q / k prefetch)
  const long trow = p.tail ? lm : lm - 1;
  const float tq = bf2f(Qb[trow * p.q.sr + lane]), tg = bf2f(Gb[trow * p.dO.sr + lane]), to = bf2f(Ob[trow * p.o.sr + lane]);
  const float tk = bf2f(Kb[trow * p.k.sr + lane]), tv = bf2f(Vb[trow * p.v.sr + lane]);
  const float tlse = p.lse[bh * p.L + trow];
  {
    // item = 16-byte chunk c of the row pair rp: rows 2rp, 2rp+1 of q, dO, k, o.  nt*128 items, two per thread.
    u32x4 qa[2], qb[2], ga[2], gb[2], ka[2], kb[2], oa[2], ob[2];
    float la[2], lb[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int it = u * nthr + tid;
      const int rp = it >> 3, c = it & 7;
      const int row = 2 * rp;
      const u32x4 z = {0u, 0u, 0u, 0u};
      // branch-free (every lane loads a row that exists, rows past l_main are zeroed by selects): a load under a lane
      // condition makes hipcc close the block with s_waitcnt vmcnt(0) - the four conditional blocks this replaces were
      // four SERIAL memory round trips
      const bool ok0 = row < lm, ok1 = row + 1 < lm;
      const long ra = ok0 ? row : 0, rb = ok1 ? row + 1 : 0;
      const u32x4 g0 = *(const u32x4*)(Gb + ra * p.dO.sr + c * 8), o0 = *(const u32x4*)(Ob + ra * p.o.sr + c * 8);
      const u32x4 g1 = *(const u32x4*)(Gb + rb * p.dO.sr + c * 8), o1 = *(const u32x4*)(Ob + rb * p.o.sr + c * 8);
      const float l0 = p.lse[bh * p.L + ra], l1 = p.lse[bh * p.L + rb];
      const u32x4 q0 = *(const u32x4*)(Qb + ra * p.q.sr + c * 8), k0 = *(const u32x4*)(Kb + ra * p.k.sr + c * 8);
      const u32x4 q1 = *(const u32x4*)(Qb + rb * p.q.sr + c * 8), k1 = *(const u32x4*)(Kb + rb * p.k.sr + c * 8);
      ga[u] = ok0 ? g0 : z; oa[u] = ok0 ? o0 : z; la[u] = ok0 ? l0 : INFINITY; qa[u] = ok0 ? q0 : z; ka[u] = ok0 ? k0 : z;
      gb[u] = ok1 ? g1 : z; ob[u] = ok1 ? o1 : z; lb[u] = ok1 ? l1 : INFINITY; qb[u] = ok1 ? q1 : z; kb[u] = ok1 ? k1 : z;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int it = u * nthr + tid;
      const int rp = it >> 3, c = it & 7;
      const int r0 = 2 * rp, r1 = r0 + 1;
      const int o0 = r0 * FB_RB + ((c ^ fb_swz(r0)) << 4), o1 = r1 * FB_RB + ((c ^ fb_swz(r1)) << 4);
      *(u32x4*)(sQ + o0) = p.qscale != 1.0f ? scale_bf16x8(qa[u], p.qscale) : qa[u];
      *(u32x4*)(sQ + o1) = p.qscale != 1.0f ? scale_bf16x8(qb[u], p.qscale) : qb[u];
      *(u32x4*)(sG + o0) = ga[u]; *(u32x4*)(sG + o1) = gb[u];
      *(u32x4*)(sK + o0) = ka[u]; *(u32x4*)(sK + o1) = kb[u];
      const float d0 = fb_sum8(dot8(__builtin_bit_cast(bf16x8, ga[u]), __builtin_bit_cast(bf16x8, oa[u]), 0.f));
      const float d1 = fb_sum8(dot8(__builtin_bit_cast(bf16x8, gb[u]), __builtin_bit_cast(bf16x8, ob[u]), 0.f));
      float nl0 = -la[u] * FB_LOG2E, nl1 = -lb[u] * FB_LOG2E;             // by every lane, outside the lane condition (see above)
      asm volatile("" : "+v"(nl0), "+v"(nl1));
      if (c == 0) {
        sNegL[r0] = nl0; sNegL[r1] = nl1;                                    // padded rows: -inf
        sNegD[r0] = -d0; sNegD[r1] = -d1;                                    // padded rows: 0 (zero-filled operands)
      }
    }
  }
  if (p.tail && wid == 0) {
    sTail[lane] = bf2f(f2bf(tq * p.qscale));       // bf16-rounded like the staged q
    sTail[64 + lane] = tg; sTail[128 + lane] = tk; sTail[192 + lane] = tv;
    bf16_t* tb16 = (bf16_t*)(sTail + 256);
    tb16[lane] = f2bf(tq * p.qscale); tb16[64 + lane] = f2bf(tg); tb16[128 + lane] = f2bf(tk); tb16[192 + lane] = f2bf(tv);
    const float dT = wave_sum_dpp(tg * to);
    if (lane == 0) { sNegL[lm] = -tlse * FB_LOG2E; sNegD[lm] = -dT; }
  }
  FB_STAMP(p, 1);
  __syncthreads();
  FB_STAMP(p, 2);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) kf[ks] = fb_rows(sK, krow, ks, fg);
  if (item + (int)gridDim.x < nitems && tid < p.L) {
    // touch the next item's lines (fire and forget: the values are never read; the loads retire under the tile loop)
    const int nb = (item + gridDim.x) / p.H, nh = (item + gridDim.x) - nb * p.H;
    const bf16_t* a0 = p.q.p + nb * p.q.sb + nh * p.q.sh + (long)tid * p.q.sr;
    const bf16_t* a1 = p.k.p + nb * p.k.sb + nh * p.k.sh + (long)tid * p.k.sr;
    const bf16_t* a2 = p.v.p + nb * p.v.sb + nh * p.v.sh + (long)tid * p.v.sr;
    const bf16_t* a3 = p.dO.p + nb * p.dO.sb + nh * p.dO.sh + (long)tid * p.dO.sr;
    const bf16_t* a4 = p.o.p + nb * p.o.sb + nh * p.o.sh + (long)tid * p.o.sr;
    // LDS-DMA into a 256-byte dummy area: no destination registers (a plain load's VGPR could be re-allocated before the
    // data arrives), and, written as inline assembly, no compiler-inserted vmcnt(0) in front of the tile loop's LDS reads
    unsigned m0s;
    const unsigned dummy = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)sDummy;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\t"
                 "global_load_lds_dword %1, off\n\tglobal_load_lds_dword %2, off\n\tglobal_load_lds_dword %3, off\n\t"
                 "global_load_lds_dword %4, off\n\tglobal_load_lds_dword %5, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(m0s) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "s"(dummy) : "memory");
  }

  // ---------------------------------------------------------------- tiles
  // Fragment addresses.  The instruction mix of a step is 20 MFMAs against ~70 VALU instructions only if the LDS addresses cost
  // (almost) nothing: left to the compiler, row * 128 + ((chunk ^ swz(row)) << 4) for 34 reads per step was ~180 VALU
  // instructions and the loop ran VALU-bound (2.7 k cycles per step against 1.3 k of MFMA time per SIMD).  Every address
  // below is ONE lane constant (byte offset inside a 32-row tile) + the tile base, and its variants are XORs with small
  // constants (a tile base is a multiple of 32 rows, so the swizzle does not depend on it):
  //   row fragment (row fr, d-slice ks)             aRows ^ ks*32
  //   dO^T / Q^T (rows 16c + 4fg + 8hi + (li>>2))   (aTr ^ hi*32 ^ t2*64) + c*2048 + hi*1024
  //   K^T        (rows 16c + 8fg + 4hi + (li>>2))   (aTrK ^ hi*16 ^ t2*64) + c*2048 + hi*512
  //   dS slot read (keys 16c + 8fg + 4hi + (li>>2)) (aErd ^ hi*32) + c*1024 + hi*256;   write (key fr, group g): aEwr ^ g*16
  const int li = lane & 15, cbl = (lane >> 4) & 1, lb3 = (li >> 3) & 1;
  const unsigned cch = (unsigned)(cbl * 2 + ((li >> 1) & 1)), hb = (unsigned)((li & 1) << 3);
  const unsigned aRows = (unsigned)(fr * 128 + ((fg ^ fb_swz(fr)) << 4));
  const unsigned aTr = (unsigned)((fg * 4 + (li >> 2)) * 128) + ((cch ^ (unsigned)((lb3 << 2) | fg)) << 4) + hb;
  const unsigned aTrK = (unsigned)((fg * 8 + (li >> 2)) * 128) + ((cch ^ (unsigned)((lb3 << 2) | (fg << 1))) << 4) + hb;
  const unsigned aErd = (unsigned)((fg * 8 + (li >> 2)) * 64) + ((cch ^ (unsigned)lb3) << 4) + hb;
  const unsigned aEwr = (unsigned)(fr * 64 + fg * 8 + (((fr >> 1) & 3) << 4));
  auto ld16 = [&](unsigned off) { return *(const bf16x8*)(smem + off); };
  auto ldtr = [&](unsigned lo, unsigned hi) {
    FbFrag2 f;
    f.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((fb_lds_s16x4)(smem + lo));
    f.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((fb_lds_s16x4)(smem + hi));
    return __builtin_bit_cast(bf16x8, f);
  };
  // transposed fragment of image `img` (byte offset of sQ / sG): tile base tb (bytes), 16-row slice c, 32-column half t2
  auto trQG = [&](unsigned img, unsigned tb, int c, int t2) {
    const unsigned a = (aTr + tb) ^ (unsigned)(t2 * 64);
    return ldtr(img + a + c * 2048, img + (a ^ 32u) + c * 2048 + 1024);
  };
  auto trK = [&](unsigned tb, int c, int t2) {
    const unsigned a = (aTrK + tb) ^ (unsigned)(t2 * 64);
    return ldtr(2 * FB_IMG + a + c * 2048, 2 * FB_IMG + (a ^ 16u) + c * 2048 + 512);
  };

  f32x16 dk[2], dv[2], dq[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) { dk[t] = zero16(); dv[t] = zero16(); dq[t] = zero16(); }
  float* myVec = sVec + wid * 64;
  float* myPart = sPart + wid * 192;
  const bool key_ok = kidx < lm;
  const bool ragged = (lm & 31) != 0 && wid == nt - 1;        // the last key tile has padded keys (wave-uniform)

  if (p.tail) {
    // ================= the lone row / column (class token of a 257-token sequence), before the tile loop =================
    // Scores through the matrix pipe too: the lone key is column 0 of a column operand that is zero elsewhere (its 32 scores
    // against this wave's query tile land in lanes 0 and 32), the lone query row 0 of a row operand (its scores against this
    // wave's keys land in slot 0 of lanes 0-31).  (First version: VALU dot products, 7 k of 50 k cycles per item.)
    const unsigned tb = (unsigned)(wid * 4096);
    const bf16_t* sTailB = (const bf16_t*)(sTail + 256);       // the four lone rows as bf16 (q2, dO, k, v)
    {
      f32x16 sc, dc;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const f32x4 l4 = *(const f32x4*)(sNegL + wid * 32 + qd * 8 + fg * 4);
        const f32x4 d4 = *(const f32x4*)(sNegD + wid * 32 + qd * 8 + fg * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { sc[qd * 4 + e] = l4[e]; dc[qd * 4 + e] = d4[e]; }
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 kT = fr == 0 ? *(const bf16x8*)(sTailB + 128 + ks * 16 + fg * 8) : zero_bf8();
        const bf16x8 vT = fr == 0 ? *(const bf16x8*)(sTailB + 192 + ks * 16 + fg * 8) : zero_bf8();
        sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld16((aRows + tb) ^ (unsigned)(ks * 32)), kT, sc, 0, 0, 0);
        dc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld16(FB_IMG + ((aRows + tb) ^ (unsigned)(ks * 32))), vT, dc, 0, 0, 0);
      }
      if (fr == 0) {       // column 0: queries (r&3) + 8*(r>>2) + 4*fg of the tile
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 pq, dq4;
#pragma unroll
          for (int e = 0; e < 4; ++e) { pq[e] = __builtin_amdgcn_exp2f(sc[g * 4 + e]); dq4[e] = pq[e] * dc[g * 4 + e]; }
          *(f32x4*)(myVec + g * 8 + fg * 4) = pq;
          *(f32x4*)(myVec + 32 + g * 8 + fg * 4) = dq4;
        }
      }
    }
    fb_wave_lds_sync();
    {
      // dQ[q, :] += dS[q, lm] * K[lm, :]
      const float dsc = myVec[32 + fr];
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 k4 = *(const f32x4*)(sTail + 128 + t2 * 32 + g * 8 + fg * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) dq[t2][g * 4 + e] = fmaf(dsc, k4[e], dq[t2][g * 4 + e]);
        }
    }
    // dV[lm, :] += sum_q P[q, lm] dO[q, :],  dK[lm, :] += sum_q dS[q, lm] Q2[q, :]: single-row MFMAs
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      f32x16 av = zero16(), ak = zero16();
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        av = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb_row0_perm(myVec, c, fr, fg), trQG(FB_IMG, tb, c, t2), av, 0, 0, 0);
        ak = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb_row0_perm(myVec + 32, c, fr, fg), trQG(0, tb, c, t2), ak, 0, 0, 0);
      }
      if (fg == 0) { myPart[128 + t2 * 32 + fr] = av[0]; myPart[64 + t2 * 32 + fr] = ak[0]; }
    }
    // ---- the lone query (row lm) against this wave's key tile: row 0 of the products, lane = key fr ----
    float pr, dsr;
    {
      f32x16 sr = zero16(), dr = zero16();
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 qT = fr == 0 ? *(const bf16x8*)(sTailB + ks * 16 + fg * 8) : zero_bf8();
        const bf16x8 gT = fr == 0 ? *(const bf16x8*)(sTailB + 64 + ks * 16 + fg * 8) : zero_bf8();
        sr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qT, kf[ks], sr, 0, 0, 0);
        dr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gT, vf[ks], dr, 0, 0, 0);
      }
      const float p0v = fg == 0 ? __builtin_amdgcn_exp2f(sr[0] + sNegL[lm]) : 0.f;
      const float d0v = fg == 0 ? p0v * (dr[0] + sNegD[lm]) : 0.f;
      pr = xhalf_sum(p0v); dsr = xhalf_sum(d0v);                // both halves of the wave hold their key's values
    }
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 g4 = *(const f32x4*)(sTail + 64 + t2 * 32 + g * 8 + fg * 4);
        const f32x4 q4 = *(const f32x4*)(sTail + t2 * 32 + g * 8 + fg * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          dv[t2][g * 4 + e] = fmaf(pr, g4[e], dv[t2][g * 4 + e]);
          dk[t2][g * 4 + e] = fmaf(dsr, q4[e], dk[t2][g * 4 + e]);
        }
      }
    // dQ[lm, :] += sum over this tile's keys dS[lm, key] K[key, :]
    fb_wave_lds_sync();
    if (fg == 0) myVec[fr] = dsr;
    fb_wave_lds_sync();
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      f32x16 aq = zero16();
#pragma unroll
      for (int c = 0; c < 2; ++c)
        aq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb_row0_nat(myVec, c, fr, fg), trK(tb, c, t2), aq, 0, 0, 0);
      if (fg == 0) myPart[t2 * 32 + fr] = aq[0];
    }
  }

  FB_STAMP(p, 3);
  int qi = wid, kj = wid;                            // (wid + t) mod nt, (wid - t) mod nt
  for (int t = 0; t < nt; ++t) {
    const unsigned tb = (unsigned)(qi * 4096);       // byte offset of the step's 32 query rows in the images
    // rows of the accumulators = queries (r&3) + 8*(r>>2) + 4*fg of the tile; column = this lane's key
    f32x16 s, dp;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const f32x4 l4 = *(const f32x4*)(sNegL + qi * 32 + qd * 8 + fg * 4);
      const f32x4 d4 = *(const f32x4*)(sNegD + qi * 32 + qd * 8 + fg * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { s[qd * 4 + e] = l4[e]; dp[qd * 4 + e] = d4[e]; }
    }
    {
      const unsigned aR = aRows + tb;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld16(aR ^ (unsigned)(ks * 32)), kf[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld16(FB_IMG + (aR ^ (unsigned)(ks * 32))), vf[ks], dp, 0, 0, 0);
      }
    }
    // transposed fragments (dO^T, Q^T) of the first 16-query slice, requested ahead of the exponentials
    bf16x8 gt0[2], qt0[2];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) { gt0[t2] = trQG(FB_IMG, tb, 0, t2); qt0[t2] = trQG(0, tb, 0, t2); }
    if (ragged) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = key_ok ? s[r] : -INFINITY;
    }
    float pv[16], ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { pv[r] = __builtin_amdgcn_exp2f(s[r]); ds[r] = pv[r] * dp[r]; }     // (padded query rows carry -inf: p = 0)
    const bf16x8 pf0 = pack8(pv), df0 = pack8(ds), pf1 = pack8(pv + 8), df1 = pack8(ds + 8);
    // dS tile -> slot [t & 1][wid] as [key][query] bf16 for the owner of these queries (the packed MFMA operands are the rows)
    {
      const unsigned sb = (unsigned)(3 * FB_IMG + ((t & 1) * 8 + wid) * FB_ESLOT) + aEwr;
      const u32x4 w0 = __builtin_bit_cast(u32x4, df0), w1 = __builtin_bit_cast(u32x4, df1);
      *(u32x2*)(smem + sb) = u32x2{w0[0], w0[1]};
      *(u32x2*)(smem + (sb ^ 16u)) = u32x2{w0[2], w0[3]};
      *(u32x2*)(smem + (sb ^ 32u)) = u32x2{w1[0], w1[1]};
      *(u32x2*)(smem + (sb ^ 48u)) = u32x2{w1[2], w1[3]};
    }
    bf16x8 gt1[2], qt1[2];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) { gt1[t2] = trQG(FB_IMG, tb, 1, t2); qt1[t2] = trQG(0, tb, 1, t2); }
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      dv[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gt0[t2], pf0, dv[t2], 0, 0, 0);
      dk[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt0[t2], df0, dk[t2], 0, 0, 0);
    }
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      dv[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gt1[t2], pf1, dv[t2], 0, 0, 0);
      dk[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt1[t2], df1, dk[t2], 0, 0, 0);
    }
    __syncthreads();                                 // every dS tile of this step is in its slot
    // ---- dQ of this wave's queries: the tile produced by the owner of keys (wid - t) mod nt ----
    {
      const unsigned sb = (unsigned)(3 * FB_IMG + ((t & 1) * 8 + kj) * FB_ESLOT) + aErd;
      const unsigned kb = (unsigned)(kj * 4096);
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const bf16x8 dst = ldtr(sb + c * 1024, (sb ^ 32u) + c * 1024 + 256);
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) dq[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trK(kb, c, t2), dst, dq[t2], 0, 0, 0);
      }
    }
    qi = qi + 1 == nt ? 0 : qi + 1;
    kj = kj == 0 ? nt - 1 : kj - 1;
  }

  FB_STAMP(p, 4);
  // ---------------------------------------------------------------- results
  const float ln2 = 0.6931471805599453f;
  if (p.tail && wid == 0) {
    // the lone row: the waves' partial sums were complete before the first barrier of the tile loop
    const int d = lane;
    float aq = 0.f, ak = 0.f, av = 0.f;
    for (int w = 0; w < nt; ++w) { aq += sPart[w * 192 + d]; ak += sPart[w * 192 + 64 + d]; av += sPart[w * 192 + 128 + d]; }
    // ... and the lone row against itself
    const float q2 = sTail[d], gT = sTail[64 + d], kT = sTail[128 + d], vT = sTail[192 + d];
    const float sT = wave_sum_dpp(q2 * kT), dT = wave_sum_dpp(gT * vT);
    const float pT = __builtin_amdgcn_exp2f(sT + sNegL[lm]);
    const float dsT = pT * (dT + sNegD[lm]);
    aq = fmaf(dsT, kT, aq); ak = fmaf(dsT, q2, ak); av = fmaf(pT, gT, av);
    const size_t row = (size_t)b * p.L + lm;
    p.dq[row * p.ld_dq + h * 64 + d] = f2bf(aq * p.scale);
    p.dk[row * p.ld_dkv + h * 64 + d] = f2bf(ak * ln2);
    p.dv[row * p.ld_dkv + h * 64 + d] = f2bf(av);
  }
  FB_STAMP(p, 5);
  const size_t orow = (size_t)b * p.L + krow;
  store_rows_t<2>(dk, ln2, p.dk + orow * p.ld_dkv + h * 64, fg, key_ok, 8);
  store_rows_t<2>(dv, 1.f, p.dv + orow * p.ld_dkv + h * 64, fg, key_ok, 8);
  store_rows_t<2>(dq, p.scale, p.dq + orow * p.ld_dq + h * 64, fg, key_ok, 8);
  FB_STAMP(p, 6);
  }   // items
}

constexpr size_t FB_LDS = (size_t)3 * FB_IMG + 2 * 8 * FB_ESLOT + (264 + 264 + 256 + 128 + 512 + 1536 + 64) * sizeof(float);

}  // namespace

extern "C" int vl_set_error(const char* msg);

// 1 when vl_attn_bwd_fused_bf16 takes the problem (the Python wrapper and the tests ask the library instead of restating it)
extern "C" int vl_attn_bwd_fused_supported(int Lq, int Lk, int dh, int causal) {
  if (dh != 64 || Lq != Lk || causal || Lq <= 0) return 0;
  const bool tail = (Lq % 32 == 1) && Lq > 32;
  const int lm = tail ? Lq - 1 : Lq;
  return lm <= 256 ? 1 : 0;
}

extern "C" int vl_attn_bwd_fused_bf16(const void* q, const void* k, const void* v, const void* dO, const void* o,
                                      const long* strides, const float* lse, void* dq, void* dk, void* dv, long ld_dq,
                                      long ld_dkv, int B, int H, int L, int dh, float qscale, float scale, hipStream_t stream) {
  if (B <= 0 || H <= 0 || L <= 0) return vl_set_error("vl_attn_bwd_fused_bf16: empty problem");
  if (!vl_attn_bwd_fused_supported(L, L, dh, 0))
    return vl_set_error("vl_attn_bwd_fused_bf16: needs head dim 64 and L <= 256 or L = 32 m + 1 <= 257 (use vl_attn_bwd_bf16)");
  if (!strides || !lse) return vl_set_error("vl_attn_bwd_fused_bf16: strides and lse are required");
  for (int i = 0; i < 15; ++i)
    if (strides[i] & 7) return vl_set_error("vl_attn_bwd_fused_bf16: operand strides must be multiples of 8 elements (16-byte rows)");
  if ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v) | ((uintptr_t)dO) | ((uintptr_t)o)) & 15)
    return vl_set_error("vl_attn_bwd_fused_bf16: operands must be 16-byte aligned");
  if ((((uintptr_t)dq) | ((uintptr_t)dk) | ((uintptr_t)dv)) & 15 || (ld_dq & 7) || (ld_dkv & 7))
    return vl_set_error("vl_attn_bwd_fused_bf16: gradient destinations must be 16-byte aligned with row strides multiple of 8");
  static const hipError_t attr = hipFuncSetAttribute((const void*)attn_bwd_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     (int)FB_LDS);
  if (attr != hipSuccess) return vl_set_error(hipGetErrorString(attr));
  const long* s = strides;
  const bool tail = (L % 32 == 1) && L > 32;
  const int lm = tail ? L - 1 : L;
  FusedBwdP p{TV{(const bf16_t*)q, s[0], s[1], s[2]},   TV{(const bf16_t*)k, s[3], s[4], s[5]},
              TV{(const bf16_t*)v, s[6], s[7], s[8]},   TV{(const bf16_t*)dO, s[9], s[10], s[11]},
              TV{(const bf16_t*)o, s[12], s[13], s[14]}, lse, (bf16_t*)dq, (bf16_t*)dk, (bf16_t*)dv, ld_dq, ld_dkv,
              B, H, L, qscale, scale, lm, tail ? 1 : 0
#ifdef VL_ATTN_PROF
              , vl_attn_prof_buf
#endif
  };
  const int nt = (lm + 31) / 32;
  int dev = 0, ncu = 0;
  static const int cus = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && ncu > 0) ? ncu : 256;
  const int items = B * H;
  hipLaunchKernelGGL(attn_bwd_fused_kernel, dim3(items < cus ? items : cus), dim3(nt * 64), FB_LDS, stream, p);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : vl_set_error(hipGetErrorString(e));
}
