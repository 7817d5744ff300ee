// Shared definitions of the NT GEMM kernels (vl_gemm.hip: tile / tail / split-K kernels and the dispatcher, vl_gemm_park.hip:
// the persistent 256x256 kernel, vl_gemm_pp.hip: 256x128 tiles, vl_gemm_tn.hip: token-major weight gradients).
#pragma once
#include "vl_common.h"
#include "vitlens_hip.h"

namespace {

enum Epi : int {
  EPI_BF16 = 0,   // out bf16 = act(acc*alpha + bias)
  EPI_F32 = 1,    // out f32  = acc*alpha + bias
  EPI_RES_F32 = 2,   // out f32  = res f32 + acc + bias      (in-place allowed)
  EPI_RES_BF16 = 3,  // out bf16 = res bf16 + acc + bias
  // (4 was the head-scatter epilogue of rounds 1-2a: the attention kernels now read the packed projection in place)
  EPI_GEGLU = 5,  // rows interleaved (a_j, gate_j): out bf16[M, N/2] = a * gelu(gate)
  EPI_DGELU = 6,  // out bf16 = acc * gelu'(aux[m,n])   (backward through GELU fused into the dX GEMM)
  EPI_DGEGLU = 7, // acc = dy[M,N]; res = h[M,2N] interleaved (a,g): out[M,2N] = (dy*gelu(g), dy*a*gelu'(g)) interleaved
};

struct GemmP {
  const bf16_t* A;   // [M, K]
  const bf16_t* W;   // [N, K]
  const float* bias; // [N] or null
  void* out;
  const void* res;   // residual (EPI_RES_*) or pre-activation u (EPI_DGELU), same shape/stride as out
  void* out2;        // EPI_BF16 + act: optional copy of the PRE-activation values (saved for backward)
  int M, N, K;
  int lda, ldw, ldo; // row strides in elements
  float alpha;
  int act;           // 0 none, 1 gelu(erf), 2 relu
  int res_div;       // residual row = m / res_div (>=1): broadcast one row over a group of res_div rows
  int m_off;         // absolute row of local row 0 of a row-split launch (EPI_RES_BF16 indexes its residual with it)
  // split-K (gemm_nt_kernel only): blockIdx.y owns k-slabs [y*ksplit_len, (y+1)*ksplit_len) and writes its
  // partial product to out + y*split_stride (f32 elements); ksplit_len == 0 -> whole K, no offset
  int ksplit_len;
  long split_stride;
  int f16;           // operands A, W (and a 16-bit output) are IEEE half instead of bf16 (persistent 256x256 kernel only: vl_gemm_f16)
  // LayerNorm folded into the GEMM (round 4, persistent kernel only; vl_gemm_lnfold_bf16 / vl_gemm_res_rowstats_bf16):
  //   consumer: A = the RAW rows x, W = bf16(W * gamma), bias = b + W beta, ln_c[n] = sum_k W'[n, k]:
  //             out = act(rstd_m * (acc - mean_m * c_n) + bias_n)  ==  act(LN(x) W^T + b)
  //   producer: EPI_RES_BF16 also writes, per row and 64-column slice, (sum, sum of squares) of the bf16 values it stores
  const float* ln_mean;   // [M] of the rows of A (consumer)
  const float* ln_rstd;   // [M]
  const float* ln_c;      // [N]
  float* row_part;        // [M][N / 64][2] partial row statistics of the output (producer)
};

// acc * alpha + bias as four plain v_fma_f32.  Left to the compiler, the vector expression becomes v_pk_fma_f32, and in
// one of the two operand orders hipcc picks (accumulator pair as src0, the (ldo, alpha) pair as src1 with op_sel:[0,1,0])
// the LOW half of the result came out as if the accumulator were zero in lanes 48-63, on the last row block of a tile, a
// few times per launch (100-2000 of 268 M elements; the other order, which round 2 happened to get, never did).  Found in
// round 3 by tests/test_hip_gemm_park.py after an unrelated refactor flipped the operand order; evidence and the bisection
// in profiles/r03_pk_fma_fault.log.  Plain FMAs were clean in every run and cost two more VALU issues per eight values.
// The FMAs below are inline assembly, which LLVM's hazard recognizer does not see: the wait states a VALU read of a matrix-pipe
// result needs (up to 18 for a 16-pass MFMA) must not depend on how many instructions the compiler happens to place between
// the last MFMA of a tile and the first scale_bias.  Every epilogue that uses scale_bias calls this once, first: 32 idle
// issue slots per 256x256 tile.
__device__ __forceinline__ void mfma_results_settled() {
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
}

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {      // cross-lane move inside a row of 16 lanes (DPP control CTRL)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

__device__ __forceinline__ f32x4 scale_bias(f32x4 v, float alpha, f32x4 b) {
  f32x4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float t;
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(v[e]), "v"(alpha), "v"(b[e]));
    r[e] = t;
  }
  return r;
}

template <int BM, int BN>
struct Smem {
  static constexpr int A_BYTES = BM * 128;
  static constexpr int B_BYTES = BN * 128;
  static constexpr int STAGE = A_BYTES + B_BYTES;
};

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  // bijective "each XCD owns a contiguous chunk" remap (dispatch puts block b on XCD b % 8)
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// The epilogue parameters (pointers, strides, scatter geometry) are re-read from the kernarg segment through an
// opaque pointer at the moment a tile is stored.  Without this the compiler keeps ~40 scalars live across
// the whole k-loop and spills SGPRs to scratch INSIDE it (measured on the round-1 head-scatter GEMM: 0.60 -> 2.49 ms).
typedef const __attribute__((address_space(4))) GemmP* KernargP;
__device__ __forceinline__ GemmP reload_params() {
  GemmP r;
#if defined(__HIP_DEVICE_COMPILE__)
  KernargP kp = (KernargP)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(kp));
  r = *kp;      // a copy THROUGH the constant address space: scalar loads (the memcpy through a generic pointer of rounds 1-3
                // became per-lane global_load_dwordx4 + readfirstlane: one more vector-memory round trip at every epilogue start)
#endif
  return r;
}


}  // namespace
