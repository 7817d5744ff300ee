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
#include "vl_common.h"
#include "vitlens_hip.h"

namespace {

enum Epi : int {
  EPI_BF16 = 0,   // out bf16 = act(acc*alpha + bias)
  EPI_F32 = 1,    // out f32  = acc*alpha + bias
  EPI_RES_F32 = 2,   // out f32  = res f32 + acc + bias      (in-place allowed)
  EPI_RES_BF16 = 3,  // out bf16 = res bf16 + acc + bias
  EPI_QKV = 4,    // scatter to q[B,H,L,dh], k[B,H,L,dh], vt[B,H,dh,Lp]
  EPI_GEGLU = 5,  // rows interleaved (a_j, gate_j): out bf16[M, N/2] = a * gelu(gate)
};

struct GemmP {
  const bf16_t* A;   // [M, K]
  const bf16_t* W;   // [N, K]
  const float* bias; // [N] or null
  void* out;
  const void* res;
  int M, N, K;
  int lda, ldw, ldo; // row strides in elements
  float alpha;
  int act;           // 0 none, 1 gelu(erf)
  // QKV scatter
  bf16_t *q, *k, *vt;
  int L, H, dh, Lp;
  float qscale;
};

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

  const int nk = p.K >> 6;

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
    gA[i] = p.A + (size_t)gm * p.lda + (pch ^ ((row >> 1) & 7)) * 8;
  }
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int row = (i * NW + wid) * 8 + srow;
    int gn = n0 + row; gn = gn < p.N ? gn : p.N - 1;
    gB[i] = p.W + (size_t)gn * p.ldw + (pch ^ ((row >> 1) & 7)) * 8;
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

  // ---- epilogue: lane owns row m = ..+fr, columns n = nb + 8*q + 4*fg + {0..3} ---------------
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = m0 + wave_m * WTM + i * 32 + fr;
    if (m >= p.M) continue;
    [[maybe_unused]] int qb = 0, ql = 0;
    if constexpr (EPI == EPI_QKV) { qb = m / p.L; ql = m - qb * p.L; }
#pragma unroll
    for (int j = 0; j < NTL; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + wave_n * WTN + j * 32 + q * 8 + fg * 4;
        if (n >= p.N) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][q * 4 + e] * p.alpha;
        if (p.bias) {
          const f32x4 b = *(const f32x4*)(p.bias + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += b[e];
        }
        if constexpr (EPI == EPI_BF16) {
          if (p.act == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
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
          const u32x2 r = *(const u32x2*)((const bf16_t*)p.res + (size_t)m * p.ldo + n);
          v[0] += bf2f((bf16_t)(r[0] & 0xffff)); v[1] += bf2f((bf16_t)(r[0] >> 16));
          v[2] += bf2f((bf16_t)(r[1] & 0xffff)); v[3] += bf2f((bf16_t)(r[1] >> 16));
          u32x2 o; o[0] = pack2bf(v[0], v[1]); o[1] = pack2bf(v[2], v[3]);
          *(u32x2*)((bf16_t*)p.out + (size_t)m * p.ldo + n) = o;
        } else if constexpr (EPI == EPI_QKV) {
          // head-dim % 8 == 0: (which, head) are wave-uniform for the 8-column group -> SALU divides
          const int nu = n0 + wave_n * WTN + j * 32 + q * 8;
          const int D = p.H * p.dh;
          const int which = nu / D;
          const int c = nu - which * D;
          const int h = c / p.dh, d = c - h * p.dh + fg * 4;
          const size_t bh = (size_t)qb * p.H + h;
          if (which == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= p.qscale;
            u32x2 o; o[0] = pack2bf(v[0], v[1]); o[1] = pack2bf(v[2], v[3]);
            *(u32x2*)(p.q + (bh * p.L + ql) * p.dh + d) = o;
          } else if (which == 1) {
            u32x2 o; o[0] = pack2bf(v[0], v[1]); o[1] = pack2bf(v[2], v[3]);
            *(u32x2*)(p.k + (bh * p.L + ql) * p.dh + d) = o;
          } else {
            bf16_t* dst = p.vt + (bh * p.dh + d) * p.Lp + ql;
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[(size_t)e * p.Lp] = f2bf(v[e]);
          }
        } else if constexpr (EPI == EPI_GEGLU) {
          // interleaved rows: (a_j, gate_j, a_j+1, gate_j+1)
          const float o0 = v[0] * gelu_erf(v[1]);
          const float o1 = v[2] * gelu_erf(v[3]);
          *(unsigned int*)((bf16_t*)p.out + (size_t)m * p.ldo + (n >> 1)) = pack2bf(o0, o1);
        }
      }
    }
  }
}

template <int BM, int BN, int WM, int WN, int EPI, bool DMA>
hipError_t launch(const GemmP& p, hipStream_t s) {
  using S = Smem<BM, BN>;
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  auto kern = gemm_nt_kernel<BM, BN, WM, WN, EPI, DMA>;
  constexpr int smem = 2 * S::STAGE;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(WM * WN * 64), smem, s, p);
  return hipGetLastError();
}

template <int EPI>
hipError_t dispatch(const GemmP& p, int cfg, hipStream_t s) {
  // cfg bit0: 0 = 256x256 tile (8 waves), 1 = 128x128 tile (4 waves); bit1: 1 = register staging
  switch (cfg & 3) {
    case 0: return launch<256, 256, 2, 4, EPI, true>(p, s);
    case 1: return launch<128, 128, 2, 2, EPI, true>(p, s);
    case 2: return launch<256, 256, 2, 4, EPI, false>(p, s);
    default: return launch<128, 128, 2, 2, EPI, false>(p, s);
  }
}

}  // namespace

extern "C" int vl_set_error(const char* msg);

static int auto_cfg(int M, int N, int cfg) {
  if (cfg >= 0) return cfg;
  // big tiles only when they still fill the chip a few times over
  const long big_tiles = (long)((M + 255) / 256) * ((N + 255) / 256);
  return big_tiles >= 512 ? 0 : 1;
}

#define VL_CHECK_ARG(c, msg) do { if (!(c)) return vl_set_error(msg); } while (0)

extern "C" int vl_gemm_bf16(const void* A, const void* W, const float* bias, void* out, const void* res,
                            int M, int N, int K, int lda, int ldw, int ldo, float alpha, int epi, int act,
                            int cfg, hipStream_t stream) {
  VL_CHECK_ARG(M > 0 && N > 0 && K > 0, "vl_gemm_bf16: empty problem");
  VL_CHECK_ARG((K & 63) == 0, "vl_gemm_bf16: K must be a multiple of 64");
  VL_CHECK_ARG((N & 3) == 0, "vl_gemm_bf16: N must be a multiple of 4");
  VL_CHECK_ARG((lda & 7) == 0 && (ldw & 7) == 0, "vl_gemm_bf16: lda/ldw must be multiples of 8");
  VL_CHECK_ARG((ldo & 3) == 0, "vl_gemm_bf16: ldo must be a multiple of 4");
  GemmP p{};
  p.A = (const bf16_t*)A; p.W = (const bf16_t*)W; p.bias = bias; p.out = out; p.res = res;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldo = ldo; p.alpha = alpha; p.act = act;
  cfg = auto_cfg(M, N, cfg);
  hipError_t e;
  switch (epi) {
    case VL_EPI_BF16: e = dispatch<EPI_BF16>(p, cfg, stream); break;
    case VL_EPI_F32: e = dispatch<EPI_F32>(p, cfg, stream); break;
    case VL_EPI_RES_F32: VL_CHECK_ARG(res, "vl_gemm_bf16: residual missing"); e = dispatch<EPI_RES_F32>(p, cfg, stream); break;
    case VL_EPI_RES_BF16: VL_CHECK_ARG(res, "vl_gemm_bf16: residual missing"); e = dispatch<EPI_RES_BF16>(p, cfg, stream); break;
    case VL_EPI_GEGLU: e = dispatch<EPI_GEGLU>(p, cfg, stream); break;
    default: return vl_set_error("vl_gemm_bf16: unknown epilogue");
  }
  if (e != hipSuccess) return vl_set_error(hipGetErrorString(e));
  return 0;
}

extern "C" int vl_gemm_qkv_bf16(const void* A, const void* W, const float* bias, void* q, void* k, void* vt,
                                int B, int L, int H, int dh, int Lp, int K, int lda, float qscale, int cfg,
                                hipStream_t stream) {
  VL_CHECK_ARG(B > 0 && L > 0 && H > 0, "vl_gemm_qkv_bf16: empty problem");
  VL_CHECK_ARG((K & 63) == 0, "vl_gemm_qkv_bf16: K must be a multiple of 64");
  VL_CHECK_ARG((dh & 7) == 0, "vl_gemm_qkv_bf16: head dim must be a multiple of 8");
  VL_CHECK_ARG(Lp >= L, "vl_gemm_qkv_bf16: Lp < L");
  GemmP p{};
  p.A = (const bf16_t*)A; p.W = (const bf16_t*)W; p.bias = bias;
  p.M = B * L; p.N = 3 * H * dh; p.K = K; p.lda = lda; p.ldw = K; p.ldo = 0; p.alpha = 1.f;
  p.q = (bf16_t*)q; p.k = (bf16_t*)k; p.vt = (bf16_t*)vt; p.L = L; p.H = H; p.dh = dh; p.Lp = Lp; p.qscale = qscale;
  cfg = auto_cfg(p.M, p.N, cfg);
  hipError_t e = dispatch<EPI_QKV>(p, cfg, stream);
  if (e != hipSuccess) return vl_set_error(hipGetErrorString(e));
  return 0;
}
