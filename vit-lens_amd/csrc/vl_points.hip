// Point-cloud grouping for the PointBERT tokenizer of the 3D Lens (latency-bound integer/index work).
//
//   vl_fps        farthest-point sampling, bit-exact restatement of misc.fps
//                 (open_clip/modal_3d/models/pointbert/misc.py:48-68): one workgroup per cloud keeps
//                 every point and its running min-distance in REGISTERS for all `npoint` dependent
//                 iterations (the reference launches ~6 tiny kernels per iteration); the argmax is a
//                 wave shuffle + one LDS hop; ties resolve to the lowest index like torch.max.
//                 Distances are computed as ((dx*dx + dy*dy) + dz*dz) with every operation rounded
//                 separately (`#pragma clang fp contract(off)` on plain operators: the __f*_rn intrinsics do NOT stop
//                 hipcc's contraction) to reproduce the CPU reference's fp32 values.
//   vl_knn_group  k nearest points of every centre with the reference's expanded distance
//                 -2ab + |a|^2 + |b|^2 (dvae.py:107-140) via an exact radix select of the k-th smallest
//                 key (no sort, no N x G distance matrix in HBM), then gather + centre-subtract
//                 (Group.forward, dvae.py:150-176) straight into the bf16 GEMM operand of the first conv.
//   vl_ball_group ball query + grouping of the `pnsa` tokenizer (query_ball_point / sample_and_group,
//                 open_clip/modal_3d/models/pointnet/pointnet_util.py:101-161): per centre the FIRST nsample point indices
//                 (ascending) with -2ab + |a|^2 + |b|^2 <= r^2, short groups padded with their first index; one wave per
//                 centre scans the cloud in index order (ballot + prefix popcount, stops at nsample hits - no N x S
//                 distance matrix, no sort), then writes the centre-subtracted xyz ++ point features as bf16 GEMM rows
//   vl_group_max  max over the points of a group (torch.max(feature, dim=2), dvae.py:206,211)
//   vl_pc_gather_normalize  the data-loader side of the 3D path on the GPU: gather the sampled points and
//                 scale them into the unit sphere (pc_norm, modal_3d/processors/pc_processor.py:32-38); with vl_fps in
//                 front of it this is PCProcessorEval (`uniform` sampling to 8192 points)
#include "vl_common.h"
#include "vitlens_hip.h"

namespace {

constexpr int FPS_THREADS = 1024;

template <int PPT>
__global__ void __launch_bounds__(FPS_THREADS) fps_kernel(const float* xyz, const int64_t* start, int64_t* out_idx,
                                                          float* centers, int N, int G) {
  __shared__ float s_val[16];
  __shared__ int s_idx[16];
  __shared__ int s_win;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* P = xyz + (size_t)b * N * 3;
  float px[PPT], py[PPT], pz[PPT], dist[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const int i = tid + k * FPS_THREADS;
    if (i < N) { px[k] = P[i * 3]; py[k] = P[i * 3 + 1]; pz[k] = P[i * 3 + 2]; dist[k] = 1e10f; }
    else { px[k] = py[k] = pz[k] = 0.f; dist[k] = -1.f; }   // never selected: real distances are >= 0
  }
  int far = (int)start[b];
  for (int it = 0; it < G; ++it) {
    if (tid == 0) {
      out_idx[(size_t)b * G + it] = far;
      if (centers) { float* c = centers + ((size_t)b * G + it) * 3; c[0] = P[far * 3]; c[1] = P[far * 3 + 1]; c[2] = P[far * 3 + 2]; }
    }
    const float cx = P[far * 3], cy = P[far * 3 + 1], cz = P[far * 3 + 2];
    float best = -2.f; int besti = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      const int i = tid + k * FPS_THREADS;
      if (i < N) {
        // no FMA: hipcc contracts a*a + b*b into v_fma_f32 even through the __f*_rn intrinsics (they are plain operators
        // in HIP), which moved one distance by an ulp and swapped two consecutive picks against the reference
        // (found with tools at N = 3000 / G = 2900; the 8192 -> 512 goldens never hit it)
#pragma clang fp contract(off)
        const float dx = px[k] - cx, dy = py[k] - cy, dz = pz[k] - cz;
        const float d = (dx * dx + dy * dy) + dz * dz;          // every operation rounded separately
        const float m = fminf(dist[k], d);
        dist[k] = m;
        if (m > best) { best = m; besti = i; }      // increasing i within a thread: strict > keeps the lowest index
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float v2 = __shfl_xor(best, o, 64); const int i2 = __shfl_xor(besti, o, 64);
      if (v2 > best || (v2 == best && i2 < besti)) { best = v2; besti = i2; }
    }
    if (lane == 0) { s_val[wv] = best; s_idx[wv] = besti; }
    __syncthreads();
    if (wv == 0) {
      float v = lane < 16 ? s_val[lane] : -3.f; int i = lane < 16 ? s_idx[lane] : 0x7fffffff;
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) {
        const float v2 = __shfl_xor(v, o, 64); const int i2 = __shfl_xor(i, o, 64);
        if (v2 > v || (v2 == v && i2 < i)) { v = v2; i = i2; }
      }
      if (lane == 0) s_win = i;
    }
    __syncthreads();
    far = s_win;
  }
}

__device__ __forceinline__ unsigned int sort_key(float f) {
  const unsigned int u = __builtin_bit_cast(unsigned int, f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// one wave per centre; the N sortable distance keys of the wave live in LDS (conflict-free lane-strided access)
__global__ void __launch_bounds__(256) knn_group_kernel(const float* xyz, const int64_t* cidx, int N, int G, int k,
                                                        int* nidx, bf16_t* patches, int Kp) {
  extern __shared__ unsigned int s_keys[];
  const int lane = threadIdx.x & 63;
  unsigned int* key = s_keys + (size_t)(threadIdx.x >> 6) * N;
  const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);        // global centre index b*G + g
  const int b = (int)(w / G);
  const float* P = xyz + (size_t)b * N * 3;
  const int ci = (int)cidx[w];
  const float cx = P[ci * 3], cy = P[ci * 3 + 1], cz = P[ci * 3 + 2];
  const float cc = __fadd_rn(__fadd_rn(__fmul_rn(cx, cx), __fmul_rn(cy, cy)), __fmul_rn(cz, cz));
  const int npl = N >> 6;
  for (int j = 0; j < npl; ++j) {
    const int i = lane + j * 64;
    const float x = P[i * 3], y = P[i * 3 + 1], z = P[i * 3 + 2];
    const float dot = fmaf(cz, z, fmaf(cy, y, cx * x));
    const float pp = __fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z));
    const float d = __fadd_rn(__fadd_rn(-2.0f * dot, cc), pp);      // dist = -2ab; dist += |a|^2; dist += |b|^2
    key[i] = sort_key(d);
  }
  // largest T with count(key < T) < k  ==  k-th smallest key   (each lane only re-reads its own LDS words)
  unsigned int T = 0;
  for (int bit = 31; bit >= 0; --bit) {
    const unsigned int trial = T | (1u << bit);
    int c = 0;
    for (int j = 0; j < npl; ++j) c += key[lane + j * 64] < trial;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if (c < k) T = trial;
  }
  int nless = 0;
  for (int j = 0; j < npl; ++j) nless += key[lane + j * 64] < T;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) nless += __shfl_xor(nless, o, 64);
  int need_eq = k - nless;                                   // ties at the k-th distance: lowest indices first
  int base = 0;
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (int j = 0; j < npl; ++j) {
    const int i = lane + j * 64;
    const unsigned int kj = key[i];
    const bool less = kj < T;
    const bool eq = kj == T;
    const unsigned long long bl = __ballot(less), be = __ballot(eq);
    const int eq_rank = __popcll(be & lt_mask);
    const bool take_eq = eq && eq_rank < need_eq;
    const unsigned long long bt = bl | __ballot(take_eq);
    const bool take = less || take_eq;
    if (take) {
      const int slot = base + __popcll(bt & lt_mask);
      if (nidx) nidx[w * k + slot] = i;
      if (patches) {
        bf16_t* o = patches + ((size_t)w * k + slot) * Kp;
        o[0] = f2bf(__fsub_rn(P[i * 3], cx)); o[1] = f2bf(__fsub_rn(P[i * 3 + 1], cy)); o[2] = f2bf(__fsub_rn(P[i * 3 + 2], cz));
        for (int e = 3; e < Kp; ++e) o[e] = 0;
      }
    }
    base += __popcll(bt);
    need_eq -= min(need_eq, __popcll(be));
  }
}

// Register form of knn_group_kernel for N = 64 * NPL points (round 3): a lane keeps its NPL sortable keys in VGPRs instead of
// LDS, and the per-bit counts of the radix select are wave ballots + scalar popcounts (no LDS, no cross-lane shuffles; the
// LDS form ran one 128 KB workgroup per CU and took 5.9 ms for 128 x 512 centres of 8192 points,
// profiles/r03_bench_c5_kernel_stats.csv).  Same arithmetic, same selection order, same outputs.
// wave-wide integer sum on the VALU: quad swaps, half-row / row mirrors (DPP), then the four rows through v_readlane
__device__ __forceinline__ int wave_sum_int(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true); v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true); v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);
  return (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) + (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
}

// LDSC (round 4): the WPB centres of a workgroup belong to ONE cloud (host: G % WPB == 0), which is copied once into LDS -
// [N][3] as it lies in memory (lane stride 3 dwords: conflict-free) plus the N squared norms - instead of every wave reading
// its 96 KB twice through L1 / L2 (12.6 GB of cache traffic per 128 x 512 centres of 8192 points); the squared norms are the
// same expression evaluated once per cloud instead of twice per centre.  Same arithmetic, same bits, same selection order.
constexpr int KNN_CAP = 256;      // candidate list entries per wave (LDS-staged form)
template <int NPL, int WPB, bool LDSC>
__global__ void __launch_bounds__(WPB * 64) knn_group_reg_kernel(const float* xyz, const int64_t* cidx, int G, int k, int* nidx,
                                                                bf16_t* patches, int Kp) {
  constexpr int N = NPL * 64;
  extern __shared__ __attribute__((aligned(16))) float s_cloud[];    // LDSC: [N * 3] points, [N] squared norms, [WPB][2][KNN_CAP] candidates
  const int lane = threadIdx.x & 63;
  const long w = (long)blockIdx.x * WPB + (threadIdx.x >> 6);      // global centre index b*G + g
  const int b = (int)(w / G);
  const float* P = xyz + (size_t)b * N * 3;
  if constexpr (LDSC) {
    const f32x4* src = (const f32x4*)P;                              // (host: 16-byte aligned; N * 12 bytes is a multiple of 16)
    for (int i = threadIdx.x; i < N * 3 / 4; i += WPB * 64) ((f32x4*)s_cloud)[i] = src[i];
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += WPB * 64) {
#pragma clang fp contract(off)
      const float x = s_cloud[i * 3], y = s_cloud[i * 3 + 1], z = s_cloud[i * 3 + 2];
      s_cloud[N * 3 + i] = (x * x + y * y) + z * z;
    }
    __syncthreads();
  }
  auto coord = [&](int i, int c) { if constexpr (LDSC) return s_cloud[i * 3 + c]; else return P[i * 3 + c]; };
  const int ci = (int)cidx[w];
  const float cx = coord(ci, 0), cy = coord(ci, 1), cz = coord(ci, 2);
  float cc;
  {
#pragma clang fp contract(off)
    cc = (cx * cx + cy * cy) + cz * cz;
  }
  auto dist_key = [&](int i) {
    // no implicit contraction: the key is computed twice (radix select, selection pass) and must come out identical
#pragma clang fp contract(off)
    const float x = coord(i, 0), y = coord(i, 1), z = coord(i, 2);
    const float dot = __builtin_fmaf(cz, z, __builtin_fmaf(cy, y, cx * x));
    float pp;
    if constexpr (LDSC) pp = s_cloud[N * 3 + i]; else pp = (x * x + y * y) + z * z;
    return sort_key((-2.0f * dot + cc) + pp);                    // dist = -2ab; dist += |a|^2; dist += |b|^2
  };
  unsigned int key[NPL];
#pragma unroll
  for (int j = 0; j < NPL; ++j) {
    key[j] = dist_key(lane + j * 64);
    if ((j & 15) == 15) __builtin_amdgcn_sched_barrier(0);      // at most 16 points' loads in flight: the keys need the registers
  }
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  auto emit = [&](int i, int slot) {
    if (nidx) nidx[w * k + slot] = i;
    if (patches) {
      bf16_t* o = patches + ((size_t)w * k + slot) * Kp;
      o[0] = f2bf(__fsub_rn(coord(i, 0), cx)); o[1] = f2bf(__fsub_rn(coord(i, 1), cy)); o[2] = f2bf(__fsub_rn(coord(i, 2), cz));
      for (int e = 3; e < Kp; ++e) o[e] = 0;
    }
  };
  if constexpr (LDSC) {
    // Candidate pre-selection (k <= 64): the k-th smallest of the 64 per-lane MINIMA is an upper bound U of the k-th smallest
    // key with at least k keys <= U (each lane minimum is a key) and - for scattered points - only about 1.4 k keys below it.
    // Those candidates go to a per-wave LDS list in INDEX order (ballot prefix), and both the radix select and the
    // tie-aware selection run on <= 4 keys per lane instead of NPL.  More than KNN_CAP candidates (masses of equal
    // distances): the full-width path below.  Same threshold, same tie rule, same output order - bit-identical results.
    unsigned int* ck_l = (unsigned int*)(s_cloud + N * 4) + (threadIdx.x >> 6) * (2 * KNN_CAP);
    int* ci_l = (int*)ck_l + KNN_CAP;
    if (k <= 64) {
      unsigned int mn = key[0];
#pragma unroll
      for (int j = 1; j < NPL; ++j) mn = min(mn, key[j]);
      unsigned int U = 0;                                      // largest U with count(mn < U) < k == k-th smallest lane minimum
      for (int bit = 31; bit >= 0; --bit) {
        const unsigned int trial = U | (1u << bit);
        if (__popcll(__ballot(mn < trial)) < k) U = trial;
      }
      int ncand = 0;
#pragma unroll
      for (int j = 0; j < NPL; ++j) {
        const bool c = key[j] <= U;
        const unsigned long long bl = __ballot(c);
        if (bl != 0ull) {                                      // (wave-uniform)
          const int pos = ncand + __popcll(bl & lt_mask);
          if (c && pos < KNN_CAP) { ck_l[pos] = key[j]; ci_l[pos] = lane + j * 64; }
          ncand += __popcll(bl);
        }
      }
      if (ncand <= KNN_CAP) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        constexpr int CPL = KNN_CAP / 64;
        unsigned int ck[CPL]; int ci[CPL]; bool cv[CPL];
#pragma unroll
        for (int m = 0; m < CPL; ++m) {
          const int pth = lane + m * 64;
          cv[m] = pth < ncand;
          ck[m] = cv[m] ? ck_l[pth] : 0xffffffffu;             // never < any trial; equality is masked by cv
          ci[m] = cv[m] ? ci_l[pth] : 0;
        }
        unsigned int T = 0;
        for (int bit = 31; bit >= 0; --bit) {
          const unsigned int trial = T | (1u << bit);
          int c = 0;
#pragma unroll
          for (int m = 0; m < CPL; ++m) c += __popcll(__ballot(ck[m] < trial));
          if (c < k) T = trial;
        }
        int nlt = 0;
#pragma unroll
        for (int m = 0; m < CPL; ++m) nlt += __popcll(__ballot(ck[m] < T));
        int need_eq = k - nlt, base = 0;
#pragma unroll
        for (int m = 0; m < CPL; ++m) {
          const bool less = ck[m] < T;
          const bool eq = cv[m] && ck[m] == T;
          const unsigned long long bl = __ballot(less), be = __ballot(eq);
          const int eq_rank = __popcll(be & lt_mask);
          const bool take_eq = eq && eq_rank < need_eq;
          const unsigned long long bt = bl | __ballot(take_eq);
          if (less || take_eq) emit(ci[m], base + __popcll(bt & lt_mask));
          base += __popcll(bt);
          need_eq -= min(need_eq, __popcll(be));
        }
        return;
      }
    }
  }
  // largest T with count(key < T) < k  ==  k-th smallest key
  unsigned int T = 0;
  for (int bit = 31; bit >= 0; --bit) {
    const unsigned int trial = T | (1u << bit);
    int c = 0;                                               // per-lane count (compare + add-with-carry), one wave sum per bit
#pragma unroll
    for (int j = 0; j < NPL; ++j) c += key[j] < trial;
    if (wave_sum_int(c) < k) T = trial;
  }
  int nl = 0;
#pragma unroll
  for (int j = 0; j < NPL; ++j) nl += key[j] < T;
  int need_eq = k - wave_sum_int(nl);                        // ties at the k-th distance: lowest indices first
  int base = 0;
  // selection in index order: a ROLLED loop that recomputes each key from the (cache- or LDS-resident) cloud - the same
  // arithmetic, hence the same bits - instead of 128 unrolled ballot groups on the register array (which spilled 178 SGPRs)
  for (int j = 0; j < NPL; ++j) {
    const int i = lane + j * 64;
    const unsigned int kj = dist_key(i);
    const bool less = kj < T;
    const bool eq = kj == T;
    const unsigned long long bl = __ballot(less), be = __ballot(eq);
    if ((bl | be) == 0ull) continue;                         // (wave-uniform) nothing of this slice is selected
    const int eq_rank = __popcll(be & lt_mask);
    const bool take_eq = eq && eq_rank < need_eq;
    const unsigned long long bt = bl | __ballot(take_eq);
    if (less || take_eq) emit(i, base + __popcll(bt & lt_mask));
    base += __popcll(bt);
    need_eq -= min(need_eq, __popcll(be));
  }
}

// one wave per centre.  The distance is the reference's expression with the CPU's rounding (pinned on 10.24 M distances
// against torch: the K = 3 matmul is an FMA chain x, y, z; the squared norms are plain sums; tests/test_oracle_pnsa.py).
__global__ void __launch_bounds__(256) ball_group_kernel(const float* xyz, const float* feats, const int64_t* cidx, int N, int S,
                                                         int D, float r2, int ns, int* out_idx, bf16_t* patches, int Kp) {
  extern __shared__ int s_ball[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int* my = s_ball + wv * ns;
  const long w = (long)blockIdx.x * 4 + wv;                        // global centre index b*S + s
  const int b = (int)(w / S);
  const float* P = xyz + (size_t)b * N * 3;
  const int ci = (int)cidx[w];
  const float cx = P[ci * 3], cy = P[ci * 3 + 1], cz = P[ci * 3 + 2];
  int count = 0, first = 0;
  {
#pragma clang fp contract(off)
    const float cc = (cx * cx + cy * cy) + cz * cz;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int base = 0; base < N && count < ns; base += 64) {
      const int i = base + lane;
      bool in = false;
      if (i < N) {
        const float x = P[i * 3], y = P[i * 3 + 1], z = P[i * 3 + 2];
        const float dot = __builtin_fmaf(cz, z, __builtin_fmaf(cy, y, cx * x));
        const float pp = (x * x + y * y) + z * z;
        const float d = (-2.0f * dot + cc) + pp;                   // (-2 is a power of two: a fused -2*dot + cc rounds identically)
        in = !(d > r2);                                            // the reference drops d > r^2
      }
      const unsigned long long m = __ballot(in);
      if (m) {
        if (count == 0) first = base + __ffsll((long long)m) - 1;
        const int rank = count + __popcll(m & lt_mask);
        if (in && rank < ns) my[rank] = i;
        count += __popcll(m);
      }
    }
  }
  count = min(count, ns);
  for (int t = lane; t < ns; t += 64)
    if (t >= count) my[t] = first;                                 // group_idx[mask] = group_first[mask]
  __syncthreads();
  if (out_idx)
    for (int t = lane; t < ns; t += 64) out_idx[w * ns + t] = my[t];
  if (patches) {
    const float* F = feats ? feats + (size_t)b * N * D : nullptr;
    for (int t = 0; t < ns; ++t) {
      const int i = my[t];
      bf16_t* o = patches + ((size_t)w * ns + t) * Kp;
      for (int e = lane; e < Kp; e += 64) {
        float v = 0.f;
        if (e < 3) v = P[i * 3 + e] - (e == 0 ? cx : (e == 1 ? cy : cz));
        else if (e < 3 + D) v = F[(size_t)i * D + (e - 3)];
        o[e] = f2bf(v);
      }
    }
  }
}

// out[g, c] = max_{m < M} x[g*M + m, c]
template <typename TOUT>
__global__ void __launch_bounds__(256) group_max_kernel(const bf16_t* x, long ldx, TOUT* out, long ldo, long groups, int M, int C) {
  const long n = groups * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long g = i / C; const int c = (int)(i - g * C);
    float m = -INFINITY;
    for (int r = 0; r < M; ++r) m = fmaxf(m, bf2f(x[(g * M + r) * ldx + c]));
    if constexpr (sizeof(TOUT) == 4) out[g * ldo + c] = m; else out[g * ldo + c] = f2bf(m);
  }
}

// centres [R,3] f32 -> bf16 [R,Kp] zero padded (operand of the pos_embed MLP's first Linear)
__global__ void __launch_bounds__(256) pad3_kernel(const float* c, bf16_t* out, long R, int Kp) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < R * Kp; i += (long)gridDim.x * 256) {
    const long r = i / Kp; const int e = (int)(i - r * Kp);
    out[i] = e < 3 ? f2bf(c[r * 3 + e]) : (bf16_t)0;
  }
}

// out[b, g, :] = (pts[b, idx[b,g], :] - mean_g) / max_g ||pts - mean||   (pc_norm after the gather), C <= 8 channels.
// One workgroup per cloud; fixed-order block reductions (deterministic): per-thread strided partials, then a tree in LDS.
__global__ void __launch_bounds__(1024) pc_gather_norm_kernel(const float* pts, const int64_t* idx, float* out, int N, int G, int C) {
  __shared__ float sh[1024];
  __shared__ float s_mean[8];
  __shared__ float s_max;
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* P = pts + (size_t)b * N * C;
  const int64_t* I = idx ? idx + (size_t)b * G : nullptr;
  auto row = [&](int g) { return P + (size_t)(I ? I[g] : g) * C; };
  for (int c = 0; c < C; ++c) {
    float a = 0.f;
    for (int g = tid; g < G; g += 1024) a += row(g)[c];
    sh[tid] = a;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
      if (tid < w) sh[tid] += sh[tid + w];
      __syncthreads();
    }
    if (tid == 0) s_mean[c] = sh[0] / (float)G;
    __syncthreads();
  }
  float m = 0.f;
  for (int g = tid; g < G; g += 1024) {
    const float* r = row(g);
    float q = 0.f;
    for (int c = 0; c < C; ++c) { const float d = r[c] - s_mean[c]; q = fmaf(d, d, q); }
    m = fmaxf(m, sqrtf(q));
  }
  sh[tid] = m;
  __syncthreads();
  for (int w = 512; w > 0; w >>= 1) {
    if (tid < w) sh[tid] = fmaxf(sh[tid], sh[tid + w]);
    __syncthreads();
  }
  if (tid == 0) s_max = sh[0];
  __syncthreads();
  const float inv = 1.0f / s_max;
  for (int g = tid; g < G; g += 1024) {
    const float* r = row(g);
    float* o = out + ((size_t)b * G + g) * C;
    for (int c = 0; c < C; ++c) o[c] = (r[c] - s_mean[c]) * inv;
  }
}

}  // namespace

extern "C" int vl_set_error(const char* msg);
#define VL_HIP_OK(e) do { hipError_t _e = (e); if (_e != hipSuccess) return vl_set_error(hipGetErrorString(_e)); } while (0)
static int grid_for(long n) { long g = (n + 255) / 256; return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g)); }

extern "C" int vl_fps(const float* xyz, const int64_t* start, int64_t* idx, float* centers, int B, int N, int G,
                      hipStream_t stream) {
  if (B <= 0 || N <= 0 || G <= 0 || G > N) return vl_set_error("vl_fps: bad shape");
  const int ppt = (N + FPS_THREADS - 1) / FPS_THREADS;
  if (ppt <= 1) hipLaunchKernelGGL(fps_kernel<1>, dim3(B), dim3(FPS_THREADS), 0, stream, xyz, start, idx, centers, N, G);
  else if (ppt <= 2) hipLaunchKernelGGL(fps_kernel<2>, dim3(B), dim3(FPS_THREADS), 0, stream, xyz, start, idx, centers, N, G);
  else if (ppt <= 4) hipLaunchKernelGGL(fps_kernel<4>, dim3(B), dim3(FPS_THREADS), 0, stream, xyz, start, idx, centers, N, G);
  else if (ppt <= 8) hipLaunchKernelGGL(fps_kernel<8>, dim3(B), dim3(FPS_THREADS), 0, stream, xyz, start, idx, centers, N, G);
  else if (ppt <= 16) hipLaunchKernelGGL(fps_kernel<16>, dim3(B), dim3(FPS_THREADS), 0, stream, xyz, start, idx, centers, N, G);
  else return vl_set_error("vl_fps: at most 16384 points per cloud");
  VL_HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int vl_knn_group(const float* xyz, const int64_t* center_idx, int* nidx, void* patches, int B, int N, int G,
                            int k, int Kp, hipStream_t stream) {
  if (B <= 0 || G <= 0 || k <= 0 || k > N) return vl_set_error("vl_knn_group: bad shape");
  if ((N & 63) || ((long)B * G) % 4) return vl_set_error("vl_knn_group: N must be a multiple of 64 and B*G of 4");
  if (patches && Kp < 3) return vl_set_error("vl_knn_group: Kp < 3");
  const dim3 grid((unsigned)(((long)B * G) / 4)), block(256);
  // keys in registers for the cloud sizes of the configs (8192: the Lens, 1024 / 2048: ablations, 256: the tests); with 8
  // centres of one cloud per workgroup the cloud itself is staged in LDS (16 bytes per point)
  const bool lds_ok = (G % 8) == 0 && (((uintptr_t)xyz) & 15) == 0;
  switch (N) {
#define VL_KNN_REG(NPLV)                                                                                                       \
  case NPLV * 64:                                                                                                              \
    if (lds_ok) {                                                                                                              \
      static const hipError_t attr = hipFuncSetAttribute((const void*)knn_group_reg_kernel<NPLV, 8, true>,                     \
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, NPLV * 64 * 16 + 8 * KNN_CAP * 8);          \
      VL_HIP_OK(attr);                                                                                                         \
      hipLaunchKernelGGL((knn_group_reg_kernel<NPLV, 8, true>), dim3((unsigned)(((long)B * G) / 8)), dim3(512),                \
                         (size_t)NPLV * 64 * 16 + 8 * KNN_CAP * 8, stream, xyz, center_idx, G, k, nidx, (bf16_t*)patches, Kp);                   \
    } else {                                                                                                                   \
      hipLaunchKernelGGL((knn_group_reg_kernel<NPLV, 4, false>), grid, block, 0, stream, xyz, center_idx, G, k, nidx,          \
                         (bf16_t*)patches, Kp);                                                                                \
    }                                                                                                                          \
    VL_HIP_OK(hipGetLastError());                                                                                              \
    return 0;
    VL_KNN_REG(4) VL_KNN_REG(16) VL_KNN_REG(32) VL_KNN_REG(128)
#undef VL_KNN_REG
    default: break;
  }
  const size_t smem = (size_t)4 * N * sizeof(unsigned int);
  if (smem > 160 * 1024) return vl_set_error("vl_knn_group: at most 10240 points per cloud");
  static bool attr = false;
  if (!attr) { VL_HIP_OK(hipFuncSetAttribute((const void*)knn_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; }
  hipLaunchKernelGGL(knn_group_kernel, grid, block, smem, stream, xyz, center_idx, N, G, k, nidx, (bf16_t*)patches, Kp);
  VL_HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int vl_ball_group(const float* xyz, const float* feats, const int64_t* center_idx, int* idx, void* patches, int B,
                             int N, int S, int D, float radius2, int nsample, int Kp, hipStream_t stream) {
  if (B <= 0 || N <= 0 || S <= 0 || nsample <= 0 || D < 0) return vl_set_error("vl_ball_group: bad shape");
  if (((long)B * S) % 4) return vl_set_error("vl_ball_group: B*S must be a multiple of 4");
  if (D > 0 && !feats) return vl_set_error("vl_ball_group: D > 0 without point features");
  if (patches && Kp < 3 + D) return vl_set_error("vl_ball_group: Kp < 3 + D");
  if (nsample > 2048) return vl_set_error("vl_ball_group: at most 2048 samples per group");
  hipLaunchKernelGGL(ball_group_kernel, dim3((unsigned)(((long)B * S) / 4)), dim3(256), (size_t)4 * nsample * sizeof(int), stream,
                     xyz, feats, center_idx, N, S, D, radius2, nsample, idx, (bf16_t*)patches, Kp);
  VL_HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int vl_group_max(const void* x, long ldx, void* out, int out_dtype, long ldo, long groups, int M, int C,
                            hipStream_t stream) {
  if (groups <= 0 || M <= 0 || C <= 0) return vl_set_error("vl_group_max: bad shape");
  if (out_dtype == VL_F32) hipLaunchKernelGGL(group_max_kernel<float>, dim3(grid_for(groups * C)), dim3(256), 0, stream, (const bf16_t*)x, ldx, (float*)out, ldo, groups, M, C);
  else hipLaunchKernelGGL(group_max_kernel<bf16_t>, dim3(grid_for(groups * C)), dim3(256), 0, stream, (const bf16_t*)x, ldx, (bf16_t*)out, ldo, groups, M, C);
  VL_HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int vl_pad3_bf16(const float* c, void* out, long R, int Kp, hipStream_t stream) {
  if (R <= 0 || Kp < 3) return vl_set_error("vl_pad3_bf16: bad shape");
  hipLaunchKernelGGL(pad3_kernel, dim3(grid_for(R * Kp)), dim3(256), 0, stream, c, (bf16_t*)out, R, Kp);
  VL_HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int vl_pc_gather_normalize(const float* pts, const int64_t* idx, float* out, int B, int N, int G, int C,
                                      hipStream_t stream) {
  if (B <= 0 || N <= 0 || G <= 0 || C < 3 || C > 8) return vl_set_error("vl_pc_gather_normalize: bad shape (3 <= C <= 8)");
  if (!idx && G != N) return vl_set_error("vl_pc_gather_normalize: without an index list G must equal N");
  hipLaunchKernelGGL(pc_gather_norm_kernel, dim3(B), dim3(1024), 0, stream, pts, idx, out, N, G, C);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : vl_set_error(hipGetErrorString(e));
}
