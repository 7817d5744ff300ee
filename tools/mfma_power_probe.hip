// What the chip's power management allows on the bf16 matrix pipes: a register-only MFMA loop (no memory traffic inside
// the timed region) on random-normal, small-integer and zero-filled operands, sustained for about a second each.
// `v_mfma_f32_32x32x16_bf16` issued back to back on 8 accumulators per wave, 2 waves per SIMD, every CU.
// The random-data number is the ceiling any bf16 GEMM on this data distribution can approach on this board (DESIGN.md
// section 7, "power roofline"); the zero-data number shows what the same instruction stream does without the multipliers'
// switching energy.  Build: tools/build_probes.sh   Run: tools/bin/mfma_power_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// MODE 0: operands stay in registers.  MODE 1: the GEMM main loop's LDS traffic is added - per 8 MFMAs six ds_read_b128
// fragment reads (4 A + 2 B) from a 64 KB operand stage that was filled once from the random data.
template <int MODE>
__global__ void __launch_bounds__(512) mfma_loop(const bf16x8* __restrict__ src, float* __restrict__ sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int gid = blockIdx.x * 512 + threadIdx.x;
  if (MODE == 1) {
    for (int i = threadIdx.x; i < 4096; i += 512) ((bf16x8*)lds)[i] = src[(size_t)blockIdx.x * 3072 + (i % 3072)];
    __syncthreads();
  }
  bf16x8 a[4], b[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = src[(size_t)gid * 6 + i];
#pragma unroll
  for (int i = 0; i < 2; ++i) b[i] = src[(size_t)gid * 6 + 4 + i];
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 1) {       // conflict-free 16-byte reads, a different 1 KB slice per fragment and iteration
      const int base = ((it & 7) * 8 + wv) * 1024 + lane * 16;
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = *(const bf16x8*)(lds + ((base + i * 8192) & 65535));
#pragma unroll
      for (int i = 0; i < 2; ++i) b[i] = *(const bf16x8*)(lds + ((base + 32768 + i * 8192 + 4096) & 65535));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  if (s == 123456.789f) sink[gid] = s;      // keeps the accumulators live
}

typedef __attribute__((ext_vector_type(4))) float f32x4;

// MODE 2: the same wave tile (128 x 64 outputs, 128 accumulator registers) on `v_mfma_f32_16x16x32_bf16`: 32 MFMAs of 16 384
// flop per k = 32 step from 8 + 4 operand fragments; a quarter of the accumulator traffic per flop.
__global__ void __launch_bounds__(512) mfma_loop_16(const bf16x8* __restrict__ src, float* __restrict__ sink, int iters) {
  const int gid = blockIdx.x * 512 + threadIdx.x;
  bf16x8 a[8], b[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = src[(size_t)gid * 6 + (i % 6)];
#pragma unroll
  for (int i = 0; i < 4; ++i) b[i] = src[(size_t)gid * 6 + ((i + 2) % 6)];
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (s == 123456.789f) sink[gid] = s;
}

// MODE 3: MODE 1's LDS fragment traffic, software-pipelined (the reads of iteration it+1 are issued before the MFMAs of
// iteration it, as the GEMM kernels do): what the LDS reads cost in POWER once their latency is out of the way.
__global__ void __launch_bounds__(512) mfma_loop_lds_pipe(const bf16x8* __restrict__ src, float* __restrict__ sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int gid = blockIdx.x * 512 + threadIdx.x;
  for (int i = threadIdx.x; i < 4096; i += 512) ((bf16x8*)lds)[i] = src[(size_t)blockIdx.x * 3072 + (i % 3072)];
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  bf16x8 a[2][4], b[2][2];
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  auto rd = [&](int it, bf16x8 (&A)[4], bf16x8 (&B)[2]) {
    const int base = ((it & 7) * 8 + wv) * 1024 + lane * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) A[i] = *(const bf16x8*)(lds + ((base + i * 8192) & 65535));
#pragma unroll
    for (int i = 0; i < 2; ++i) B[i] = *(const bf16x8*)(lds + ((base + 32768 + i * 8192 + 4096) & 65535));
  };
  rd(0, a[0], b[0]);
  for (int it = 0; it < iters; it += 2) {
    rd(it + 1, a[1], b[1]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[0][j], a[0][i], acc[i][j], 0, 0, 0);
    rd(it + 2, a[0], b[0]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[1][j], a[1][i], acc[i][j], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  if (s == 123456.789f) sink[gid] = s;
}


// MODE 4: the persistent GEMM's CURRENT main loop without DMA / barrier / epilogue: `v_mfma_f32_16x16x32_bf16` on the 128 x 64
// wave tile (8 waves, two per SIMD), 12 ds_read_b128 fragment reads per 32 MFMAs, software-pipelined.  The ceiling of an
// LDS-fed 16x16x32 loop on this board = what the GEMM main loop is measured against (round 4).
__global__ void __launch_bounds__(512) mfma16_lds_8w(const bf16x8* __restrict__ src, float* __restrict__ sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int gid = blockIdx.x * 512 + threadIdx.x;
  for (int i = threadIdx.x; i < 4096; i += 512) ((bf16x8*)lds)[i] = src[(size_t)blockIdx.x * 3072 + (i % 3072)];
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  bf16x8 a[2][8], b[2][4];
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto rd = [&](int it, bf16x8 (&A)[8], bf16x8 (&B)[4]) {
    const int base = ((it & 3) * 8 + wv) * 1024 + lane * 16;
#pragma unroll
    for (int i = 0; i < 8; ++i) A[i] = *(const bf16x8*)(lds + ((base + i * 4096) & 65535));
#pragma unroll
    for (int i = 0; i < 4; ++i) B[i] = *(const bf16x8*)(lds + ((base + 32768 + i * 4096 + 2048) & 65535));
  };
  rd(0, a[0], b[0]);
  for (int it = 0; it < iters; it += 2) {
    rd(it + 1, a[1], b[1]);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[0][j], a[0][i], acc[i][j], 0, 0, 0);
    rd(it + 2, a[0], b[0]);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[1][j], a[1][i], acc[i][j], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (s == 123456.789f) sink[gid] = s;
}

// MODE 5: the same 256 x 256 workgroup tile as FOUR waves of 128 x 128 (one per SIMD, 256 accumulator registers): 16 fragment
// reads per 64 MFMAs - two thirds of MODE 4's LDS bytes per flop.  What a 128 x 128 wave tile could buy before its DMA /
// barrier / epilogue problems are even considered.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
    mfma16_lds_4w(const bf16x8* __restrict__ src, float* __restrict__ sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int gid = blockIdx.x * 256 + threadIdx.x;
  for (int i = threadIdx.x; i < 4096; i += 256) ((bf16x8*)lds)[i] = src[(size_t)blockIdx.x * 3072 + (i % 3072)];
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  bf16x8 a[2][8], b[2][8];
  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto rd = [&](int it, bf16x8 (&A)[8], bf16x8 (&B)[8]) {
    const int base = ((it & 3) * 4 + wv) * 1024 + lane * 16;
#pragma unroll
    for (int i = 0; i < 8; ++i) A[i] = *(const bf16x8*)(lds + ((base + i * 4096) & 65535));
#pragma unroll
    for (int i = 0; i < 8; ++i) B[i] = *(const bf16x8*)(lds + ((base + 32768 + i * 4096 + 2048) & 65535));
  };
  rd(0, a[0], b[0]);
  for (int it = 0; it < iters; it += 2) {
    rd(it + 1, a[1], b[1]);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[0][j], a[0][i], acc[i][j], 0, 0, 0);
    rd(it + 2, a[0], b[0]);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[1][j], a[1][i], acc[i][j], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (s == 123456.789f) sink[gid] = s;
}

template <typename K>
static double run_k(K kern, size_t smem, double flop_per_wave_iter, const unsigned short* d, float* sink, int blocks, int iters, int launches,
                    hipEvent_t e0, hipEvent_t e1, int waves = 8) {
  if (smem) hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kern, dim3(blocks), dim3(waves * 64), smem, 0, (const bf16x8*)d, sink, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int rep = 0; rep < launches; ++rep) hipLaunchKernelGGL(kern, dim3(blocks), dim3(waves * 64), smem, 0, (const bf16x8*)d, sink, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return (double)launches * blocks * waves * iters * flop_per_wave_iter / (ms * 1e-3) / 1e12;
}

static unsigned short f2bf(float f) {
  unsigned u; std::memcpy(&u, &f, 4);
  u += 0x7fff + ((u >> 16) & 1);
  return (unsigned short)(u >> 16);
}

template <int MODE>
static double run(const unsigned short* d, float* sink, int blocks, int iters, int launches, hipEvent_t e0, hipEvent_t e1) {
  const size_t smem = MODE == 1 ? 65536 : 0;
  hipFuncSetAttribute((const void*)mfma_loop<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(mfma_loop<MODE>, dim3(blocks), dim3(512), smem, 0, (const bf16x8*)d, sink, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int rep = 0; rep < launches; ++rep) hipLaunchKernelGGL(mfma_loop<MODE>, dim3(blocks), dim3(512), smem, 0, (const bf16x8*)d, sink, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return (double)launches * blocks * 8 * iters * 8.0 * (2.0 * 32 * 32 * 16) / (ms * 1e-3) / 1e12;
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 12;      // 12 launches ~ 60 ms per fill; 400 ~ 2 s (sustained)
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int ncu = prop.multiProcessorCount, blocks = ncu, threads = blocks * 512;
  std::vector<unsigned short> h((size_t)threads * 48);
  unsigned short* d; float* sink;
  hipMalloc(&d, h.size() * 2); hipMalloc(&sink, (size_t)threads * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[3] = {"randn", "smallint", "zeros"};
  for (int fill = 0; fill < 3; ++fill) {
    srand(1234);
    for (size_t i = 0; i < h.size(); ++i) {
      float v = 0.f;
      if (fill == 0) {           // Box-Muller normal, scaled like post-LayerNorm activations x 1/sqrt(K) weights would accumulate
        const float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = rand() / (float)RAND_MAX;
        v = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2) * 0.05f;       // small enough that 1e6 accumulations stay finite
      } else if (fill == 1) v = (float)(rand() % 5 - 2) * 0.001f;
      h[i] = f2bf(v);
    }
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const int iters = 20000;                      // 8 MFMAs per iteration per wave
    const double tf0 = run<0>(d, sink, blocks, iters, launches, e0, e1);
    const double tf1 = run<1>(d, sink, blocks, iters, launches, e0, e1);
    const double tf2 = run_k(mfma_loop_16, 0, 32.0 * (2.0 * 16 * 16 * 32), d, sink, blocks, iters / 2, launches, e0, e1);
    const double tf3 = run_k(mfma_loop_lds_pipe, 65536, 8.0 * (2.0 * 32 * 32 * 16), d, sink, blocks, iters, launches, e0, e1);
    const double tf4 = run_k(mfma16_lds_8w, 65536, 32.0 * (2.0 * 16 * 16 * 32), d, sink, blocks, iters / 2, launches, e0, e1);
    const double tf5 = run_k(mfma16_lds_4w, 65536, 64.0 * (2.0 * 16 * 16 * 32), d, sink, blocks, iters / 2, launches, e0, e1, 4);
    printf("mfma_power_probe %-8s: 16x16x32 + pipelined LDS reads, 8 waves x 128x64 (12 reads / 32 MFMAs) %7.1f TF/s (%.1f %%) | 4 waves x 128x128 (16 / 64) %7.1f TF/s (%.1f %%)\n",
           names[fill], tf4, 100.0 * tf4 / 2500.0, tf5, 100.0 * tf5 / 2500.0);
    printf("mfma_power_probe %-8s: 16x16x32 registers only %7.1f TF/s (%.1f %%) | 32x32x16 + pipelined LDS fragment reads %7.1f TF/s (%.1f %%)\n",
           names[fill], tf2, 100.0 * tf2 / 2500.0, tf3, 100.0 * tf3 / 2500.0);
    printf("mfma_power_probe %-8s: registers only %7.1f TF/s (%.1f %% of 2500, implied clock %.2f GHz) | + LDS fragment reads %7.1f TF/s (%.1f %%)  [%d CUs, %d launches]\n",
           names[fill], tf0, 100.0 * tf0 / 2500.0, tf0 / 2500.0 * 2.4, tf1, 100.0 * tf1 / 2500.0, ncu, launches);
  }
  return 0;
}
