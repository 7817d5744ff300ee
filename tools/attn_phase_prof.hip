// Phase timeline of the attention kernels at the bench shape (B=256, H=16, L=257, dh=64), operands read in place from
// token-major matrices.  Build: tools/build_attn_phase_prof.sh (the kernel sources compiled with -DVL_ATTN_PROF).
// Prints, per kernel, the mean shader-clock cycles between the phase stamps of vl_attn_common.h over all workgroups,
// and the mean interval between consecutive workgroup starts on a CU slot.
#include <hip/hip_runtime.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include "vitlens_hip.h"
extern "C" long* vl_attn_prof_buf;
long* vl_attn_prof_buf = nullptr;
extern "C" int vl_set_error(const char* m) { fprintf(stderr, "error: %s\n", m); return 1; }

static void report(const char* name, long* dprof, int nwg, float ms) {
  std::vector<long> h((size_t)nwg * 8);
  hipMemcpy(h.data(), dprof, h.size() * 8, hipMemcpyDeviceToHost);
  double d[5] = {0, 0, 0, 0, 0}; double life = 0; long t0 = h[0], t1 = h[5];
  for (int w = 0; w < nwg; ++w) {
    for (int i = 0; i < 5; ++i) d[i] += (double)(h[w * 8 + i + 1] - h[w * 8 + i]);
    life += (double)(h[w * 8 + 5] - h[w * 8]);
    t0 = std::min(t0, h[w * 8]); t1 = std::max(t1, h[w * 8 + 5]);
  }
  printf("%-8s %.3f ms  span %ld clk  | per WG: load+stage %.0f  barrier %.0f  main %.0f  store %.0f  tail %.0f  = life %.0f clk\n",
         name, ms, t1 - t0, d[0] / nwg, d[1] / nwg, d[2] / nwg, d[3] / nwg, d[4] / nwg, life / nwg);
}

int main(int argc, char** argv) {
  const int B = 256, H = 16, dh = 64, L = argc > 1 ? atoi(argv[1]) : 257, D = H * dh;
  const size_t T = (size_t)B * L;
  unsigned short *qkv, *o, *dO, *dqkv; float *lse, *delta; long* prof;
  hipMalloc(&qkv, T * 3 * D * 2); hipMalloc(&o, T * D * 2); hipMalloc(&dO, T * D * 2); hipMalloc(&dqkv, T * 3 * D * 2);
  hipMalloc(&lse, (size_t)B * H * L * 4); hipMalloc(&delta, (size_t)B * H * L * 4);
  const int nwg = B * H; hipMalloc(&prof, (size_t)2 * nwg * 8 * 8);
  std::vector<unsigned short> h(T * 3 * D);
  srand(1);
  for (auto& x : h) { float f = (rand() / (float)RAND_MAX - 0.5f) * 2.f; unsigned u; memcpy(&u, &f, 4); x = u >> 16; }
  hipMemcpy(qkv, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dO, h.data(), T * D * 2, hipMemcpyHostToDevice);
  const long W = 3 * D;
  long st[15] = {(long)L * W, dh, W, (long)L * W, dh, W, (long)L * W, dh, W, (long)L * D, dh, D, (long)L * D, dh, D};
  const float qs = 0.125f * 1.4426950408889634f;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timed = [&](auto fn) { fn(); hipDeviceSynchronize(); hipEventRecord(e0); fn(); hipEventRecord(e1); hipEventSynchronize(e1);
                              float ms; hipEventElapsedTime(&ms, e0, e1); return ms; };
  vl_attn_prof_buf = prof;
  float ms = timed([&] { vl_attn_fwd_bf16(qkv, qkv + D, qkv + 2 * D, st, o, lse, B, H, L, L, dh, qs, 0, 0); });
  report("fwd", prof, nwg, ms);
  // backward: kernel A's stamps, then kernel B's (ms = the pair)
  ms = timed([&] { vl_attn_bwd_bf16(qkv, qkv + D, qkv + 2 * D, dO, o, st, lse, delta, dqkv, dqkv + D, dqkv + 2 * D, W, W,
                                    B, H, L, L, dh, qs, 0, 0.125f, 0); });
  report("bwd dq", prof, nwg, ms);
  report("bwd dkv", prof + (size_t)nwg * 8, nwg, ms);
  if (vl_attn_bwd_fused_supported(L, L, dh, 0)) {
    if (argc > 2 && atoi(argv[2]) == 1) {      // every (batch, head) reads the SAME operand rows: staging out of L2 (is the load phase HBM-bound?)
      for (int i = 0; i < 15; i += 3) { st[i] = 0; st[i + 1] = 0; }
      printf("(all items alias item 0's operands)\n");
    }
    ms = timed([&] { vl_attn_bwd_fused_bf16(qkv, qkv + D, qkv + 2 * D, dO, o, st, lse, dqkv, dqkv + D, dqkv + 2 * D, W, W, B, H, L, dh, qs, 0.125f, 0); });
    std::vector<long> hh((size_t)nwg * 8);
    hipMemcpy(hh.data(), prof, hh.size() * 8, hipMemcpyDeviceToHost);
    double d[6] = {0, 0, 0, 0, 0, 0};
    for (int w = 0; w < nwg; ++w)
      for (int i = 0; i < 6; ++i) d[i] += (double)(hh[w * 8 + i + 1] - hh[w * 8 + i]);
    printf("bwd ONE  %.3f ms | per WG: load+stage %.0f  barrier %.0f  lone row/col %.0f  tiles %.0f  lone finish %.0f  stores %.0f  = %.0f clk\n", ms,
           d[0] / nwg, d[1] / nwg, d[2] / nwg, d[3] / nwg, d[4] / nwg, d[5] / nwg, (d[0] + d[1] + d[2] + d[3] + d[4] + d[5]) / nwg);
  }
  return 0;
}
