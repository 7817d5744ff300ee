#!/usr/bin/env python
"""EXPERIMENTAL kernel check (vl_gemm_pk4.hip, cfg = 14: one wave per SIMD, 128x128 wave tiles; written in round 4 without a GPU
in reach): correctness against fp32 torch and against the shipped 8-wave kernel (cfg = 8) on the ViT-L block shapes, then
interleaved timing of both.  Run this FIRST when a GPU is available again; nothing in the product dispatches cfg 14."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-lens_amd"))
import torch
from vitlens_hip import ops

T = 256 * 256
SHAPES = {"small": (1024, 512, 512), "ragged-rounds": (256 * 37, 768, 1024), "qkv": (T, 3072, 1024), "fc": (T, 4096, 1024),
          "dfc": (T, 1024, 4096), "sq8k": (8192, 8192, 8192)}


def relerr(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / (b.norm() + 1e-30))


ok = True
for name, (M, N, K) in SHAPES.items():
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    bias = torch.randn(N, device="cuda")
    o8 = ops.gemm(a, w, bias, epi=ops.EPI_BF16, cfg=8)
    o4 = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    ops.gemm(a, w, bias, out=o4, epi=ops.EPI_BF16, cfg=14)
    torch.cuda.synchronize()
    e84 = relerr(o4, o8)
    same = bool(torch.equal(o4.view(torch.int16), o8.view(torch.int16)))
    ref_rows = slice(0, min(M, 2048))
    ref = a[ref_rows].float() @ w.float().t() + bias
    e4, e8 = relerr(o4[ref_rows], ref), relerr(o8[ref_rows], ref)
    fin = bool(torch.isfinite(o4.float()).all())
    good = fin and e4 < 4e-3 and e84 < 2e-3
    ok &= good
    print(f"{name:14s} {M}x{N}x{K}: finite {fin}  vs fp32 (first rows): pk4 {e4:.2e}  pk8 {e8:.2e}   pk4 vs pk8 {e84:.2e}  bit-identical {same}  {'OK' if good else 'WRONG'}", flush=True)
    # no-bias and alpha
    o4b = ops.gemm(a, w, None, epi=ops.EPI_BF16, cfg=14, alpha=0.5)
    o8b = ops.gemm(a, w, None, epi=ops.EPI_BF16, cfg=8, alpha=0.5)
    if relerr(o4b, o8b) > 2e-3:
        ok = False; print("   no-bias / alpha variant WRONG", relerr(o4b, o8b))
    if not good:
        continue
    fns = {8: lambda: ops.gemm(a, w, bias, out=o8, epi=ops.EPI_BF16, cfg=8), 14: lambda: ops.gemm(a, w, bias, out=o4, epi=ops.EPI_BF16, cfg=14)}
    ts = {8: [], 14: []}
    for _ in range(7):
        for c, f in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); f(); f(); e1.record(); torch.cuda.synchronize()
            ts[c].append(e0.elapsed_time(e1) / 2)
    for c in (8, 14):
        v = sorted(ts[c]); med = v[len(v) // 2]
        print(f"   cfg {c:2d}: med {med:7.4f} ms  {2.0 * M * N * K / med / 1e9:7.1f} TF/s")
    del a, w, o8, o4
print("ALL OK" if ok else "FAILURES")
