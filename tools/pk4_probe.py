#!/usr/bin/env python
"""EXPERIMENTAL kernel check (vl_gemm_pk4.hip, cfg = 14: one wave per SIMD, 128x128 wave tiles; written in round 4 without a GPU
in reach): every epilogue it has against the shipped 8-wave kernel (cfg = 8) and fp32 torch on the ViT-L block shapes, then
interleaved timing of both.  Run this FIRST when a GPU is available again; nothing in the product dispatches cfg 14."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-lens_amd"))
import torch
from vitlens_hip import ops

T = 256 * 256
SHAPES = {"small": (1024, 512, 512), "ragged-rounds": (256 * 37, 768, 1024), "qkv": (T, 3072, 1024), "out": (T, 1024, 1024),
          "fc": (T, 4096, 1024), "proj": (T, 1024, 4096), "sq8k": (8192, 8192, 8192)}
# variant -> shapes it is timed on (every variant is CHECKED on "small" and "ragged-rounds" first)
VARIANTS = {"bf16": ["qkv", "fc", "proj", "sq8k"], "gelu": ["fc"], "gelu+dsave": ["fc"], "res_bf16": ["out", "proj"], "dgelu_saved": ["fc"]}


def relerr(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / (b.norm() + 1e-30))


def run(variant, cfg, a, w, bias, out, aux, out2, inplace=False):
    if variant == "bf16":
        return ops.gemm(a, w, bias, out=out, epi=ops.EPI_BF16, cfg=cfg)
    if variant == "gelu":
        return ops.gemm(a, w, bias, out=out, epi=ops.EPI_BF16, act=ops.ACT_GELU, cfg=cfg)
    if variant == "gelu+dsave":
        return ops.gemm(a, w, bias, out=out, epi=ops.EPI_BF16, act=ops.ACT_GELU_DSAVE, cfg=cfg, out2=out2)
    if variant == "res_bf16":
        return ops.gemm(a, w, bias, out=out, res=out if inplace else aux, epi=ops.EPI_RES_BF16, cfg=cfg)
    return ops.gemm(a, w, None, out=out, res=aux, epi=ops.EPI_DGELU, act=ops.ACT_GELU_DSAVE, cfg=cfg)


ok = True
for variant, timed in VARIANTS.items():
    for name in ["small", "ragged-rounds"] + timed:
        M, N, K = SHAPES[name]
        a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
        bias = torch.randn(N, device="cuda")
        aux = torch.randn(M, N, device="cuda").bfloat16()
        o8 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); o4 = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
        u8 = torch.empty_like(o8); u4 = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
        run(variant, 8, a, w, bias, o8, aux, u8); run(variant, 14, a, w, bias, o4, aux, u4)
        torch.cuda.synchronize()
        e = relerr(o4, o8)
        same = bool(torch.equal(o4.view(torch.int16), o8.view(torch.int16)))
        fin = bool(torch.isfinite(o4.float()).all())
        good = fin and e < 2e-3
        if variant == "gelu+dsave":
            e2 = relerr(u4, u8); good = good and e2 < 2e-3 and bool(torch.isfinite(u4.float()).all())
        if variant == "res_bf16":       # in place, as the towers call it
            x8, x4 = aux.clone(), aux.clone()
            run(variant, 8, a, w, bias, x8, None, None, inplace=True); run(variant, 14, a, w, bias, x4, None, None, inplace=True)
            good = good and relerr(x4, x8) < 2e-3
        if variant == "bf16" and name == "small":
            ref = a.float() @ w.float().t() + bias
            print(f"   vs fp32: pk4 {relerr(o4, ref):.2e}  pk8 {relerr(o8, ref):.2e}")
            o4b = ops.gemm(a, w, None, epi=ops.EPI_BF16, cfg=14, alpha=0.5); o8b = ops.gemm(a, w, None, epi=ops.EPI_BF16, cfg=8, alpha=0.5)
            good = good and relerr(o4b, o8b) < 2e-3
        ok &= good
        print(f"{variant:12s} {name:14s} {M}x{N}x{K}: finite {fin}  pk4 vs pk8 {e:.2e}  bit-identical {same}  {'OK' if good else 'WRONG'}", flush=True)
        if not good or name not in timed:
            continue
        fns = {c: (lambda c=c: run(variant, c, a, w, bias, o8 if c == 8 else o4, aux, u8 if c == 8 else u4)) for c in (8, 14)}
        ts = {8: [], 14: []}
        for _ in range(7):
            for c, f in fns.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); f(); f(); e1.record(); torch.cuda.synchronize()
                ts[c].append(e0.elapsed_time(e1) / 2)
        for c in (8, 14):
            v = sorted(ts[c]); med = v[len(v) // 2]
            print(f"      cfg {c:2d}: med {med:7.4f} ms  {2.0 * M * N * K / med / 1e9:7.1f} TF/s")
        del a, w, o8, o4, u8, u4, aux
print("ALL OK" if ok else "FAILURES")
