#!/usr/bin/env python
"""Does the row stride of the K = 4096 operands cost the persistent GEMM its k-loop rate?  (The phase counters show 2 830-2 940
cycles per k-step for the K = 4096 launches against 2 466 for K = 1024: rows of A and W 8 KB apart may alias in the L2 / fabric
channel interleave.)  c_proj + residual and the dX of c_fc with the operands' rows padded by 64 / 128 / 192 / 320 elements."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-lens_amd"))
import torch
from vitlens_hip import ops

T = 65536


def padded(rows, cols, pad, scale=1.0):
    buf = torch.empty(rows, cols + pad, device="cuda", dtype=torch.bfloat16)
    v = buf[:, :cols]
    v.copy_((torch.randn(rows, cols, device="cuda") * scale).bfloat16())
    return v


def bench(fn, n=7):
    fn(); fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 2)
    ts.sort()
    return ts[len(ts) // 2]


for name, (M, N, K) in (("c_proj + residual", (T, 1024, 4096)), ("dX of c_fc", (T, 1024, 4096)), ("qkv (K = 1024, for reference)", (T, 3072, 1024))):
    bias = torch.randn(N, device="cuda")
    for pa, pw in ((0, 0), (64, 0), (0, 64), (64, 64), (128, 128), (192, 192), (320, 320)):
        a = padded(M, K, pa); w = padded(N, K, pw, K ** -0.5)
        res = torch.randn(M, N, device="cuda").bfloat16()
        if name.startswith("c_proj"):
            fn = lambda: ops.gemm(a, w, bias, out=res, res=res, epi=ops.EPI_RES_BF16, cfg=8)
        else:
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            fn = lambda: ops.gemm(a, w, None, out=out, epi=ops.EPI_BF16, cfg=8)
        ms = bench(fn)
        print(f"{name:32s} lda = K + {pa:3d}  ldw = K + {pw:3d}: {ms:7.4f} ms  {2.0 * M * N * K / ms / 1e9:7.1f} TF/s", flush=True)
        del a, w
