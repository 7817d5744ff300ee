#!/usr/bin/env python
"""Effect of the persistent kernel's phase offset (vl_gemm_set_stagger) on the C2 GEMM shapes and on the C2 step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vit-lens_amd"))
import torch
from vitlens_hip import ops
from tools.epi_probe import timeit  # noqa


T = 257 * 256
shapes = (("fc+gelu", T, 4096, 1024, ops.EPI_BF16, 1), ("out res_bf16", T, 1024, 1024, ops.EPI_RES_BF16, 0),
          ("proj res_bf16", T, 1024, 4096, ops.EPI_RES_BF16, 0), ("out res_f32", T, 1024, 1024, ops.EPI_RES_F32, 0))
bufs = []
for name, M, N, K, epi, act in shapes:
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    bias = torch.randn(N, device="cuda")
    out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if epi == ops.EPI_RES_F32 else torch.bfloat16)
    bufs.append((a, w, bias, out))
B, L, H, dh = 256, 257, 16, 64
x = torch.randn(B * L, 1024, device="cuda").bfloat16(); wq = (torch.randn(3072, 1024, device="cuda") / 32).bfloat16(); bq = torch.randn(3072, device="cuda")
q = torch.empty(B, H, L, dh, device="cuda", dtype=torch.bfloat16); k = torch.empty_like(q); vt = torch.zeros(B, H, dh, 264, device="cuda", dtype=torch.bfloat16)
for st in (0, 1, 2, 3, 5, 8):
    ops.set_stagger(st)
    row = []
    for (name, M, N, K, epi, act), (a, w, bias, out) in zip(shapes, bufs):
        res = out if epi != ops.EPI_BF16 else None
        row.append(f"{name} {timeit(lambda: ops.gemm(a, w, bias, out=out, res=res, epi=epi, act=act, cfg=-1)):.3f}")
    row.append(f"qkv {timeit(lambda: ops.gemm_qkv(x, wq, bq, q, k, vt, B, L, H, dh, cfg=-1)):.3f}")
    print(f"stagger {st}: " + " | ".join(row), flush=True)
ops.set_stagger(0)
