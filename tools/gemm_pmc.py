#!/usr/bin/env python
"""Run GEMM kernel variants a few times on the fc / sq8k shapes (for rocprofv3 --pmc passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-lens_amd"))
import torch
from vitlens_hip import ops
T = 257 * 256
for name, M, N, K in (("fc", T, 4096, 1024), ("sq8k", 8192, 8192, 8192)):
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for cfg in (5, 1):
        for _ in range(2):
            ops.gemm(a, w, None, out=out, epi=ops.EPI_BF16, cfg=cfg)
    torch.cuda.synchronize()
