#!/usr/bin/env python
"""Target of the rocprofv3 --pmc passes that compare the vendor GEMM with this repository's persistent kernel, counter by
counter, on the shapes where they differ most (tools/gpu_r06_visit4.sh; summary by tools/pmc_summary.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-lens_amd"))
import torch
from vitlens_hip import ops
T = 256 * 256
for name, M, N, K in (("proj", T, 1024, 4096), ("fc", T, 4096, 1024), ("sq8k", 8192, 8192, 8192)):
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); out2 = torch.empty_like(out)
    wt = w.t()
    for _ in range(3):
        torch.matmul(a, wt, out=out2)
        ops.gemm(a, w, None, out=out, epi=ops.EPI_BF16)
    torch.cuda.synchronize()
    del a, w, out, out2
