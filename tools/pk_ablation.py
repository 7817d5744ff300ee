#!/usr/bin/env python
"""Where the 8-wave persistent kernel's k-loop goes: timing of cfg 8 on four shapes, run once per library variant by
tools/lib_ab.sh - ablation builds of vl_gemm_park.hip with the k-loop's DMA issue (-DPK_X_NODMA), its barriers (-DPK_X_NOBAR)
or its fragment reads (-DPK_X_NOREAD) compiled out.  Timing only: their results are garbage."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-lens_amd"))
import torch
from vitlens_hip import ops
line = "  "
for name, M, N, K in (("c_fc", 65536, 4096, 1024), ("dX c_fc", 65536, 1024, 4096), ("out", 65536, 1024, 1024), ("sq8k", 8192, 8192, 8192)):
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    f = lambda: ops.gemm(a, w, None, out=out, epi=ops.EPI_BF16, cfg=8)
    f(); f(); torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); f(); f(); f(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 4)
    ts.sort()
    line += f" {name} {2.0 * M * N * K / ts[len(ts) // 2] / 1e9:7.1f}"
    del a, w, out
print(line + "  TF/s", flush=True)
