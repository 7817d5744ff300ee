#!/usr/bin/env python
"""Round 4: kNN grouping of the point tokenizer at the C5 shape (128 clouds x 8192 points, 512 centres, k = 32):
the LDS-staged kernel (aligned cloud pointer) against the direct one (a pointer 4 bytes off a 16-byte boundary)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-lens_amd"))
import torch
from vitlens_hip import ops

B, N, G, k = 128, 8192, 512, 32
g = torch.Generator().manual_seed(0)
pts = (torch.rand(B, N, 3, generator=g) * 2 - 1).cuda()
start = torch.zeros(B, dtype=torch.long, device="cuda")
cidx, _ = ops.fps(pts, start, G)
buf = torch.empty(B * N * 3 + 1, device="cuda")
mis = buf[1:].view(B, N, 3); mis.copy_(pts)


def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


a = ops.knn_group(pts, cidx, k, Kp=64, want_idx=True)
b = ops.knn_group(mis, cidx, k, Kp=64, want_idx=True)
print("identical:", torch.equal(a[1], b[1]) and torch.equal(a[0].view(torch.int16), b[0].view(torch.int16)))
print("knn LDS-staged  %.3f ms" % t(lambda: ops.knn_group(pts, cidx, k, Kp=64)))
print("knn direct      %.3f ms" % t(lambda: ops.knn_group(mis, cidx, k, Kp=64)))
