"""Leftover-row GEMMs (M = 256 rows of M = 257*b): the split-K 32x32 tail kernel (cfg 9) against 128x128 (1), 64x64 (11) and
128x64 (12) LDS-DMA tiles, per shape of the ViT-L blocks.  usage: python tools/tail_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vit-lens_amd"))
from vitlens_hip import ops  # noqa: E402


def timeit(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    M = int(os.environ.get("M", 256))
    for (N, K, epi, act) in [(4096, 1024, ops.EPI_BF16, ops.ACT_GELU), (1024, 4096, ops.EPI_RES_BF16, 0), (3072, 1024, ops.EPI_BF16, 0),
                             (1024, 1024, ops.EPI_RES_BF16, 0), (4096, 1024, ops.EPI_DGELU, 0), (1024, 3072, ops.EPI_BF16, 0)]:
        a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
        res = torch.randn(M, N, device="cuda").bfloat16() if epi != ops.EPI_BF16 else None
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        line = f"M={M} N={N} K={K} epi={epi} act={act}:"
        ref = None
        for cfg in (9, 1, 11, 12):
            f = lambda: ops.gemm(a, w, None, out=out, res=res, epi=epi, act=act, cfg=cfg)
            f(); torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
            err = float((out.float() - ref.float()).norm() / ref.float().norm())
            line += f"  cfg{cfg} {timeit(f):6.1f}us (d {err:.1e})"
        print(line, flush=True)


if __name__ == "__main__":
    main()
