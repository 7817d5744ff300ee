#!/usr/bin/env python
"""Where does the per-tile overhead of the K=1024 GEMMs go?  Times the C2 GEMM shapes with the epilogue (a) normal,
(b) LDS transpose + arithmetic but no global stores (mode 4), (c) skipped (mode 2).  Mode 3 (old direct path with
the stores folded onto 256 L2-resident rows) is still available through ops.set_wide_stores(3)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-lens_amd"))
import torch
from vitlens_hip import ops


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2]


def main():
    T = 257 * 256
    for name, M, N, K, epi, act in (("fc+gelu", T, 4096, 1024, ops.EPI_BF16, 1), ("fc", T, 4096, 1024, ops.EPI_BF16, 0),
                                    ("out bf16", T, 1024, 1024, ops.EPI_BF16, 0), ("out res_f32", T, 1024, 1024, ops.EPI_RES_F32, 0),
                                    ("proj res_f32", T, 1024, 4096, ops.EPI_RES_F32, 0)):
        a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
        bias = torch.randn(N, device="cuda")
        out = torch.empty(M, N, device="cuda", dtype=torch.float32 if epi == ops.EPI_RES_F32 else torch.bfloat16)
        res = out if epi == ops.EPI_RES_F32 else None
        row = []
        for mode in (0, 4, 2):
            ops.set_wide_stores(mode)
            row.append(timeit(lambda: ops.gemm(a, w, bias, out=out, res=res, epi=epi, act=act, cfg=int(os.environ.get('GEMM_CFG', 5)))))
        ops.set_wide_stores(0)
        fl = 2.0 * M * N * K
        print(f"{name:13s} normal {row[0]:.3f} ms ({fl / row[0] / 1e9:.0f} TF/s) | transpose+math, no stores {row[1]:.3f} ms | no epilogue {row[2]:.3f} ms ({fl / row[2] / 1e9:.0f} TF/s)", flush=True)
        del a, w, out



if __name__ == "__main__":
    main()
