#!/usr/bin/env python
"""MLP of one ViT-L block at the bench shape (65 536 rows): c_fc + GELU -> c_proj + residual as two whole launches (hid = 539 MB:
written to and re-read from HBM) against row chunks whose hid slice stays in the 256 MB Infinity Cache between the two launches
(rolling buffer).  With and without the gelu' output of the trained tower (another 539 MB written), and the backward pair
(dX through gelu' -> dX of c_fc)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-lens_amd"))
import torch
from vitlens_hip import ops

T, D, Hd = 65536, 1024, 4096
x = torch.randn(T, D, device="cuda").bfloat16()
wfc = (torch.randn(Hd, D, device="cuda") * D ** -0.5).bfloat16(); bfc = torch.randn(Hd, device="cuda")
wpr = (torch.randn(D, Hd, device="cuda") * Hd ** -0.5).bfloat16(); bpr = torch.randn(D, device="cuda")
hid = torch.empty(T, Hd, device="cuda", dtype=torch.bfloat16)
u = torch.empty(T, Hd, device="cuda", dtype=torch.bfloat16)
xo = torch.empty_like(x)
dxb = torch.randn(T, D, device="cuda").bfloat16(); du = torch.empty(T, Hd, device="cuda", dtype=torch.bfloat16); dh = torch.empty_like(x)
wprT = wpr.t().contiguous(); wfcT = wfc.t().contiguous()


def fwd(chunk, dsave):
    for r0 in range(0, T, chunk):
        r1 = r0 + chunk
        h = hid[:chunk] if chunk < T else hid
        if dsave:
            ops.gemm(x[r0:r1], wfc, bfc, out=h, epi=ops.EPI_BF16, act=ops.ACT_GELU_DSAVE, cfg=8, out2=u[r0:r1])
        else:
            ops.gemm(x[r0:r1], wfc, bfc, out=h, epi=ops.EPI_BF16, act=ops.ACT_GELU, cfg=8)
        ops.gemm(h, wpr, bpr, out=xo[r0:r1], res=x[r0:r1], epi=ops.EPI_RES_BF16, cfg=8)


def bwd(chunk):
    for r0 in range(0, T, chunk):
        r1 = r0 + chunk
        d = du[:chunk] if chunk < T else du
        ops.gemm(dxb[r0:r1], wprT, None, out=d, res=u[r0:r1], epi=ops.EPI_DGELU, act=ops.ACT_GELU_DSAVE, cfg=8)
        ops.gemm(d, wfcT, None, out=dh[r0:r1], epi=ops.EPI_BF16, cfg=8)


def bench(fn, n=9):
    fn(); fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 2)
    ts.sort()
    return ts[len(ts) // 2]


for rounds in range(2):
    for name, fn in (("fwd c_fc+GELU -> c_proj+res", lambda c: fwd(c, False)), ("fwd c_fc+GELU+gelu' -> c_proj+res", lambda c: fwd(c, True)),
                     ("bwd dX(gelu') -> dX(c_fc)", bwd)):
        for chunk in (T, 32768, 16384):
            ms = bench(lambda: fn(chunk))
            print(f"{name:36s} rows per launch {chunk:6d}: {ms:7.4f} ms per 65 536 rows", flush=True)
