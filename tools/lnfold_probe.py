#!/usr/bin/env python
"""Round 4: the four GEMMs of a frozen ViT-L block at the bench shape (65 792 rows) with the LayerNorms folded in
(ln_row_stats + gemm_lnfold, gemm_res_rowstats) against the plain sequence (layernorm + gemm, gemm + residual)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-lens_amd"))
import torch
from vitlens_hip import ops

M, D = 257 * 256, 1024
g = torch.Generator().manual_seed(0)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc)
x = r(M, D).bfloat16().cuda()
gam, bet = (1 + 0.2 * r(D)).cuda(), (0.1 * r(D)).cuda()
w_in, b_in = r(3 * D, D, sc=D ** -0.5).cuda(), r(3 * D, sc=0.02).cuda()
w_fc, b_fc = r(4 * D, D, sc=D ** -0.5).cuda(), r(4 * D, sc=0.02).cuda()
w_out, b_out = r(D, D, sc=D ** -0.5).bfloat16().cuda(), r(D, sc=0.02).cuda()
w_pr, b_pr = r(D, 4 * D, sc=(4 * D) ** -0.5).bfloat16().cuda(), r(D, sc=0.02).cuda()
f_in, f_fc = ops.fold_ln_linear(w_in, b_in, gam, bet), ops.fold_ln_linear(w_fc, b_fc, gam, bet)
w_in16, w_fc16 = w_in.bfloat16(), w_fc.bfloat16()
h = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
qkv = torch.empty(M, 3 * D, device="cuda", dtype=torch.bfloat16)
hid = torch.empty(M, 4 * D, device="cuda", dtype=torch.bfloat16)
u = torch.empty(M, 4 * D, device="cuda", dtype=torch.bfloat16)
a = r(M, D).bfloat16().cuda()
xo = torch.empty_like(x)
mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
part = torch.empty(M * (D // 64) * 2, device="cuda")
n = int(os.environ.get("N", 10))


def t(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


mm = ops.gemm_res_rowstats(a, w_out, b_out, xo, x, part)
rows = [
    ("ln_1 + in-projection (plain)", lambda: (ops.layernorm(x, gam, bet, h, M, D), ops.gemm(h, w_in16, b_in, out=qkv))),
    ("row stats + folded in-projection", lambda: (ops.ln_row_stats(part, x, mm, mean, rstd), ops.gemm_lnfold(x, f_in, mean, rstd, qkv, w_in16, b_in, gam, bet, h))),
    ("ln_2 + c_fc + GELU (plain)", lambda: (ops.layernorm(x, gam, bet, h, M, D), ops.gemm(h, w_fc16, b_fc, out=hid, act=ops.ACT_GELU))),
    ("row stats + folded c_fc + GELU", lambda: (ops.ln_row_stats(part, x, mm, mean, rstd), ops.gemm_lnfold(x, f_fc, mean, rstd, hid, w_fc16, b_fc, gam, bet, h, act=ops.ACT_GELU))),
    ("ln_2 + c_fc + GELU + gelu' (plain)", lambda: (ops.layernorm(x, gam, bet, h, M, D), ops.gemm(h, w_fc16, b_fc, out=hid, act=ops.ACT_GELU_DSAVE, out2=u))),
    ("row stats + folded c_fc + GELU + gelu'", lambda: (ops.ln_row_stats(part, x, mm, mean, rstd), ops.gemm_lnfold(x, f_fc, mean, rstd, hid, w_fc16, b_fc, gam, bet, h, act=ops.ACT_GELU_DSAVE, out2=u))),
    ("out-projection + residual (plain)", lambda: ops.gemm(a, w_out, b_out, out=xo, res=x, epi=ops.EPI_RES_BF16)),
    ("out-projection + residual + row sums", lambda: ops.gemm_res_rowstats(a, w_out, b_out, xo, x, part)),
    ("c_proj + residual (plain)", lambda: ops.gemm(hid, w_pr, b_pr, out=xo, res=x, epi=ops.EPI_RES_BF16)),
    ("c_proj + residual + row sums", lambda: ops.gemm_res_rowstats(hid, w_pr, b_pr, xo, x, part)),
    ("layernorm alone", lambda: ops.layernorm(x, gam, bet, h, M, D)),
    ("row stats from partial sums alone", lambda: ops.ln_row_stats(part, x, mm, mean, rstd)),
    ("row stats from the rows alone", lambda: ops.ln_row_stats(part, x, 0, mean, rstd)),
]
for _ in range(2):
    for name, fn in rows:
        print("%-42s %.4f ms" % (name, t(fn)))
    print()
