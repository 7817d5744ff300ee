#!/usr/bin/env python
"""The K >= 2048 launches of the C3 step on the one-wave-per-SIMD kernel (cfg 14, vl_gemm_w4.hip) next to the 8-wave kernel
(cfg 8) and the vendor GEMM (torch.matmul, plain epilogue only), interleaved rounds in one process, N(0,1) operands, plus a
6 s sustained loop per contender (the board is power-limited: DESIGN.md 7.1)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-lens_amd"))
import torch
from vitlens_hip import ops

T = 256 * 256
CASES = [("c_proj + res      ", T, 1024, 4096, "res"), ("c_proj + res+stats", T, 1024, 4096, "stats"), ("dX of c_fc        ", T, 1024, 4096, "bf16"),
         ("sq8k              ", 8192, 8192, 8192, "bf16"), ("K = 2048          ", T, 1024, 2048, "bf16"), ("text c_proj shape ", 77 * 256, 768, 3072, "bf16")]


def timed(fn, reps=4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, M, N, K, kind in CASES:
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    bias = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    res = torch.randn(M, N, device="cuda").bfloat16(); part = torch.empty(M * (N // 64) * 2, device="cuda")
    fns = {}
    if kind == "bf16":
        for c in (8, 14):
            fns[f"cfg{c}"] = (lambda c=c: ops.gemm(a, w, bias, out=out, epi=ops.EPI_BF16, cfg=c))
        wt = w.t(); ov = torch.empty_like(out)
        fns["vendor"] = lambda: torch.matmul(a, wt, out=ov)
    elif kind == "res":
        for c in (8, 14):
            fns[f"cfg{c}"] = (lambda c=c: ops.gemm(a, w, bias, out=out, res=res, epi=ops.EPI_RES_BF16, cfg=c))
    else:
        fns["auto(w4)"] = lambda: ops.gemm_res_rowstats(a, w, bias, out, res, part)
        fns["cfg8 plain res"] = lambda: ops.gemm(a, w, bias, out=out, res=res, epi=ops.EPI_RES_BF16, cfg=8)
    for f in fns.values():
        f(); f()
    torch.cuda.synchronize()
    ts = {k: [] for k in fns}
    for _ in range(9):
        for k, f in fns.items():
            ts[k].append(timed(f))
    fl = 2.0 * M * N * K
    line = f"{name} "
    for k in fns:
        v = sorted(ts[k]); med = v[len(v) // 2]
        line += f"| {k}: {med:.4f} ms {fl / med / 1e9:7.1f} TF/s "
    if os.environ.get("SUSTAIN", "1") == "1":
        for k, f in fns.items():
            t0 = time.time(); n = 0
            while time.time() - t0 < 5.0:
                for _ in range(50):
                    f()
                torch.cuda.synchronize(); n += 50
            line += f"| sustained {k}: {fl * n / (time.time() - t0) / 1e12:7.1f} "
    print(line, flush=True)
    del a, w, out, res
