#!/usr/bin/env python
"""Where a persistent-GEMM launch spends its cycles: k-loop vs epilogue per tile, from the shader clock (wave 0 of every
workgroup), measurement build tools/bin/variants/libprof.so (tools/build_gemm_prof.sh).  Run through tools/gpu_gemm_phase_prof.sh, which puts
that library in the in-tree library's place for the duration of this process (the product has no library override)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-lens_amd"))
import torch  # noqa: E402
from vitlens_hip import ops  # noqa: E402

T = 256 * 256
CASES = [("qkv bf16", (T, 3072, 1024), dict(epi=ops.EPI_BF16)), ("c_fc gelu", (T, 4096, 1024), dict(epi=ops.EPI_BF16, act=ops.ACT_GELU)),
         ("c_fc gelu+dsave", (T, 4096, 1024), dict(epi=ops.EPI_BF16, act=ops.ACT_GELU_DSAVE, out2=True)),
         ("dproj dgelu_saved", (T, 4096, 1024), dict(epi=ops.EPI_DGELU, act=ops.ACT_GELU_DSAVE, res=True)),
         ("c_proj res_bf16", (T, 1024, 4096), dict(epi=ops.EPI_RES_BF16, res=True)), ("out res_bf16", (T, 1024, 1024), dict(epi=ops.EPI_RES_BF16, res=True)),
         ("dfc bf16", (T, 1024, 4096), dict(epi=ops.EPI_BF16)), ("plain fc bf16", (T, 4096, 1024), dict(epi=ops.EPI_BF16))]
lib = ops._lib
buf_sym = ctypes.c_void_p.in_dll(lib, "vl_gemm_prof_buf")
prof = torch.zeros(256 * 4, dtype=torch.int64, device="cuda")
for name, (M, N, K), kw in CASES:
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    bias = torch.randn(N, device="cuda") if kw["epi"] != ops.EPI_DGELU else None
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    aux = torch.randn(M, N, device="cuda").bfloat16()
    k2 = dict(epi=kw["epi"], act=kw.get("act", 0), cfg=8)
    if kw.get("out2"):
        k2["out2"] = torch.empty_like(out)
    if kw.get("res"):
        k2["res"] = aux
    for _ in range(3):
        ops.gemm(a, w, bias, out=out, **k2)
    torch.cuda.synchronize()
    prof.zero_(); buf_sym.value = prof.data_ptr()
    n = 5
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ops.gemm(a, w, bias, out=out, **k2)
    e1.record(); torch.cuda.synchronize()
    buf_sym.value = None
    p = prof.view(256, 4).cpu().double()
    tiles = p[:, 2].sum()
    ms = e0.elapsed_time(e1) / n
    print(f"{name:20s} {ms:7.4f} ms {2.0 * M * N * K / ms / 1e9:7.1f} TF/s | per tile: k-loop {p[:, 0].sum() / tiles:8.0f} clk ({K // 64} k-steps, "
          f"{p[:, 0].sum() / tiles / (K // 64):6.0f} per step)  epilogue {p[:, 1].sum() / tiles:7.0f} clk = {100 * p[:, 1].sum() / (p[:, 0].sum() + p[:, 1].sum()):4.1f} %", flush=True)

# ---- round 4: the LayerNorm-folding variants next to their plain counterparts ----
def _prof(name, fn, M, N, K):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    prof.zero_(); buf_sym.value = prof.data_ptr()
    n = 5
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    buf_sym.value = None
    p = prof.view(256, 4).cpu().double()
    tiles = p[:, 2].sum()
    ms = e0.elapsed_time(e1) / n
    print(f"{name:28s} {ms:7.4f} ms {2.0 * M * N * K / ms / 1e9:7.1f} TF/s | per tile: k-loop {p[:, 0].sum() / tiles:8.0f} clk ({K // 64} k-steps, "
          f"{p[:, 0].sum() / tiles / (K // 64):6.0f} per step)  epilogue {p[:, 1].sum() / tiles:7.0f} clk = {100 * p[:, 1].sum() / (p[:, 0].sum() + p[:, 1].sum()):4.1f} %", flush=True)


if os.environ.get("LNFOLD", "1") != "0":
    D = 1024
    x = torch.randn(T, D, device="cuda").bfloat16()
    gam, bet = 1 + 0.2 * torch.randn(D, device="cuda"), 0.1 * torch.randn(D, device="cuda")
    mean, rstd = torch.empty(T, device="cuda"), torch.empty(T, device="cuda")
    ops.ln_row_stats(None, x, 0, mean, rstd)
    part = torch.empty(T * (D // 64) * 2, device="cuda")
    hws = torch.empty(1, D, device="cuda", dtype=torch.bfloat16)
    for name, N, act in (("qkv", 3072, ops.ACT_NONE), ("c_fc gelu", 4096, ops.ACT_GELU), ("c_fc gelu+dsave", 4096, ops.ACT_GELU_DSAVE)):
        w = torch.randn(N, D, device="cuda") * D ** -0.5; b = 0.02 * torch.randn(N, device="cuda")
        f = ops.fold_ln_linear(w, b, gam, bet); w16 = w.bfloat16()
        out = torch.empty(T, N, device="cuda", dtype=torch.bfloat16)
        out2 = torch.empty_like(out) if act == ops.ACT_GELU_DSAVE else None
        _prof(name + " plain", lambda: ops.gemm(x, w16, b, out=out, epi=ops.EPI_BF16, act=act, cfg=8, out2=out2), T, N, D)
        _prof(name + " LN-folded", lambda: ops.gemm_lnfold(x, f, mean, rstd, out, w16, b, gam, bet, hws, act=act, out2=out2), T, N, D)
    for name, K in (("out + res", 1024), ("c_proj + res", 4096)):
        a = torch.randn(T, K, device="cuda").bfloat16(); w = (torch.randn(D, K, device="cuda") * K ** -0.5).bfloat16(); b = 0.02 * torch.randn(D, device="cuda")
        xo = torch.empty_like(x)
        _prof(name + " plain", lambda: ops.gemm(a, w, b, out=xo, res=x, epi=ops.EPI_RES_BF16, cfg=8), T, D, K)
        _prof(name + " + row sums", lambda: ops.gemm_res_rowstats(a, w, b, xo, x, part), T, D, K)


# ---- round 5: what makes the k-step of the K = 4096 launches 15 % slower than at K = 1024?  (KSTEP=1 python tools/gemm_phase_prof.py) ----
if os.environ.get("KSTEP", "0") != "0":
    def case(name, M, N, K, pad=0):
        buf = torch.empty(M, K + pad, device="cuda", dtype=torch.bfloat16); a = buf[:, :K]; a.copy_(torch.randn(M, K, device="cuda").bfloat16())
        wb = torch.empty(N, K + pad, device="cuda", dtype=torch.bfloat16); w = wb[:, :K]; w.copy_((torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16())
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        _prof(name, lambda: ops.gemm(a, w, None, out=out, epi=ops.EPI_BF16, cfg=8), M, N, K)
    case("K=1024 N=1024", T, 1024, 1024)
    case("K=2048 N=1024", T, 1024, 2048)
    case("K=4096 N=1024", T, 1024, 4096)
    case("K=4096 N=1024 rows + 64", T, 1024, 4096, pad=64)
    case("K=4096 N=4096", T, 4096, 4096)
    case("K=4096 N=1024 M=16384 (A in MALL)", 16384, 1024, 4096)
    case("K=8192 N=1024 M=16384", 16384, 1024, 8192)
    case("K=1024 N=4096", T, 4096, 1024)
