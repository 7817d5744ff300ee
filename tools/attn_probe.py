#!/usr/bin/env python
"""Attention forward/backward at the bench shape (B=256, H=16, L=257, dh=64): timing, and a target for rocprofv3 --pmc.
Operands are read in place from the packed in-projection output, as the towers do."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-lens_amd"))
import torch
from vitlens_hip import ops

B, H, dh = 256, 16, 64
L = int(os.environ.get("L", 257))
D = H * dh
torch.manual_seed(0)
x = torch.randn(B * L, D, device="cuda").bfloat16(); w = (torch.randn(3 * D, D, device="cuda") * D ** -0.5).bfloat16()
bias = torch.randn(3 * D, device="cuda")
qkv = torch.empty(B * L, 3 * D, device="cuda", dtype=torch.bfloat16)
ops.gemm(x, w, bias, out=qkv, epi=ops.EPI_BF16)
q, k, v = (ops.heads_view(qkv, B, L, H, dh, i * D) for i in range(3))
if os.environ.get("LAYOUT") == "bhld":          # head-major contiguous copies (each (b,h) matrix = one 32 KB block)
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
o = torch.empty(B * L, D, device="cuda", dtype=torch.bfloat16)
lse = torch.empty(B, H, L, device="cuda")
dO = torch.randn(B * L, D, device="cuda").bfloat16()
dOv = ops.heads_view(dO, B, L, H, dh)
if os.environ.get("LAYOUT") == "bhld":
    dOv = dOv.contiguous()
delta = torch.empty(B, H, L, device="cuda")
dqkv = torch.empty(B * L, 3 * D, device="cuda", dtype=torch.bfloat16)
n = int(os.environ.get("N", 5))
qs = dh ** -0.5 * ops.LOG2E


def t(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


fl = 4.0 * B * H * L * L * dh
ms = t(lambda: ops.attn_fwd(q, k, v, o, lse=lse, qscale=qs))
print(f"L={L} attn fwd   {ms:.3f} ms  {fl / ms / 1e9:.0f} TF/s")
ms = t(lambda: ops.attn_bwd(q, k, v, dOv, ops.heads_view(o, B, L, H, dh), lse, delta,
                            dqkv, dqkv[:, D:], dqkv[:, 2 * D:], 3 * D, 3 * D, fused=False))
print(f"L={L} attn bwd   {ms:.3f} ms  {2.5 * fl / ms / 1e9:.0f} TF/s (dq + delta, dkv kernels)")
ref = dqkv.clone()
if ops._lib.vl_attn_bwd_fused_supported(L, L, dh, 0) and os.environ.get("LAYOUT") != "bhld":
    dqkv.fill_(float("nan"))
    ms = t(lambda: ops.attn_bwd(q, k, v, dOv, ops.heads_view(o, B, L, H, dh), lse, None,
                                dqkv, dqkv[:, D:], dqkv[:, 2 * D:], 3 * D, 3 * D, fused=True))
    err = float((dqkv.float() - ref.float()).norm() / ref.float().norm())
    print(f"L={L} attn bwd   {ms:.3f} ms  {2.5 * fl / ms / 1e9:.0f} TF/s (ONE kernel; vs the two-kernel result: rel L2 {err:.2e})")
