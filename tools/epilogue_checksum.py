"""Checksums of every GELU-carrying GEMM epilogue on fixed seeded operands - run once per library variant (tools/lib_ab.sh):
equal lines = bit-identical arithmetic.  Not imported by the product."""
import hashlib
import sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vit-lens_amd"))
import torch
from vitlens_hip import ops

def h(*ts):
    m = hashlib.sha256()
    for t in ts:
        m.update(t.detach().contiguous().cpu().view(torch.uint8).numpy().tobytes())
    return m.hexdigest()[:16]

g = torch.Generator().manual_seed(7)
for (M, N, K) in [(512, 512, 512), (1024, 1024, 1024), (512, 384, 256), (192, 320, 64)]:   # persistent kernel (K >= 512), ping-pong kernel (N % 256 == 128), generic
    a = (torch.randn(M, K, generator=g) * 0.6).bfloat16().cuda()
    w = (torch.randn(N, K, generator=g) * 0.3).bfloat16().cuda()
    b = torch.randn(N, generator=g).cuda()
    tag = f"{M}x{N}x{K}"
    print(tag, "gelu", h(ops.gemm(a, w, b, act=ops.ACT_GELU)))
    o2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    o = ops.gemm(a, w, b, act=ops.ACT_GELU_DSAVE, out2=o2)
    print(tag, "gelu+dsave", h(o, o2))
    o2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    o = ops.gemm(a, w, b, act=3, out2=o2)                    # ACT 3: gelu + the pre-activation copy
    print(tag, "gelu+pre", h(o, o2))
    pre = (torch.randn(M, N, generator=g) * 2.5).bfloat16().cuda()
    print(tag, "dgelu", h(ops.gemm(a, w, None, res=pre, epi=ops.EPI_DGELU)))
    if N % 2 == 0:
        hp = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        try:
            o = ops.gemm(a, w, b, epi=ops.EPI_GEGLU, out2=hp)
            print(tag, "geglu", h(o, hp))
        except Exception as e:
            print(tag, "geglu refused", type(e).__name__)
    hs = (torch.randn(M, 2 * N, generator=g) * 2.0).bfloat16().cuda()
    o = torch.empty(M, 2 * N, device="cuda", dtype=torch.bfloat16)
    try:
        ops.gemm(a, w, None, out=o, res=hs, epi=ops.EPI_DGEGLU)
        print(tag, "dgeglu", h(o))
    except Exception as e:
        print(tag, "dgeglu refused", type(e).__name__)
    if M % 256 == 0 and N % 256 == 0:
        a16 = torch.randn(M, 512, generator=g).half().cuda() * 0.5; w16 = torch.randn(N, 512, generator=g).half().cuda() * 0.2
        print(tag, "f16 gelu", h(ops.gemm_f16(a16, w16, b, act=ops.ACT_GELU)))
