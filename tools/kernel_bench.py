#!/usr/bin/env python
"""Per-kernel micro-benchmarks on the GPU box (HIP-event timing, interleaved variants, random data).
Writes gpurun_out/kernel_bench.json.  Usage: python tools/kernel_bench.py [--quick]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-lens_amd"))
import torch  # noqa: E402
from vitlens_hip import ops  # noqa: E402


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    quick = "--quick" in sys.argv
    res = {"device": ops.device_info(0), "gemm": [], "attn": [], "rows": []}
    T = 257 * 256
    shapes = [("qkv", T, 3072, 1024), ("out", T, 1024, 1024), ("fc", T, 4096, 1024), ("proj", T, 1024, 4096),
              ("sq8k", 8192, 8192, 8192)]
    cfgs = [int(c) for c in os.environ.get("KB_CFGS", "5,8").split(",")]
    for name, M, N, K in shapes:
        a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
        bias = torch.randn(N, device="cuda")
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        resb = torch.randn(M, N, device="cuda").bfloat16()
        u2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)

        def variant(label, cfg, av, ov, rv, uv):
            if label == "bf16":
                return lambda: ops.gemm(av, w, bias, out=ov, epi=ops.EPI_BF16, cfg=cfg)
            if label == "bf16+gelu":
                return lambda: ops.gemm(av, w, bias, out=ov, epi=ops.EPI_BF16, act=ops.ACT_GELU, cfg=cfg)
            if label == "gelu+save":
                return lambda: ops.gemm(av, w, bias, out=ov, epi=ops.EPI_BF16, act=ops.ACT_GELU, cfg=cfg, out2=uv)
            if label == "res_bf16":
                return lambda: ops.gemm(av, w, bias, out=rv, res=rv, epi=ops.EPI_RES_BF16, cfg=cfg)
            return lambda: ops.gemm(av, w, None, out=ov, res=rv, epi=ops.EPI_DGELU, cfg=cfg)

        for label in ("bf16", "bf16+gelu", "gelu+save", "res_bf16", "dgelu"):
            if quick and label in ("bf16+gelu", "gelu+save", "dgelu") and name not in ("fc",):
                continue
            if quick and name.startswith("sq") and label != "bf16":
                continue
            for cfg in cfgs:
                Mq = M // 256 * 256 if cfg == 8 else M       # the 4-wave kernel takes whole tiles (auto dispatch adds the tail kernel)
                med, mn = timeit(variant(label, cfg, a[:Mq], out[:Mq], resb[:Mq], u2[:Mq]))
                tf = 2.0 * Mq * N * K / (med * 1e-3) / 1e12
                res["gemm"].append({"shape": name, "M": Mq, "N": N, "K": K, "cfg": cfg, "epi": label, "ms": med, "min_ms": mn, "tflops": tf})
                print(f"gemm {name:5s} cfg{cfg:2d} {label:10s} M={Mq} {med:8.3f} ms  {tf:7.1f} TF/s", flush=True)
        del a, w, out, resb, u2
    # in-projection + attention at the bench shape (q / k / v read in place from the packed projection output)
    B, L, H, dh = 256, 257, 16, 64
    D = H * dh
    x = torch.randn(B * L, D, device="cuda").bfloat16(); w = (torch.randn(3 * D, D, device="cuda") * D ** -0.5).bfloat16()
    bias = torch.randn(3 * D, device="cuda")
    qkv = torch.empty(B * L, 3 * D, device="cuda", dtype=torch.bfloat16)
    o = torch.empty(B * L, D, device="cuda", dtype=torch.bfloat16)
    med, mn = timeit(lambda: ops.gemm(x, w, bias, out=qkv, epi=ops.EPI_BF16))
    print(f"in-projection {med:8.3f} ms {2.0 * B * L * 3 * D * D / med / 1e9:7.1f} TF/s", flush=True)
    res["gemm"].append({"shape": "in_proj", "cfg": -1, "ms": med, "tflops": 2.0 * B * L * 3 * D * D / med / 1e9})
    q, k, v = (ops.heads_view(qkv, B, L, H, dh, i * D) for i in range(3))
    med, mn = timeit(lambda: ops.attn_fwd(q, k, v, o, qscale=dh ** -0.5 * ops.LOG2E))
    fl = 4.0 * B * H * L * L * dh
    print(f"attn fwd {med:8.3f} ms {fl / med / 1e9:7.1f} TF/s", flush=True)
    res["attn"].append({"B": B, "L": L, "H": H, "dh": dh, "ms": med, "tflops": fl / med / 1e9})
    # layernorm
    for dt, nm in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
        xr = torch.randn(B * L, D, device="cuda").to(dt)
        wln = torch.ones(D, device="cuda"); bln = torch.zeros(D, device="cuda")
        med, mn = timeit(lambda: ops.layernorm(xr, wln, bln, o, B * L, D))
        gb = (xr.numel() * xr.element_size() + o.numel() * 2) / 1e9
        print(f"layernorm {nm} {med:8.3f} ms {gb / med * 1e3:7.1f} GB/s", flush=True)
        res["rows"].append({"op": "layernorm_" + nm, "ms": med, "gbps": gb / med * 1e3})
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "kernel_bench.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
