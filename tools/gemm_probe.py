#!/usr/bin/env python
"""GEMM A/B probe on the GPU box: the ViT-L block shapes x epilogues on the kernel configurations in KB_CFGS
(8 = round-2 persistent 256x256 kernel, 10 = round-3 ping-pong kernel), interleaved rounds in ONE process, random data,
HIP-event timing.  Tuning inputs (VL_GEMM_PF, VL_PP_DELAY) are read by the library once per process: sweep them by
running this script once per value (tools/gpu_r03_gemm.sh).  Prints one line per (case, cfg): median / min ms, TF/s."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-lens_amd"))
import torch  # noqa: E402
from vitlens_hip import ops  # noqa: E402

T = 256 * 256
SHAPES = {"qkv": (T, 3072, 1024), "out": (T, 1024, 1024), "fc": (T, 4096, 1024), "proj": (T, 1024, 4096),
          "sq8k": (8192, 8192, 8192), "dfc": (T, 1024, 4096), "dproj": (T, 4096, 1024),
          # text tower (ViT-L/14 text: width 768, 77 tokens x 256 captions = 77 row tiles: no whole round of 256 workgroups)
          "tqkv": (19712, 2304, 768), "tout": (19712, 768, 768), "tfc": (19712, 3072, 768), "tproj": (19712, 768, 3072)}


def main():
    cfgs = [int(c) for c in os.environ.get("KB_CFGS", "8,10").split(",")]
    cases = os.environ.get("KB_CASES", "fc:bf16,fc:gelu+save,fc:dgelu,proj:res_bf16,qkv:bf16,out:res_bf16,sq8k:bf16").split(",")
    rounds = int(os.environ.get("KB_ROUNDS", "7"))
    tag = os.environ.get("KB_TAG", "")
    for case in cases:
        name, label = case.split(":")
        M, N, K = SHAPES[name]
        a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
        bias = torch.randn(N, device="cuda")
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        resb = torch.randn(M, N, device="cuda").bfloat16()
        u2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        if label == "res_stats":
            part = torch.empty(M * (N // 64) * 2, device="cuda")
        if label.startswith("ln_"):
            gam, bet = 1 + 0.2 * torch.randn(K, device="cuda"), 0.1 * torch.randn(K, device="cuda")
            fold = ops.fold_ln_linear(w.float(), bias, gam, bet)
            mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
            ops.ln_row_stats(None, a, 0, mean, rstd)
            hws = torch.empty(1, K, device="cuda", dtype=torch.bfloat16)

        def fn(cfg):
            if label == "bf16":
                return lambda: ops.gemm(a, w, bias, out=out, epi=ops.EPI_BF16, cfg=cfg)
            if label == "gelu":
                return lambda: ops.gemm(a, w, bias, out=out, epi=ops.EPI_BF16, act=ops.ACT_GELU, cfg=cfg)
            if label == "gelu+save":
                return lambda: ops.gemm(a, w, bias, out=out, epi=ops.EPI_BF16, act=ops.ACT_GELU, cfg=cfg, out2=u2)
            if label == "gelu+dsave":      # forward of a trained MLP: gelu and gelu' (VL_ACT_GELU_DSAVE)
                return lambda: ops.gemm(a, w, bias, out=out, epi=ops.EPI_BF16, act=ops.ACT_GELU_DSAVE, cfg=cfg, out2=u2)
            if label == "dgelu_saved":     # its dX GEMM: the epilogue multiplies by the saved gelu'
                return lambda: ops.gemm(a, w, None, out=out, res=resb, epi=ops.EPI_DGELU, act=ops.ACT_GELU_DSAVE, cfg=cfg)
            if label == "res_bf16":
                return lambda: ops.gemm(a, w, bias, out=resb, res=resb, epi=ops.EPI_RES_BF16, cfg=cfg)
            if label == "res_stats":       # round 4: + the partial row sums for the LayerNorm that follows (ACT 20)
                return lambda: ops.gemm_res_rowstats(a, w, bias, resb, resb, part)
            if label in ("ln_bf16", "ln_gelu", "ln_dsave"):      # round 4: the LayerNorm folded into the epilogue (ACT 10 / 11 / 14)
                act = {"ln_bf16": ops.ACT_NONE, "ln_gelu": ops.ACT_GELU, "ln_dsave": ops.ACT_GELU_DSAVE}[label]
                return lambda: ops.gemm_lnfold(a, fold, mean, rstd, out, w, bias, gam, bet, hws, act=act,
                                               out2=u2 if act == ops.ACT_GELU_DSAVE else None)
            return lambda: ops.gemm(a, w, None, out=out, res=resb, epi=ops.EPI_DGELU, cfg=cfg)
        fns = {c: fn(c) for c in cfgs}
        for f in fns.values():
            f(); f()
        torch.cuda.synchronize()
        ts = {c: [] for c in cfgs}
        for _ in range(rounds):
            for c in cfgs:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fns[c](); fns[c](); e1.record(); torch.cuda.synchronize()
                ts[c].append(e0.elapsed_time(e1) / 2)
        for c in cfgs:
            v = sorted(ts[c]); med = v[len(v) // 2]
            print(f"{tag:14s} {name:5s} {label:10s} cfg{c:3d}  med {med:7.4f} ms  min {v[0]:7.4f}  {2.0 * M * N * K / med / 1e9:7.1f} TF/s", flush=True)
        del a, w, out, resb, u2


if __name__ == "__main__":
    main()
