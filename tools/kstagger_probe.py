#!/usr/bin/env python
"""k-rotation of the persistent GEMM (vl_gemm_park.hip, PK_KSTAG_*): the ViT-L block shapes with the rotation off / by row tile /
by workgroup slot and several offsets, interleaved rounds in one process, N(0,1) operands, HIP-event timing.  Needs the probe
build (tools/build_gemm_probe.sh) in the in-tree library's place: run through tools/gpu_kstagger_probe.sh."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-lens_amd"))
import torch  # noqa: E402
from vitlens_hip import ops  # noqa: E402

knob = (ctypes.c_int * 2).in_dll(ops._lib, "vl_gemm_probe_kstag")
T = 256 * 256
CASES = [("c_proj + res   ", (T, 1024, 4096), "res"), ("dX of c_fc     ", (T, 1024, 4096), "bf16"), ("sq8k           ", (8192, 8192, 8192), "bf16"),
         ("c_fc gelu      ", (T, 4096, 1024), "gelu"), ("qkv            ", (T, 3072, 1024), "bf16"), ("out + res      ", (T, 1024, 1024), "res"),
         ("dX c_proj dgelu", (T, 4096, 1024), "dgelu"), ("c_proj+res 257 ", (257 * 256, 1024, 4096), "res")]
MODES = [(0, 0), (1, 1), (1, 2), (1, 4), (1, 8), (1, 16), (2, 1), (2, 2), (2, 8)]
if os.environ.get("KS_MODES"):
    MODES = [tuple(int(x) for x in m.split(":")) for m in os.environ["KS_MODES"].split(",")]
rounds = int(os.environ.get("KS_ROUNDS", "7"))
for name, (M, N, K), kind in CASES:
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); aux = torch.randn(M, N, device="cuda").bfloat16()
    if kind == "res":
        fn = lambda: ops.gemm(a, w, bias, out=out, res=aux, epi=ops.EPI_RES_BF16)
    elif kind == "gelu":
        fn = lambda: ops.gemm(a, w, bias, out=out, epi=ops.EPI_BF16, act=ops.ACT_GELU)
    elif kind == "dgelu":
        fn = lambda: ops.gemm(a, w, None, out=out, res=aux, epi=ops.EPI_DGELU, act=ops.ACT_GELU_DSAVE)
    else:
        fn = lambda: ops.gemm(a, w, bias, out=out, epi=ops.EPI_BF16)
    ref = None
    ts = {m: [] for m in MODES}
    for r in range(rounds + 1):
        for m in MODES:
            knob[0], knob[1] = m
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); fn(); e1.record(); torch.cuda.synchronize()
            if r:
                ts[m].append(e0.elapsed_time(e1) / 2)
            elif m == MODES[0]:
                ref = out.float().clone()
            else:       # same products, rotated order: fp32 summation noise only (bf16 outputs: mostly identical)
                d = float((out.float() - ref).abs().max() / ref.abs().max())
                assert d < 1e-2, (name, m, d)
    base = None
    for m in MODES:
        v = sorted(ts[m]); med = v[len(v) // 2]
        base = base or med
        print(f"{name} mode {m[0]} step {m[1]:2d}: med {med:7.4f} ms  min {v[0]:7.4f}  {2.0 * M * N * K / med / 1e9:7.1f} TF/s  {100 * (base / med - 1):+5.1f} %", flush=True)
    del a, w, out, aux
