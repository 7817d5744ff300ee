"""Weight-gradient GEMMs of one unlocked ViT-L block at the C3 micro-batch (65 792 tokens): the transposing path
(transpose64 of both operands, bias column sum fused into the first, NT split-K GEMM) against the token-major path
(vl_gemm_tn_splitk_accum_f32 + separate column sum).  usage: python tools/dw_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vit-lens_amd"))
from vitlens_hip import ops  # noqa: E402


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    R = 257 * 256
    for name, M, N in (("c_fc", 4096, 1024), ("c_proj", 1024, 4096), ("in_proj", 3072, 1024)):
        pad = int(os.environ.get("DW_PAD", "0"))          # extra columns per row: row strides off the power of two
        dy = torch.randn(R, M + pad, device="cuda").bfloat16()[:, :M]; x = torch.randn(R, N + pad, device="cuda").bfloat16()[:, :N]
        g = torch.zeros(M, N, device="cuda"); gb = torch.zeros(M, device="cuda")

        def old():
            dyt = ops.transpose_colsum(dy, R, colsum_out=gb)
            xt = ops.transpose_colsum(x, R)
            ops.gemm_dw(dyt, xt, g)

        def new():
            assert ops.gemm_dw_tn(dy, x, g)
            ops.colsum(dy, gb)

        def new_gemm_only():
            ops.gemm_dw_tn(dy, x, g)
        dyt, xt = ops.transpose_to_bf16(dy, ldo=R), ops.transpose_to_bf16(x, ldo=R)

        def colsum_only():
            ops.colsum(dy, gb)

        def old_gemm_only():
            ops.gemm_dw(dyt, xt, g)
        if os.environ.get("DW_ONLY"):       # for rocprofv3 --pmc passes: a few launches of one GEMM variant
            for _ in range(4):
                (new_gemm_only if os.environ["DW_ONLY"] == "tn" else old_gemm_only)()
            torch.cuda.synchronize()
            continue
        fl = 2.0 * R * M * N
        print(f"{name:8s} dW [{M}x{N}], {R} tokens: transposes + NT {timeit(old):.3f} ms | TN + colsum {timeit(new):.3f} ms | "
              f"GEMM alone NT {timeit(old_gemm_only):.3f} ms ({fl / timeit(old_gemm_only) / 1e9:.0f} TF/s)  TN {timeit(new_gemm_only):.3f} ms "
              f"({fl / timeit(new_gemm_only) / 1e9:.0f} TF/s) | colsum {timeit(colsum_only):.3f} ms", flush=True)


if __name__ == "__main__":
    main()
