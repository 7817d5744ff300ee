#!/usr/bin/env python
"""Vendor yardstick (round 6): what does the vendor GEMM (`torch.matmul` on bf16 = hipBLASLt /
rocBLAS Tensile kernels) reach on THIS board, on the five shapes the C3 step is made of, with the SAME N(0,1) operands and
the same timing harness as this repository's persistent kernel?  Never imported by the product: `vit-lens_amd/` contains no
BLAS call; this script exists to price the "power wall" claim of DESIGN.md 7.1.

Per shape, interleaved rounds in one process (so both contenders see the same clocks / temperature):
    vendor   : torch.matmul(a, w.t())                       (bf16 in, bf16 out, fp32 accumulate)
    ours     : ops.gemm(a, w, None, out, epi=EPI_BF16)      (persistent 256x256 kernel + leftover-row launch)
Also the M = 65 792 = 257 x 256 row count of the step (does the vendor kernel pay a partial-wave tax on 257 row tiles?).

    python tools/vendor_gemm_yardstick.py                    # table on stdout
    YS_SUSTAIN=1 ...                                         # additionally a 6 s sustained loop per contender (power-limited rate)
Kernel names of the vendor side: run under `rocprofv3 --kernel-trace --stats` (tools/gpu_r06_visit1.sh does)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-lens_amd"))
import torch  # noqa: E402
from vitlens_hip import ops  # noqa: E402

T = 256 * 256
SHAPES = [("fc     ", T, 4096, 1024), ("proj   ", T, 1024, 4096), ("qkv    ", T, 3072, 1024), ("out    ", T, 1024, 1024),
          ("sq8k   ", 8192, 8192, 8192),
          ("fc257  ", 257 * 256, 4096, 1024), ("proj257", 257 * 256, 1024, 4096), ("out257 ", 257 * 256, 1024, 1024)]


def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    rounds = int(os.environ.get("YS_ROUNDS", "9"))
    reps = int(os.environ.get("YS_REPS", "4"))
    sustain = os.environ.get("YS_SUSTAIN", "0") == "1"
    print(f"# torch {torch.__version__}  device {torch.cuda.get_device_name(0)}  rounds {rounds} x {reps} launches, N(0,1) bf16 operands")
    print(f"# {'shape':8s} {'M':>6s} {'N':>5s} {'K':>5s} | vendor med ms / TF/s (min ms) | ours med ms / TF/s (min ms) | ours/vendor")
    for name, M, N, K in SHAPES:
        a = torch.randn(M, K, device="cuda").bfloat16()
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
        wt = w.t()                                              # [K, N] view: matmul(a, wt) = a @ w^T, the NT problem both sides solve
        out_v = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        out_o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        fv = lambda: torch.matmul(a, wt, out=out_v)
        fo = lambda: ops.gemm(a, w, None, out=out_o, epi=ops.EPI_BF16)
        for f in (fv, fo):
            f(); f()
        torch.cuda.synchronize()
        err = float((out_v.float() - out_o.float()).abs().max() / out_v.float().abs().max())
        tv, to = [], []
        for _ in range(rounds):
            tv.append(timed(fv, reps)); to.append(timed(fo, reps))
        tv.sort(); to.sort()
        mv, mo = tv[len(tv) // 2], to[len(to) // 2]
        fl = 2.0 * M * N * K
        line = (f"  {name:8s} {M:6d} {N:5d} {K:5d} | {mv:7.4f} {fl / mv / 1e9:7.1f} ({tv[0]:.4f}) | {mo:7.4f} {fl / mo / 1e9:7.1f} ({to[0]:.4f}) | "
                f"{mv / mo:5.3f}  maxrel {err:.1e}")
        if sustain:
            res = []
            for f in (fv, fo):
                t0 = time.time(); n = 0
                while time.time() - t0 < 6.0:
                    for _ in range(50):
                        f()
                    torch.cuda.synchronize(); n += 50
                res.append(fl * n / (time.time() - t0) / 1e12)
            line += f" | sustained 6 s: vendor {res[0]:7.1f} ours {res[1]:7.1f} TF/s"
        print(line, flush=True)
        del a, w, out_v, out_o


if __name__ == "__main__":
    main()
