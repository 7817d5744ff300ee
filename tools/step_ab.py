#!/usr/bin/env python
"""Same-box A/B of the C3 step with one product switch flipped from OUTSIDE the product (no debug modes inside it): runs
bench.py's main() in this process twice per variant, alternating, and prints ms/step.  Variants are monkey-patches of the
Python wrappers (e.g. `ops.attn_bwd(..., fused=False)` = the two-kernel attention backward).
usage: python tools/step_ab.py attn_bwd_2k [more variants] -- [bench.py arguments]"""
import functools
import io
import json
import os
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vit-lens_amd"))


def patch(name):
    from vitlens_hip import ops
    if name == "base":
        return lambda: None
    if name == "attn_bwd_2k":
        orig = ops.attn_bwd
        ops.attn_bwd = functools.partial(orig, fused=False)
        return lambda: setattr(ops, "attn_bwd", orig)
    if name == "ln_left_separate":      # round 5: the leftover rows' LayerNorm as its own launch again (what the fused row-statistics launch replaced)
        o_stats, o_fold = ops.ln_row_stats, ops.gemm_lnfold

        def stats(part, x, m_main, mean, rstd, eps=1e-5, **kw):
            return o_stats(part, x, m_main, mean, rstd, eps)

        def fold(*a, **kw):
            kw["h_ready"] = False
            return o_fold(*a, **kw)
        ops.ln_row_stats, ops.gemm_lnfold = stats, fold
        return lambda: (setattr(ops, "ln_row_stats", o_stats), setattr(ops, "gemm_lnfold", o_fold))
    raise SystemExit(f"unknown variant {name}")


def run(bench_args):
    import bench
    sys.argv = ["bench.py"] + bench_args
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.main()
    line = [l for l in buf.getvalue().splitlines() if l.startswith("{")][-1]
    return json.loads(line)


if __name__ == "__main__":
    args = sys.argv[1:]
    cut = args.index("--") if "--" in args else len(args)
    variants, bench_args = ["base"] + args[:cut], args[cut + 1:] or ["--steps", "6", "--warmup", "2", "--no-cpu-baseline"]
    for rep in range(2):
        for v in variants:
            undo = patch(v)
            d = run(bench_args)
            undo()
            print(f"rep {rep} {v:14s} {d['ms_per_step']:9.3f} ms/step  all-GEMM {d['roofline']['all_gemm_tflops']:7.1f} TF/s  step {d['roofline']['step_tflops']:6.1f} TF/s", flush=True)
