#!/usr/bin/env python
"""Counters of the vendor GEMM next to this repository's persistent kernel, per shape (passes of tools/yardstick_pmc_target.py:
every shape runs each contender three times, in the order proj, fc, sq8k).  usage: yardstick_pmc_summary.py "<glob of pass dirs>" out.json"""
import collections
import csv
import glob
import json
import re
import sys

csv.field_size_limit(1 << 30)
SHAPES = ["proj 65536x1024x4096", "fc 65536x4096x1024", "sq8k 8192^3"]
FLOPS = [2.0 * 65536 * 1024 * 4096, 2.0 * 65536 * 4096 * 1024, 2.0 * 8192 ** 3]


def who(name):
    if "gemm_nt_pk_kernel" in name or "gemm_nt_w4_kernel" in name:
        return "ours"
    if re.search(r"Cijk_\w*MT\d+x\d+x\d+", name):
        return "vendor"
    return None


agg = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for d in glob.glob(sys.argv[1]):
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
        seen = collections.defaultdict(dict)          # contender -> dispatch id -> ordinal
        for r in csv.DictReader(open(f, newline="")):
            w = who(r["Kernel_Name"])
            if not w:
                continue
            did = r["Dispatch_Id"]
            if did not in seen[w]:
                seen[w][did] = len(seen[w])
            shape = seen[w][did] // 3
            if shape >= len(SHAPES):
                continue
            agg[(w, shape)][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[(w, shape)] = {k: r.get(k) for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size")}
            meta[(w, shape)]["kernel"] = r["Kernel_Name"][:140]
out = {}
for (w, shape), d in sorted(agg.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    e = dict(meta[(w, shape)])
    if "SQ_WAVE_CYCLES" in m and "SQ_WAVES" in m:
        cyc = 4.0 * m["SQ_WAVE_CYCLES"] / m["SQ_WAVES"]
        e.update(waves=round(m["SQ_WAVES"]), kernel_cycles=round(cyc),
                 wait_any_frac=round(m.get("SQ_WAIT_ANY", 0) / m["SQ_WAVE_CYCLES"], 4),
                 wait_inst_any_frac=round(m.get("SQ_WAIT_INST_ANY", 0) / m["SQ_WAVE_CYCLES"], 4),
                 active_inst_frac=round(m.get("SQ_ACTIVE_INST_ANY", 0) / m["SQ_WAVE_CYCLES"], 4))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            e["mfma_busy_frac"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), 4)
            e["mfma_cycles_per_flop_ideal"] = round(FLOPS[shape] / (1024 * 1024) / m["SQ_VALU_MFMA_BUSY_CYCLES"] * 1024, 4)
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SALU", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE",
              "SQ_WAIT_INST_LDS", "SQ_INSTS_SMEM", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum", "TCC_REQ_sum", "GRBM_GUI_ACTIVE", "FETCH_SIZE", "WRITE_SIZE"):
        if c in m:
            e[c] = round(m[c])
    out[f"{SHAPES[shape]} | {w}"] = e
    print(f"{SHAPES[shape]:22s} {w:6s}", {k: v for k, v in e.items() if k != "kernel"})
json.dump(out, open(sys.argv[2], "w"), indent=1)
