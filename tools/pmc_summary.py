#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes (SQ counters) over tools/gemm_probe.py per GEMM kernel instantiation.
usage: pmc_summary.py "<glob of pass directories>" out.json"""
import collections
import csv
import glob
import json
import re
import sys

csv.field_size_limit(1 << 30)


def short(name):
    m = re.search(r"(gemm_nt_pk4?_kernel<[^>]*>|gemm_tn_kernel[^(]*|attn_\w+_kernel(<[^>]*>)?)", name)
    if m:
        return m.group(1)
    m = re.search(r"(Cijk_\w*MT\d+x\d+x\d+\w*)", name)          # vendor (Tensile) kernels: tools/vendor_gemm_yardstick.py
    return ("vendor:" + m.group(1)[-60:]) if m else None


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in glob.glob(sys.argv[1]):
        for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
            for r in csv.DictReader(open(f, newline="")):
                k = short(r["Kernel_Name"])
                if k:
                    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, d in sorted(agg.items()):
        m = {c: sum(v) / len(v) for c, v in d.items()}
        if "SQ_WAVE_CYCLES" not in m:
            continue
        waves = m.get("SQ_WAVES", 2048.0)
        cyc = 4.0 * m["SQ_WAVE_CYCLES"] / waves
        e = {"waves": round(waves), "kernel_cycles": round(cyc), "launches": len(d["SQ_WAVE_CYCLES"]),
             "wait_any_frac": round(m.get("SQ_WAIT_ANY", 0.0) / m["SQ_WAVE_CYCLES"], 4),
             "wait_inst_any_frac": round(m.get("SQ_WAIT_INST_ANY", 0.0) / m["SQ_WAVE_CYCLES"], 4)}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            e["mfma_busy_frac"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), 4)
        for c, n in (("SQ_INSTS_VALU", "valu_insts"), ("SQ_INSTS_LDS", "lds_insts"), ("SQ_INSTS_VMEM", "vmem_insts"),
                     ("SQ_INSTS_VALU_MFMA_MOPS_BF16", "mfma_mops_bf16"), ("SQ_LDS_BANK_CONFLICT", "lds_bank_conflict"),
                     ("SQ_LDS_IDX_ACTIVE", "lds_idx_active")):
            if c in m:
                e[n] = round(m[c])
        if "lds_insts" in e and "mfma_mops_bf16" in e and e["mfma_mops_bf16"]:
            e["lds_insts_per_kmop"] = round(1000.0 * e["lds_insts"] / e["mfma_mops_bf16"], 3)
        out[k] = e
        print(k, e)
    json.dump({"kernels": out, "source": "rocprofv3 --pmc passes over tools/gemm_probe.py, summarised by tools/pmc_summary.py"}, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
