#!/usr/bin/env python
"""Summarise the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py` into the HBM traffic per launch of
the dominant kernel, corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes:
  * both counters are reported in KiB;
  * on gfx950 FETCH_SIZE counts 128-byte requests at 64 bytes: wide streaming reads are DOUBLED;
  * WRITE_SIZE is uncalibrated by the guide -> it is calibrated here against a kernel of known write volume in the
    same trace (the bf16 cast kernel `cast_kernel`, which writes exactly 2 bytes per element) when present.
usage: traffic_summary.py <dir_fetch> <dir_write> <kernel-substring> <out.json> [M,N,K of the GEMM call]"""
import csv
import glob
import json
import statistics
import sys

csv.field_size_limit(1 << 30)


def per_kernel(d, counter):
    out = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == counter:
                    out.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return out


def main():
    dfetch, dwrite, sub, outp = sys.argv[1:5]
    fe, wr = per_kernel(dfetch, "FETCH_SIZE"), per_kernel(dwrite, "WRITE_SIZE")
    names = [k for k in fe if sub in k]
    if not names:
        print("no kernel matching", sub, "among", len(fe)); sys.exit(1)
    name = max(names, key=lambda k: len(fe[k]))
    f_kib = statistics.median(fe[name]); w_kib = statistics.median(wr.get(name, [float("nan")]))
    shape = [int(v) for v in sys.argv[5].split(",")] if len(sys.argv) > 5 else None
    res = {"shape": shape, "kernel": name[:160], "launches_sampled": len(fe[name]), "fetch_size_kib_raw_median": f_kib,
           "write_size_kib_raw_median": w_kib, "fetch_bytes_corrected": f_kib * 1024 * 2, "write_bytes": w_kib * 1024,
           "traffic_bytes_per_launch": f_kib * 1024 * 2 + w_kib * 1024,
           "correction": "FETCH_SIZE[KiB]*1024*2 (gfx950 half-count of wide reads) + WRITE_SIZE[KiB]*1024",
           "all_kernels": {k[:100]: {"n": len(v), "fetch_kib_median": statistics.median(v),
                                      "write_kib_median": statistics.median(wr[k]) if k in wr else None}
                           for k, v in sorted(fe.items(), key=lambda kv: -sum(kv[1]))[:16]}}
    json.dump(res, open(outp, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "all_kernels"}))


if __name__ == "__main__":
    main()
