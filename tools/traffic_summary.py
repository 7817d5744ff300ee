#!/usr/bin/env python
"""Summarise the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py` into the HBM traffic per launch of
the dominant kernel, corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes:
  * both counters are reported in KiB;
  * on gfx950 FETCH_SIZE counts 128-byte requests at 64 bytes: wide streaming reads are DOUBLED;
  * WRITE_SIZE is uncalibrated by the guide -> it is calibrated here against a kernel of known write volume in the
    same trace (the bf16 cast kernel `cast_kernel`, which writes exactly 2 bytes per element) when present.
usage: see main()."""
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


def summarise(fe, wr, sub, shape, sel="all"):
    """One kernel (name substring): median FETCH / WRITE per launch.  `sel` = "hi" / "lo" keeps the launches whose FETCH_SIZE
    lies above / below the midpoint of the kernel's range - one template instance serves two GEMM shapes (c_proj, K = 4096,
    and out_proj, K = 1024, both `gemm_nt_pk_kernel<3, 0>`), whose reads differ by the size of A."""
    names = [k for k in fe if sub in k]
    if not names:
        return None
    name = max(names, key=lambda k: len(fe[k]))
    f, w = fe[name], wr.get(name, [])
    if sel in ("hi", "lo") and len(f) > 1:
        mid = (min(f) + max(f)) / 2
        keep = [i for i, v in enumerate(f) if (v > mid) == (sel == "hi")]
        f = [f[i] for i in keep]
        if len(w) == len(fe[name]):          # same launch order in both passes (same command, deterministic schedule)
            w = [w[i] for i in keep]
    f_kib = statistics.median(f); w_kib = statistics.median(w) if w else float("nan")
    return {"shape": shape, "kernel": name[:160], "select": sel, "launches_sampled": len(f), "fetch_size_kib_raw_median": f_kib,
            "write_size_kib_raw_median": w_kib, "fetch_bytes_corrected": f_kib * 1024 * 2, "write_bytes": w_kib * 1024,
            "traffic_bytes_per_launch": f_kib * 1024 * 2 + w_kib * 1024,
            "correction": "FETCH_SIZE[KiB]*1024*2 (gfx950 half-count of wide reads) + WRITE_SIZE[KiB]*1024"}


def main():
    """traffic_summary.py <dir_fetch> <dir_write> <kernel-substring> <out.json> [M,N,K]          (one kernel, round-2 form)
       traffic_summary.py <dir_fetch> <dir_write> <out.json> "<substr>|M,N,K|all/hi/lo[|epi,act]" [...]       (several; first = headline)"""
    dfetch, dwrite = sys.argv[1:3]
    fe, wr = per_kernel(dfetch, "FETCH_SIZE"), per_kernel(dwrite, "WRITE_SIZE")
    if sys.argv[3].endswith(".json"):
        outp = sys.argv[3]
        specs = [a.split("|") for a in sys.argv[4:]]
    else:
        outp = sys.argv[4]
        specs = [[sys.argv[3], sys.argv[5] if len(sys.argv) > 5 else "", "all"]]
    entries = []
    for sub, shp, *rest in specs:          # "<substr>|M,N,K|sel|epi,act"
        e = summarise(fe, wr, sub, [int(v) for v in shp.split(",")] if shp else None, rest[0] if rest else "all")
        if e is None:
            print("no kernel matching", sub, "among", len(fe))
            continue
        if len(rest) > 1:
            e["epi"], e["act"] = (int(v) for v in rest[1].split(","))
        entries.append(e)
    if not entries:
        sys.exit(1)
    res = dict(entries[0])
    res["shapes"] = entries
    res["all_kernels"] = {k[:100]: {"n": len(v), "fetch_kib_median": statistics.median(v),
                                     "write_kib_median": statistics.median(wr[k]) if k in wr else None}
                          for k, v in sorted(fe.items(), key=lambda kv: -sum(kv[1]))[:16]}
    json.dump(res, open(outp, "w"), indent=1)
    for e in entries:
        print(json.dumps(e))


if __name__ == "__main__":
    main()
