#!/usr/bin/env python
"""Export the per-kernel summary (rocprofv3 --kernel-trace --stats, rocpd sqlite output) as CSV.
usage: python tools/rocpd_stats.py gpurun_out/prof/x_results.db profiles/r01_kernel_stats.csv"""
import csv
import sqlite3
import sys


def main(db, out):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    extra = {}
    for name, vg, av, lds, wx, gx in c.execute(
            "select name, max(vgpr_count), max(accum_vgpr_count), max(lds_size), max(workgroup_x), max(grid_x) from kernels group by name"):
        extra[name] = (vg, av, lds, wx, gx)
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage", "VGPR", "AGPR", "LDS", "WorkgroupX", "GridX"])
        for r in rows:
            w.writerow([r[0], r[1], round(r[2], 3), round(r[3], 3), round(r[4], 3), *extra.get(r[0], ("",) * 5)])
    print(f"{len(rows)} kernels -> {out}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
