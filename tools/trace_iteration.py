#!/usr/bin/env python
"""Print, in launch order, the kernels of ONE attack iteration from a rocprofv3 kernel-trace CSV of `bench.py`: everything
between the last two launches of a marker kernel (the attack's update step, e.g. `pgd_l2_fused_kernel`).

    python tools/trace_iteration.py <kernel_trace.csv> <marker substring>
"""
import csv
import re
import sys


def main():
    path, marker = sys.argv[1], sys.argv[2]
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                         r.get("Grid_Size", ""), r.get("Workgroup_Size", "")))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if marker in r[2]]
    if len(marks) < 2:
        raise SystemExit(f"fewer than two launches of a kernel matching {marker!r}")
    lo, hi = marks[-2], marks[-1]
    total = 0
    prev_end = rows[lo][1]
    for s, e, n, grid, wg in rows[lo + 1:hi + 1]:
        n = re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", n))
        total += e - s
        print(f"{(e - s) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:6.1f}  grid {grid:>9s}/{wg:<5s} {n[:110]}")
        prev_end = e
    print(f"iteration: {hi - lo} launches, {total / 1e6:.3f} ms busy, {(rows[hi][1] - rows[lo][1]) / 1e6:.3f} ms wall")


if __name__ == "__main__":
    main()
