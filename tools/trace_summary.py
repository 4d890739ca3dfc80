#!/usr/bin/env python
"""Reduce a rocprofv3 kernel-trace CSV of `bench.py` to a per-step table (the raw trace is ~45 MB).

The last timed step is delimited by the last two launches of `minmax_partial_kernel` (first kernel of every step);
kernels are grouped by (shortened) name and reported per PGD iteration.

    python tools/trace_summary.py <kernel_trace.csv> <out.json> [pgd_iterations_per_step=40]
"""
import collections
import csv
import json
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:150]


def main():
    path, out = sys.argv[1], sys.argv[2]
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    starts = [s for s, _, n in rows if "minmax_partial_kernel" in n]
    if len(starts) < 2:
        raise SystemExit("need at least two steps in the trace")
    lo, hi = starts[-2], starts[-1]  # the step before the last one (complete by construction)
    step = [(s, e, n) for s, e, n in rows if lo <= s < hi]
    busy = sum(e - s for s, e, _ in step)
    agg = collections.defaultdict(lambda: [0, 0])
    for s, e, n in step:
        a = agg[short(n)]
        a[0] += 1
        a[1] += e - s
    table = sorted(({"kernel": k, "calls": c, "total_us": t / 1e3, "avg_us": t / c / 1e3, "pct_of_busy": 100.0 * t / busy,
                     "us_per_pgd_iteration": t / 1e3 / iters} for k, (c, t) in agg.items()),
                   key=lambda r: -r["total_us"])
    summary = {"step_wall_ms": (hi - lo) / 1e6, "gpu_busy_ms": busy / 1e6, "launches": len(step),
               "pgd_iterations": iters, "kernels": table}
    with open(out, "w") as f:
        json.dump(summary, f, indent=1)
    print(f"step wall {summary['step_wall_ms']:.1f} ms, GPU busy {summary['gpu_busy_ms']:.1f} ms, {len(step)} launches")
    for r in table[:40]:
        print(f"{r['total_us'] / 1e3:8.2f} ms {r['pct_of_busy']:5.1f}% calls={r['calls']:5d} avg={r['avg_us']:9.1f} us  {r['kernel'][:100]}")


if __name__ == "__main__":
    main()
