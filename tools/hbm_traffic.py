#!/usr/bin/env python
"""HBM traffic per launch of the kernels bench.py prices (`roofline.traffic`), from rocprofv3 PMC passes.

Counters cannot be read inside an un-profiled bench run, so they are collected here, in separate passes (FETCH_SIZE and
WRITE_SIZE do not fit one pass; /opt/skills/guides/MI355X_MICROARCH.md, sections HBM and rocprofv3 PMC slots), over
tools/kernel_microbench.py in its "cold" regime (5 buffer sets, 0.5-0.8 GB: beyond the 256 MiB Infinity Cache), and reduced
to profiles/hbm_traffic.json, which bench.py reads and labels with this provenance:

    cd /tmp && export TMPDIR=/tmp
    for c in FETCH_SIZE WRITE_SIZE; do for k in pgd_linf_step:128 pgd_l2_step:128 cw_adam_step:64; do
      rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_${c}_${k%%:*} -- \
        python $REPO/tools/kernel_microbench.py --only ${k%%:*} --batch ${k##*:} --launches 10 --sets 5; done; done
    python tools/hbm_traffic.py profiles/hbm_traffic.json /tmp/pmc_*/*/*counter_collection.csv

Units / corrections (calibrated in round 1 on kernels that read or write exactly one (128, 64600) array, profiles/README.md):
both counters report KiB; FETCH_SIZE reports half the bytes of a 16 B/lane coalesced streaming read on gfx950 (x 2)."""
import collections
import csv
import json
import re
import sys

T = 64_600
# entry point -> (kernel-name patterns of ONE call, algorithmic bytes per sample, batch size of the pass = bench.py's)
ENTRY_POINTS = {
    "pgd_linf_step": ([r"flat_vec_kernel<3,.*PgdLinfOp"], 16, 128),
    # one call = the single-pass kernel + the repair kernel queued behind it (reads one flag per row unless a row was abandoned)
    "pgd_l2_step": ([r"pgd_l2_fused_kernel", r"pgd_l2_repair_kernel"], 16, 128),   # (three-kernel path: sumsq_partial + pgd_l2_delta + pgd_l2_project)
    "cw_adam_step": ([r"cw_adam_vec_kernel"], 32, 64),
}


def reduce(paths, batch_override=None):
    """{entry point: traffic row} from rocprofv3 counter_collection CSVs (FETCH_SIZE and WRITE_SIZE passes)."""
    # kernel name -> counter -> [(grid size, value)]
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in paths:
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                rows[r["Kernel_Name"]][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    table = {}
    for entry, (patterns, bytes_per_sample, batch) in ENTRY_POINTS.items():
        batch = batch_override or batch
        fetch = write = 0.0
        launches, found = None, True
        for n_pat, pat in enumerate(patterns):
            names = [k for k in rows if re.search(pat, k)]
            if not names or not all(c in rows[names[0]] for c in ("FETCH_SIZE", "WRITE_SIZE")):
                found = n_pat > 0 and launches is not None        # companion kernels are optional (older builds lack them)
                break
            k = names[0]
            f_vals = [v for _, v in rows[k]["FETCH_SIZE"]]
            w_vals = [v for _, v in rows[k]["WRITE_SIZE"]]
            fetch += 2.0 * 1024.0 * sum(f_vals) / len(f_vals)      # KiB, x2 gfx950 coalesced-read correction
            write += 1024.0 * sum(w_vals) / len(w_vals)
            launches = len(f_vals)
        if not found:
            continue
        # sanity: every priced kernel writes whole (batch, T) f32 arrays (1, 1 and 3 of them)
        arrays_written = {"pgd_linf_step": 1, "pgd_l2_step": 1, "cw_adam_step": 3}[entry]
        if abs(write / arrays_written / (batch * T * 4) - 1.0) > 0.02:
            raise SystemExit(f"{entry}: {write:.0f} B written per launch does not look like {arrays_written} x ({batch}, {T}) f32")
        table[entry] = {"batch": batch, "hbm_bytes_per_launch": fetch + write, "fetch_bytes_corrected": fetch,
                        "write_bytes": write, "algorithmic_bytes_per_launch": bytes_per_sample * batch * T,
                        "ratio_to_algorithmic": (fetch + write) / (bytes_per_sample * batch * T), "launches_averaged": launches}
    return table


SOURCE = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), cold regime (5 buffer sets); both counters in KiB, "
          "FETCH_SIZE x 2 (gfx950 16 B/lane coalesced-read correction); reduced by tools/hbm_traffic.py")


def main():
    out, paths = sys.argv[1], sys.argv[2:]
    table = reduce(paths)
    doc = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/kernel_microbench.py, cold regime "
                     "(5 buffer sets); both counters in KiB, FETCH_SIZE x 2 (gfx950 16 B/lane coalesced-read correction); "
                     "reduced by tools/hbm_traffic.py",
           "entry_points": table}
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    for k, v in table.items():
        print(f"{k:16s} B={v['batch']:4d}  {v['hbm_bytes_per_launch'] / 1e6:8.1f} MB/launch  = {v['ratio_to_algorithmic']:.3f} x algorithmic")


if __name__ == "__main__":
    main()
