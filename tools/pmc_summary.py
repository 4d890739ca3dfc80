#!/usr/bin/env python
"""Reduce rocprofv3 --pmc counter CSVs of tools/model_kernel_bench.py to one line per kernel (average per launch).

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA --kernel-trace --output-format csv -d out -- \\
        python tools/model_kernel_bench.py --launches 3
    python tools/pmc_summary.py out/*/*counter_collection.csv [more.csv ...]

mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs  over  SQ_BUSY_CYCLES / 32 shader engines: the fraction of the kernel's
duration the matrix pipes were executing."""
import collections
import csv
import re
import sys

KERNELS = re.compile(r"(wino3x3_kernel<[^>]*>|conv1x1_mfm_\w+_kernel<[^>]*>|conv5_mfm_pool2_\w+_kernel|stft_\w+_kernel(?:<[^>]*>)?|"
                     r"lfcc_project\w*_kernel(?:<[^>]*>)?|lstm_\w+_kernel<[^>]*>)")


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in sys.argv[1:]:
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                m = KERNELS.search(r["Kernel_Name"])
                if m:
                    agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(agg):
        row = {c: sum(v) / len(v) for c, v in agg[k].items()}
        line = f"{k:44s}"
        if row.get("SQ_BUSY_CYCLES", 0) > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in row:
            line += f" mfma_busy={row['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / (row['SQ_BUSY_CYCLES'] / 32):5.2f}"
        for c in sorted(row):
            line += f"  {c}={row[c]:.4g}"
        print(line)


if __name__ == "__main__":
    main()
