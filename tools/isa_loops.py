#!/usr/bin/env python
"""Instruction mix of every loop of the kernels in a hipcc -S listing whose mangled name contains a substring.

    hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only file.hip -o /tmp/file.s
    python tools/isa_loops.py /tmp/file.s wide_kernelILi1ELb0ELi0ELi4
"""
import re
import sys

text = open(sys.argv[1]).read().splitlines()
pat = sys.argv[2]
starts = [i for i, l in enumerate(text) if re.match(r"^_Z\S*:", l)]
for k, a in enumerate(starts):
    name = text[a].split(":")[0]
    if pat not in name:
        continue
    b = starts[k + 1] if k + 1 < len(starts) else len(text)
    lines = text[a:b]
    print(name, len(lines), "lines; scratch ops total", sum("scratch_" in x for x in lines))
    labels = {m.group(1): i for i, l in enumerate(lines) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    for i, l in enumerate(lines):
        m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            body = lines[labels[m.group(1)]:i]
            def cnt(rx):
                return sum(1 for x in body if re.search(rx, x))
            print("  loop %s (%d lines): mfma %d valu %d accvgpr %d vmem %d ds %d scratch %d salu %d waitcnt %d nop %d" % (
                m.group(1), len(body), cnt(r"v_mfma"), cnt(r"^\s+v_(?!mfma|accvgpr)"), cnt(r"v_accvgpr"),
                cnt(r"buffer_load|global_load"), cnt(r"^\s+ds_"), cnt(r"scratch_"), cnt(r"^\s+s_(?!waitcnt|nop|barrier)"),
                cnt(r"s_waitcnt"), cnt(r"s_nop")))
