#!/usr/bin/env python
"""Straight-line VGPR liveness of one kernel in a hipcc -S listing: where the register allocation's peak comes from.

    python tools/isa_liveness.py /tmp/file.s KERNEL_SUBSTRING [window]

Treats the kernel body as one basic block (backward scan: a register is live from its last definition before a use to that
use) — exact inside unrolled loop bodies, approximate across branches.  Prints the live count every `window` instructions with
the most frequent opcodes there, and the peak."""
import collections
import re
import sys


def regs(tok):
    out = []
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(3) is not None:
            out.append(int(m.group(3)))
        else:
            out.extend(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def main():
    text = open(sys.argv[1]).read().splitlines()
    pat = sys.argv[2]
    window = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    starts = [i for i, l in enumerate(text) if re.match(r"^_Z\S*:", l)]
    for k, a in enumerate(starts):
        if pat in text[a]:
            body = text[a:starts[k + 1] if k + 1 < len(starts) else len(text)]
            break
    else:
        raise SystemExit("kernel not found")
    ins = []
    for l in body:
        l = l.split(";")[0].strip()
        if not l or l.startswith(".") or l.endswith(":"):
            continue
        op, _, rest = l.partition(" ")
        ops = [o.strip() for o in rest.split(",")]
        has_dst = (op.startswith("v_") and not op.startswith("v_cmp") and not op.startswith("v_nop")) or \
            re.match(r"(ds_read|ds_bpermute|ds_swizzle|global_load|buffer_load|scratch_load|flat_load|global_atomic.*rtn)", op) is not None
        if "lds" in op and op.startswith(("global_load_lds", "buffer_load")) and "lds" in rest:
            has_dst = False
        dst = regs(ops[0]) if has_dst and ops else []
        src = [r for o in (ops[1:] if has_dst else ops) for r in regs(o)]
        if op.startswith(("v_fmac", "v_mac", "v_mfma")) or "dpp" in l:      # read-modify-write destinations
            src += dst
        ins.append((op, dst, src))
    live, counts = set(), [0] * len(ins)
    for i in range(len(ins) - 1, -1, -1):
        op, dst, src = ins[i]
        live -= set(dst)
        live |= set(src)
        counts[i] = len(live)
    peak = max(counts)
    print(f"{len(ins)} instructions, peak {peak} live VGPRs at instruction {counts.index(peak)}")
    for i in range(0, len(ins), window):
        c = collections.Counter(op for op, _, _ in ins[i:i + window]).most_common(4)
        print(f"{i:6d}  live max {max(counts[i:i + window]):4d}  {c}")


if __name__ == "__main__":
    main()
