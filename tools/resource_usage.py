#!/usr/bin/env python
"""Per-kernel register / spill / occupancy table of one .hip source as hipcc reports it for gfx950.

    python tools/resource_usage.py audio_deepfake_adversarial_attacks_amd/csrc/lcnn_wino.hip [name-substring]
"""
import re
import subprocess
import sys

src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ("/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -Iinclude "
       f"-c {src} -o /tmp/_ru.o -Rpass-analysis=kernel-resource-usage")
out = subprocess.run(cmd.split(), capture_output=True, text=True).stderr
cur, rows = None, {}
for line in out.splitlines():
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", cur.replace("(anonymous namespace)::", "").replace("void ", ""))
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    if pat in k:
        print(f"{k:58s} VGPR {v.get('VGPRs', 0):4d} AGPR {v.get('AGPRs', 0):4d} spillV {v.get('VGPRs Spill', 0):4d} "
              f"spillS {v.get('SGPRs Spill', 0):3d} SGPR {v.get('TotalSGPRs', 0):4d} occ {v.get('Occupancy', 0)} "
              f"scratch {v.get('ScratchSize', 0)}")
if " error" in out:
    print(out[-3000:])
