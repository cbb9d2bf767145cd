#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage remarks: one line per kernel.
usage: tools/kres.py file.hip [extra hipcc flags]"""
import re, subprocess, sys
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for ln in out.splitlines():
    m = re.search(r"remark: [^ ]+ +(Function Name|Name): (\S+)", ln) or re.search(r": +(Function Name|Name): (\S+)", ln)
    if m:
        cur = subprocess.run(["c++filt", m.group(2)], capture_output=True, text=True).stdout.strip()
        rows[cur] = {}
        continue
    m = re.search(r": +([A-Za-z ]+(?:\[[^\]]*\])?): +(\d+)", ln)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    name = re.sub(r"\(.*", "", k)
    print(f"{name:60s} vgpr={v.get('VGPRs')} agpr={v.get('AGPRs')} sgpr={v.get('TotalSGPRs')} "
          f"spillV={v.get('VGPRs Spill')} spillS={v.get('SGPRs Spill')} scratch={v.get('ScratchSize [bytes/lane]')} "
          f"occ={v.get('Occupancy [waves/SIMD]')} lds={v.get('LDS Size [bytes/block]')}")
