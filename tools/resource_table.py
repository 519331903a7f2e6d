#!/usr/bin/env python3
"""hipcc -Rpass-analysis=kernel-resource-usage output -> one line per kernel (name, VGPRs, AGPRs, scratch, waves/SIMD, spills, LDS).
usage: hipcc ... -Rpass-analysis=kernel-resource-usage driver.hip -o x.so 2> usage.txt ; python tools/resource_table.py usage.txt"""
import re, subprocess, sys
rows, cur = [], None
for line in open(sys.argv[1]):
    m = re.search(r"remark: (?:\s*)Function Name: (\S+)", line)
    if m:
        cur = dict(name=m.group(1)); rows.append(cur); continue
    m = re.search(r"remark:\s+(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).split(" [")[0]] = int(m.group(2))
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
print("| kernel | VGPRs | AGPRs | scratch B/lane | waves/SIMD | VGPRs spilled | LDS B/block (static) |")
print("|---|---|---|---|---|---|---|")
for r, n in sorted(zip(rows, names), key=lambda x: x[1]):
    n = re.sub(r"\(.*", "", n).replace("void rrlw::", "").replace("rrlw::", "")
    print(f"| `{n}` | {r.get('VGPRs')} | {r.get('AGPRs')} | {r.get('ScratchSize')} | {r.get('Occupancy')} | {r.get('VGPRs Spill')} | {r.get('LDS Size')} |")
