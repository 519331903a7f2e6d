#!/usr/bin/env python3
"""Print a one-line summary of bench.py JSON lines (tuning helper)."""
import json
import sys

for path in sys.argv[1:]:
    for l in open(path):
        l = l.strip()
        if l.startswith("#"):
            print("   ", l)
        elif l.startswith("{"):
            r = json.loads(l)
            top = ", ".join(f"{t['kernel']}={t['ms_per_step']}" for t in r["path"]["top"])
            fam = r["path"].get("families", {})
            print(f"{path}: {r['value']:.0f} col/s  {r['ms_per_step']} ms/step  kernels {r['path']['kernels_ms_per_step']} ms | {fam} | {top}")
        elif l and "amdgpu.ids" not in l:
            print("   ", l[:200])
