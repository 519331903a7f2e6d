#!/usr/bin/env python3
"""Turn a tools/pmc_quick.sh summary (gpurun_out/<dir>/summary.md) into an entry of profiles/pmc_compute.json, the per-column
instruction and LDS-byte counts that bench.py's `compute` object multiplies with live times.

usage: tools/pmc_to_compute.py <summary.md> <key e.g. cloudy_L72> <columns of the profiled call> [--out profiles/pmc_compute.json]

valu_wave_instr_per_column = sum over kernels of waves x SQ_INSTS_VALU per wave / columns (a wave-instruction covers 64 lanes);
lds_bytes_per_column       = sum of waves x SQ_INSTS_LDS per wave x bytes per LDS wave-instruction / columns, with 1024 B for the
                             kernels that read 16 B per lane (k_layer's table rows) and 512 B for the 8-byte gathers of the sweeps.
"""
import argparse
import json
import os


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("summary")
    ap.add_argument("key")
    ap.add_argument("columns", type=int)
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_compute.json"))
    a = ap.parse_args()
    rows, hdr = [], None
    for line in open(a.summary):
        if not line.startswith("|") or set(line.strip()) <= set("|-"):
            continue
        cells = [c.strip() for c in line.strip().strip("|").split("|")]
        if hdr is None:
            hdr = cells
            continue
        rows.append(dict(zip(hdr, cells)))
    valu = lds_b = salu = 0.0
    per_kernel = {}
    for r in rows:
        w = float(r["waves"])
        v, s_, l = float(r["valu_per_wave"]), float(r["salu_per_wave"]), float(r.get("lds_per_wave", 0) or 0)
        width = 1024.0 if r["kernel"].startswith("k_layer") else 512.0
        valu += w * v
        salu += w * s_
        lds_b += w * l * width
        per_kernel[r["kernel"]] = dict(waves=w, valu_per_wave=v, salu_per_wave=s_, lds_per_wave=l, avg_us=float(r["avg_us"]),
                                       wait_any=float(r["wait_any"]), wait_inst_any=float(r["wait_inst_any"]),
                                       active_inst_valu=float(r["active_inst_valu"]))
    ent = dict(valu_wave_instr_per_column=round(valu / a.columns, 2), salu_wave_instr_per_column=round(salu / a.columns, 2),
               lds_bytes_per_column=round(lds_b / a.columns, 1), profiled_columns=a.columns,
               source=f"rocprofv3 --pmc passes of tools/pmc_quick.sh ({os.path.basename(os.path.dirname(os.path.abspath(a.summary)))})",
               kernels=per_kernel)
    data = json.load(open(a.out)) if os.path.exists(a.out) else {}
    data[a.key] = ent
    json.dump(data, open(a.out, "w"), indent=1, sort_keys=True)
    print(a.key, {k: v for k, v in ent.items() if k != "kernels"})


if __name__ == "__main__":
    main()
