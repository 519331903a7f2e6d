#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (tools/pmc_collect.sh) into per-kernel averages.

usage: tools/pmc_summarize.py <dir with fetch/ write/ sq1/ sq2/> [--json profiles/pmc_traffic.json] [--md profiles/<name>.md]

HBM traffic follows MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are collected in separate passes and
reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE tallies a 128-byte request as 64 bytes for wide (16 B per lane)
streaming reads, so it is doubled; WRITE_SIZE is exact for 16 B per lane stores.  The calibration kernel k_calibrate
(known: reads N bytes and writes N bytes, 16 B per lane) is part of every pass and its measured/known ratio is printed
so that the unit and the x2 correction can be checked in the same session.  Kernels that mix access widths are
"uncalibrated" in the guide's sense: their absolute traffic is indicative, ratios between variants are reliable.
"""
import argparse
import csv
import glob
import json
import os
import re
from collections import defaultdict


def short_name(k):
    k = re.sub(r"^void\s+", "", k)
    k = re.sub(r"\(.*$", "", k)
    k = k.replace("rrlw::", "")
    m = re.match(r"k_layer<(\w+), (\d+), (\d+)>", k)
    if m:
        return "k_layer<%s,%s>" % ({"0": "clear", "1": "cloud", "2": "mcica", "3": "mcmask"}[m.group(2)], m.group(3))
    m = re.match(r"k_cloudmc<(\w+)>", k)
    if m:
        return "k_cloudmc<%s>" % ("mask" if m.group(1) == "true" else "arrays")
    m = re.match(r"(k_colprep|k_cloudscan|k_cloudlay|k_cloud)<\w+>", k)
    if m:
        return m.group(1)
    m = re.match(r"k_sweepc<(\d+), (\d+), (\w+)>", k)      # (quads per thread, phase); the d(flux)/dT instantiation shares the bench's kernel name
    if m:
        return "k_sweepc<%s,%s>" % (m.group(1), m.group(2))
    m = re.match(r"k_sweepz<(\d+)>", k)
    if m:
        return "k_sweepz<%s>" % m.group(1)
    m = re.match(r"k_sweepc<(\d+), (\d+), (\w+), (\d+)>", k)      # (with the waves-per-band parameter)
    if m:
        return "k_sweepc<%s,%s>" % (m.group(1), m.group(2))
    m = re.match(r"k_n1<\w+>", k)
    if m:
        return "k_n1"
    m = re.match(r"k_sweep<(\d+), (\d+), (\w+)>", k)       # the d(flux)/dT instantiation shares the bench's kernel name
    if m:
        return "k_sweep<%s,%s>" % (m.group(1), m.group(2))
    return k.replace(", ", ",")


def load(dirname):
    """-> {kernel: {counter: [values per dispatch]}}, {kernel: [durations ns]}"""
    vals = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(dict)
    for path in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            k = short_name(r["Kernel_Name"])
            vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return vals, dur


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--json")
    ap.add_argument("--md")
    ap.add_argument("--key", default="cloudy_L72", help="configuration key of the entry written to --json (bench.py: <config>_L<nlay>[_mcica<icld>])")
    ap.add_argument("--columns", type=int, default=250000, help="columns of the profiled call (tools/pmc_run.py --ncol)")
    ap.add_argument("--calib-bytes", type=float, default=float(1 << 30))
    ap.add_argument("--skip-first", action="store_true", default=True,
                    help="drop the first half of each kernel's dispatches (the warm-up call)")
    args = ap.parse_args()
    merged = defaultdict(dict)
    launches = {}
    for sub in sorted(os.listdir(args.dir)):
        p = os.path.join(args.dir, sub)
        if not os.path.isdir(p):
            continue
        vals, dur = load(p)
        for k, cs in vals.items():
            for c, v in cs.items():
                if args.skip_first and len(v) >= 2 and k != "k_calibrate":
                    v = v[len(v) // 2:]
                merged[k][c] = sum(v) / len(v)
                launches[k] = len(v)
            d = list(dur[k].values())
            if args.skip_first and len(d) >= 2 and k != "k_calibrate":
                d = d[len(d) // 2:]
            merged[k].setdefault("_dur_us", sum(d) / len(d) / 1e3)
    cal = merged.get("k_calibrate", {})
    lines = []
    if cal:
        f = cal.get("FETCH_SIZE", 0.0) * 1024.0
        w = cal.get("WRITE_SIZE", 0.0) * 1024.0
        lines.append(f"calibration k_calibrate: known {args.calib_bytes:.0f} B read + same written, 16 B/lane; "
                     f"FETCH_SIZE*1024 = {f:.0f} (x{f / args.calib_bytes:.3f} of known; guide: 0.5), "
                     f"WRITE_SIZE*1024 = {w:.0f} (x{w / args.calib_bytes:.3f}; guide: 1.0)")
    traffic = {}
    rows = []
    for k in sorted(merged, key=lambda k: -merged[k].get("_dur_us", 0)):
        if not k.startswith("k_") or k == "k_calibrate":
            continue
        m = merged[k]
        rd = 2.0 * m.get("FETCH_SIZE", 0.0) * 1024.0
        wr = m.get("WRITE_SIZE", 0.0) * 1024.0
        traffic[k] = rd + wr
        waves = m.get("SQ_WAVES", 0.0)
        row = dict(kernel=k, launches_averaged=launches.get(k, 0), avg_us=round(m.get("_dur_us", 0.0), 1), hbm_read_MB=round(rd / 1e6, 2),
                   hbm_write_MB=round(wr / 1e6, 2),
                   hbm_GBps=round((rd + wr) / (m.get("_dur_us", 1.0) * 1e-6) / 1e9, 1) if m.get("_dur_us") else None,
                   waves=waves, valu_per_wave=round(m.get("SQ_INSTS_VALU", 0.0) / waves, 1) if waves else None,
                   salu_per_wave=round(m.get("SQ_INSTS_SALU", 0.0) / waves, 1) if waves else None,
                   vmem_rd_per_wave=round(m.get("SQ_INSTS_VMEM_RD", 0.0) / waves, 1) if waves else None,
                   vmem_wr_per_wave=round(m.get("SQ_INSTS_VMEM_WR", 0.0) / waves, 1) if waves else None)
        wc = m.get("SQ_WAVE_CYCLES", 0.0)
        if wc:
            row.update(wait_any=round(m.get("SQ_WAIT_ANY", 0.0) / wc, 3), wait_inst_any=round(m.get("SQ_WAIT_INST_ANY", 0.0) / wc, 3),
                       active_inst_any=round(m.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 3),
                       active_inst_valu=round(m.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, 3))
            if "SQ_INSTS_LDS" in m:
                row.update(lds_per_wave=round(m["SQ_INSTS_LDS"] / waves, 1) if waves else None,
                           wait_inst_lds=round(m.get("SQ_WAIT_INST_LDS", 0.0) / wc, 3),
                           lds_bank_conflict_frac=round(m.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(m.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0), 3))
        rows.append(row)
    cols = list(rows[0].keys()) if rows else []
    for r in rows:
        for c in r:
            if c not in cols:
                cols.append(c)
    lines.append("| " + " | ".join(cols) + " |")
    lines.append("|" + "---|" * len(cols))
    for r in rows:
        lines.append("| " + " | ".join(str(r.get(c, "")) for c in cols) + " |")
    text = "\n".join(lines)
    print(text)
    if args.md:
        open(args.md, "w").write(text + "\n")
    if args.json:
        # profiles/pmc_traffic.json: one entry per configuration key (cloudy_L72, clear_L72, cloudy_L72_mcica5, aer_idrv_L137 ...): HBM bytes
        # per launch of every kernel (FETCH_SIZE x 2 + WRITE_SIZE, calibrated above) for the profiled call and their sum per column
        data = {}
        if os.path.exists(args.json):
            try:
                data = json.load(open(args.json))
            except Exception:
                data = {}
        if not all(isinstance(v, dict) for v in data.values()):
            data = {}                         # (the round-2 layout: one flat dictionary of the cloudy configuration)
        per_call = {k: launches.get(k, 1) for k in traffic}
        path = sum(traffic[k] * max(per_call[k], 1) for k in traffic)
        import hashlib
        h = hashlib.sha256()                  # (what bench.py compares: the kernel sources these counters were collected for)
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        for f in ("kernels.hip", "driver.hip"):
            h.update(open(os.path.join(root, "rrtmg_lw_amd", "csrc", f), "rb").read())
        data[args.key] = dict(columns=args.columns, kernels={k: round(v, 1) for k, v in traffic.items()}, launches_per_call=per_call,
                              path_bytes_per_call=round(path, 1), bytes_per_column=round(path / args.columns, 1), kernels_sha16=h.hexdigest()[:16])
        json.dump(data, open(args.json, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
