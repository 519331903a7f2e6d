#!/usr/bin/env python3
"""Reads a rocprofv3 kernel-trace CSV: per kernel family the launches, mean duration, and how much of the time k_layer runs it shares with a
sweep kernel (the overlap the CU partition is for).   usage: python tools/trace_overlap.py kernel_trace.csv [skip_first_n_layer_launches]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    nm = r.get("Kernel_Name") or r.get("Name")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    m = re.search(r"(k_\w+)", nm)
    fam = m.group(1) if m else nm[:30]
    ev.append((s, e, fam, r.get("Queue_Id", "?")))
ev.sort()
t00 = ev[0][0]
fams = {}
for s, e, f, q in ev:
    fams.setdefault(f, []).append((s, e, q))
print("| kernel | launches | mean ms | total ms | queues |")
print("|---|---|---|---|---|")
for f, v in sorted(fams.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
    print(f"| {f} | {len(v)} | {sum(e - s for s, e, _ in v) / len(v) / 1e6:.3f} | {sum(e - s for s, e, _ in v) / 1e6:.2f} | {len(set(q for _, _, q in v))} |")
def union(iv):
    iv = sorted(iv); out = []
    for s, e in iv:
        if out and s <= out[-1][1]: out[-1][1] = max(out[-1][1], e)
        else: out.append([s, e])
    return out
def inter(a, b):
    i = j = 0; tot = 0
    while i < len(a) and j < len(b):
        lo, hi = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if hi > lo: tot += hi - lo
        if a[i][1] < b[j][1]: i += 1
        else: j += 1
    return tot
L = union([(s, e) for s, e, f, q in ev if f == "k_layer"])
S = union([(s, e) for s, e, f, q in ev if f.startswith("k_sweep") or f == "k_flux"])
tl, ts = sum(e - s for s, e in L), sum(e - s for s, e in S)
print(f"\nk_layer busy {tl / 1e6:.2f} ms, sweeps + k_flux busy {ts / 1e6:.2f} ms, both at once {inter(L, S) / 1e6:.2f} ms, span {(ev[-1][1] - t00) / 1e6:.2f} ms")
if len(sys.argv) > 2:
    print("\ntimeline of the last step (ms from its first kernel):")
    n = int(sys.argv[2])
    last = ev[-n:]
    t0 = last[0][0]
    for s, e, f, q in last:
        print(f"  {f:14s} q{q:>3s} {(s - t0) / 1e6:8.3f} -> {(e - t0) / 1e6:8.3f}  ({(e - s) / 1e6:.3f})")
