#!/usr/bin/env python3
"""The column order of k_colsort (rrtmg_lw_hip_set_column_sort) against the columns as they lie, device-resident calls, the settings taken in
turn several times so that drift of the box cancels.   usage: python tools/ab_colsort.py [--configs cloudy,cloudy_scatter,...] [--mins 0,20,40,60]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="cloudy,cloudy_towers,cloudy_scatter,cloudy_deep")
ap.add_argument("--mins", default="0,20,40,60")
ap.add_argument("--ncol", type=int, default=1000000)
ap.add_argument("--nlay", type=int, default=72)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--rounds", type=int, default=4)
args = ap.parse_args()
import torch
from rrtmg_lw_amd import api
from rrtmg_lw_amd.synth import make_gcm_inputs
from rrtmg_lw_amd.shard import output_rows, output_views
dev = torch.device("cuda", 0)
api.rrtmg_lw_ini(1004.0, kdata=api.REAL_KDATA if os.path.exists(api.REAL_KDATA) else api.STANDIN_KDATA, device=0)
stream = torch.cuda.current_stream().cuda_stream
settings = [("off", 0, -1)] + [(f"min {m}", 1, int(m)) for m in args.mins.split(",")]
print("| config | " + " | ".join(s[0] for s in settings) + " | bit-identical |")
print("|---|" + "---|" * (len(settings) + 1))
for cfg in args.configs.split(","):
    slab = 131072
    parts = [make_gcm_inputs(min(slab, args.ncol - s), args.nlay, cfg, col0=s, backend="torch", device=dev) for s in range(0, args.ncol, slab)]
    d = dict(parts[0])
    d["ncol"] = args.ncol
    for k, v in parts[0].items():           # (column-fastest storage, as bench.py builds it)
        if torch.is_tensor(v) and len(parts) > 1:
            cat = torch.cat([p[k] for p in parts], dim=1 if k == "taucld" else 0)
            nd = cat.dim()
            d[k] = cat.permute(*reversed(range(nd))).contiguous().permute(*reversed(range(nd))) if nd > 1 else cat.contiguous()
    del parts
    idrv = d["idrv"]
    bufs = [torch.zeros((output_rows(args.nlay, idrv), args.ncol), dtype=torch.float64, device=dev) for _ in settings]
    best = [1e9] * len(settings)
    for rnd in range(args.rounds):
        for i, (name, on, mn) in enumerate(settings):
            api.set_column_sort(on, mn)
            o = output_views(bufs[i], args.nlay, idrv)
            api.rrtmg_lw_device(d, o, stream=stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                api.rrtmg_lw_device(d, o, stream=stream)
            torch.cuda.synchronize()
            best[i] = min(best[i], 1e3 * (time.perf_counter() - t0) / args.reps)
    api.check(stream)
    same = all(bool(torch.equal(bufs[0].view(torch.int64), b.view(torch.int64))) for b in bufs[1:])
    print(f"| {cfg} | " + " | ".join(f"{b:.2f}" for b in best) + f" | {same} |", flush=True)
    del d, bufs
    torch.cuda.empty_cache()
api.finalize()
