#!/usr/bin/env python3
"""Device-resident calls that do not fill the chip, as plain launches against replayed as one graph (rrtmg_lw_hip_set_graph_max).
usage: python tools/small_graph.py [--cases cloudy:256,cloudy:1024,cloudy:4096,cloudy:16384,clear:10000] > table.md"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--cases", default="cloudy:256,cloudy:1024,cloudy:4096,cloudy:16384,clear:10000")
ap.add_argument("--nlay", type=int, default=72)
ap.add_argument("--reps", type=int, default=200)
ap.add_argument("--side-stream", action="store_true", help="call on a non-default torch stream instead of the null stream")
ap.add_argument("--ab", default="graph", choices=["graph", "split"], help="what the two columns compare: plain launches / one graph, or bands in groups / one band per workgroup (both as graphs)")
args = ap.parse_args()
import torch
from rrtmg_lw_amd import api
from rrtmg_lw_amd.synth import make_gcm_inputs
from rrtmg_lw_amd.shard import output_rows, output_views
dev = torch.device("cuda", 0)
api.rrtmg_lw_ini(1004.0, kdata=api.REAL_KDATA if os.path.exists(api.REAL_KDATA) else api.STANDIN_KDATA, device=0)
side = torch.cuda.Stream(device=dev) if args.side_stream else None
stream = side.cuda_stream if side else torch.cuda.current_stream().cuda_stream
print("| call | plain launches, ms | one graph, ms | M columns/s (plain / graph) | bit-identical | graphs captured / replays |" if args.ab == "graph" else
      "| call | bands in groups, ms | one band per workgroup, ms | M columns/s | bit-identical | graphs captured / replays |")
print("|---|---|---|---|---|---|")
for case in args.cases.split(","):
    cfg, ncol = case.split(":"); ncol = int(ncol)
    d = make_gcm_inputs(ncol, args.nlay, cfg, col0=11, backend="torch", device=dev)
    idrv = d["idrv"]
    res, ms = [], []
    c0, r0 = api.graph_stats()
    for mx in (0, 1 << 20):
        if args.ab == "graph":
            api.set_graph_max(mx)
        else:
            api.set_split_max(mx)
        buf = torch.zeros((output_rows(args.nlay, idrv), ncol), dtype=torch.float64, device=dev)
        o = output_views(buf, args.nlay, idrv)
        torch.cuda.synchronize()
        for _ in range(5):
            api.rrtmg_lw_device(d, o, stream=stream)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(args.reps):
                api.rrtmg_lw_device(d, o, stream=stream)
            torch.cuda.synchronize()
            best = min(best, 1e3 * (time.perf_counter() - t0) / args.reps)
        api.check(stream)
        ms.append(best); res.append(buf.clone())
    c1, r1 = api.graph_stats()
    same = bool(torch.equal(res[0].view(torch.int64), res[1].view(torch.int64)))
    print(f"| {ncol} x {args.nlay} {cfg} | {ms[0]:.3f} | {ms[1]:.3f} | {ncol / ms[0] / 1e3:.2f} / {ncol / ms[1] / 1e3:.2f} | {same} | {c1 - c0} / {r1 - r0} |", flush=True)
api.finalize()
