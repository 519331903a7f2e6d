#!/usr/bin/env python3
"""Device-resident cloudy calls that do not fill the chip: three sweep launches per band group against one (rrtmg_lw_hip_set_one_sweep_max).
usage: python tools/small_sweep.py [--ncols 256,1024,4096,8192,16384,32768] [--config cloudy] > table.md"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--ncols", default="256,1024,4096,8192,16384,32768")
ap.add_argument("--config", default="cloudy")
ap.add_argument("--nlay", type=int, default=72)
ap.add_argument("--reps", type=int, default=200)
args = ap.parse_args()
import torch
from rrtmg_lw_amd import api
from rrtmg_lw_amd.synth import make_gcm_inputs
from rrtmg_lw_amd.shard import output_rows, output_views
dev = torch.device("cuda", 0)
api.rrtmg_lw_ini(1004.0, kdata=api.REAL_KDATA if os.path.exists(api.REAL_KDATA) else api.STANDIN_KDATA, device=0)
stream = torch.cuda.current_stream().cuda_stream
print("| columns per call | three launches, ms | one launch, ms | M columns/s (three / one) | bit-identical |")
print("|---|---|---|---|---|")
for ncol in [int(x) for x in args.ncols.split(",")]:
    d = make_gcm_inputs(ncol, args.nlay, args.config, col0=11, backend="torch", device=dev)
    idrv = d["idrv"]
    res, ms = [], []
    for mx in (0, 1 << 30):
        api.set_one_sweep_max(mx)
        buf = torch.zeros((output_rows(args.nlay, idrv), ncol), dtype=torch.float64, device=dev)
        o = output_views(buf, args.nlay, idrv)
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
    same = bool(torch.equal(res[0].view(torch.int64), res[1].view(torch.int64)))
    print(f"| {ncol} | {ms[0]:.3f} | {ms[1]:.3f} | {ncol / ms[0] / 1e3:.2f} / {ncol / ms[1] / 1e3:.2f} | {same} |", flush=True)
api.finalize()
