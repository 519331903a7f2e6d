#!/usr/bin/env python3
"""One configuration of the overlapped pipeline under `rocprofv3 --kernel-trace`: a few steps of the benchmark workload with
--cus N (-1 no overlap, 0 overlap, n = k_layer on n CUs) and --batch B.   Companion: tools/trace_overlap.py <kernel_trace.csv>."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="cloudy")
ap.add_argument("--ncol", type=int, default=1_000_000)
ap.add_argument("--nlay", type=int, default=72)
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--batch", type=int, default=262144)
ap.add_argument("--cus", type=int, default=128)
args = ap.parse_args()
import torch
from rrtmg_lw_amd import api
from rrtmg_lw_amd.synth import make_gcm_inputs
from rrtmg_lw_amd.shard import ShardedStep
dev = torch.device("cuda", 0)
api.rrtmg_lw_ini(1004.0, kdata=api.REAL_KDATA if os.path.exists(api.REAL_KDATA) else api.STANDIN_KDATA, device=0)
ncol, nlay = args.ncol, args.nlay
slab = 131072
parts = [make_gcm_inputs(min(slab, ncol - s), nlay, args.config, col0=s, backend="torch", device=dev) for s in range(0, ncol, slab)]
d = dict(parts[0]); d["ncol"] = ncol
for k, v in parts[0].items():
    if torch.is_tensor(v) and len(parts) > 1:
        cat = torch.cat([p[k] for p in parts], dim=1 if k == "taucld" else 0)
        nd = cat.dim()
        d[k] = cat.permute(*reversed(range(nd))).contiguous().permute(*reversed(range(nd))) if nd > 1 else cat.contiguous()
del parts
sh = ShardedStep(nlay, d["idrv"], ncol, 1, device=dev, gather=False)
stream = torch.cuda.current_stream().cuda_stream
api.set_batch(args.batch)
if args.cus < 0: api.set_overlap(False)
elif args.cus == 0: api.set_overlap(True)
else: api.set_cu_partition(args.cus)
solve = lambda o: api.rrtmg_lw_device(d, o, stream=stream)
sh.step(solve); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps): sh.step(solve)
torch.cuda.synchronize()
print(f"cus {args.cus} batch {args.batch}: {1e3 * (time.perf_counter() - t0) / args.steps:.2f} ms per step")
api.finalize()
