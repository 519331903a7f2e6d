#!/usr/bin/env python3
"""CU-partitioned overlap of k_layer (batch i+1) with the sweeps (batch i): step time over (columns per batch) x (CUs of k_layer's share),
one process, inputs built once, the variants interleaved over several rounds (median and min reported), outputs compared bit for bit with
the unpartitioned run.   usage: python tools/partition_sweep.py [--config cloudy] [--ncol 1000000] [--mcica 0] > table.md"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cloudy")
    ap.add_argument("--ncol", type=int, default=1_000_000)
    ap.add_argument("--nlay", type=int, default=72)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--batches", default="262144,131072,65536")
    ap.add_argument("--cus", default="-1,0,96,128,160,192", help="-1 = no overlap, 0 = overlap without partition, n = k_layer on n CUs")
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
    d = dict(parts[0])
    d["ncol"] = ncol
    for k, v in parts[0].items():
        if torch.is_tensor(v) and len(parts) > 1:
            cat = torch.cat([p[k] for p in parts], dim=1 if k == "taucld" else 0)
            nd = cat.dim()
            d[k] = cat.permute(*reversed(range(nd))).contiguous().permute(*reversed(range(nd))) if nd > 1 else cat.contiguous()
    del parts
    sh = ShardedStep(nlay, d["idrv"], ncol, 1, device=dev, gather=False)
    stream = torch.cuda.current_stream().cuda_stream

    def solve(o):
        api.rrtmg_lw_device(d, o, stream=stream)

    variants = [(int(b), int(c)) for b in args.batches.split(",") for c in args.cus.split(",")]
    times = {v: [] for v in variants}
    ref = None
    same = {}
    for r in range(args.rounds):
        for v in variants:
            b, c = v
            api.set_batch(b)
            if c < 0:
                api.set_cu_partition(0)
                api.set_overlap(False)
            elif c == 0:
                api.set_cu_partition(0)
                api.set_overlap(True)
            else:
                api.set_cu_partition(c)
            sh.count = 0
            sh.step(solve)                     # warm-up (workspace, streams)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                sh.step(solve)
            torch.cuda.synchronize()
            times[v].append(1e3 * (time.perf_counter() - t0) / args.steps)
            api.check(stream)
            if r == 0:
                cur = sh.outbufs[(sh.count - 1) & 1]
                if ref is None:
                    ref = cur.clone()
                same[v] = bool(torch.equal(cur.view(torch.int64), ref.view(torch.int64)))
            print(f"# round {r} batch {b} cus {c}: {times[v][-1]:.2f} ms", file=sys.stderr, flush=True)
    print(f"| columns per batch | k_layer CUs | ms per step (median of {args.rounds}) | min | bit-identical to the first variant |")
    print("|---|---|---|---|---|")
    for v in variants:
        ts = sorted(times[v])
        lab = "no overlap" if v[1] < 0 else ("overlap, no partition" if v[1] == 0 else f"{v[1]} / {256 - v[1]}")
        print(f"| {v[0]} | {lab} | {ts[len(ts) // 2]:.2f} | {ts[0]:.2f} | {same[v]} |")
    api.finalize()


if __name__ == "__main__":
    main()
