#!/usr/bin/env python3
"""Fused McICA entry with the Mersenne-Twister generator (irng = 1): first call with a seed (jump-ahead of the chunk states) and calls
with the states cached, beside kissvec.  usage: python tools/mt_timing.py [ncol ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from rrtmg_lw_amd import api
    from rrtmg_lw_amd.shard import output_rows, output_views
    from rrtmg_lw_amd.synth import make_gcm_inputs
    dev = torch.device("cuda", 0)
    api.rrtmg_lw_ini(1004.0, device=0)
    stream = torch.cuda.current_stream().cuda_stream
    nlay = 72
    for ncol in [int(a) for a in sys.argv[1:]] or [65536, 1000000]:
        d = make_gcm_inputs(ncol, nlay, "cloudy", backend="torch", device=dev)
        out = output_views(torch.zeros((output_rows(nlay, 0), ncol), dtype=torch.float64, device=dev), nlay, 0)
        row = dict(ncol=ncol)
        for irng, seed in ((0, 140), (1, 140), (1, 141)):
            t0 = time.perf_counter()
            api.rrtmg_lw_mcica_subcol_device(d, out, seed, irng, icld=2, stream=stream)
            api.check(stream)
            first = time.perf_counter() - t0
            t0 = time.perf_counter()
            for _ in range(3):
                api.rrtmg_lw_mcica_subcol_device(d, out, seed, irng, icld=2, stream=stream)
            api.check(stream)
            row["irng%d_seed%d" % (irng, seed)] = dict(first_call_ms=round(1e3 * first, 2), cached_ms=round(1e3 * (time.perf_counter() - t0) / 3, 2))
        print(json.dumps(row), flush=True)
    api.finalize()


if __name__ == "__main__":
    main()
