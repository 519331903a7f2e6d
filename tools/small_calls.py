#!/usr/bin/env python3
"""Latency of small calls (a GCM that chunks its columns, reference src/rrtmg_lw_rad.f90:486) and of the Mersenne-Twister generator path:
device-resident calls of ncol = 64 ... 16384 columns (cloudy, rtrnmr) timed with HIP events over 20 back-to-back calls, the same with the
host-pointer entry, and the fused McICA entry with irng = 0 (kissvec, on the device) against irng = 1 (MT19937 stream drawn on the host)."""
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
    api.rrtmg_lw_ini(1004.0, kdata=api.STANDIN_KDATA, device=0)
    stream = torch.cuda.current_stream().cuda_stream
    nlay, rows = 72, []
    for ncol in (64, 256, 1024, 4096, 16384):
        d = make_gcm_inputs(ncol, nlay, "cloudy", backend="torch", device=dev)
        out = output_views(torch.zeros((output_rows(nlay, 0), ncol), dtype=torch.float64, device=dev), nlay, 0)
        for _ in range(3):
            api.rrtmg_lw_device(d, out, stream=stream)
        api.check(stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            api.rrtmg_lw_device(d, out, stream=stream)
        e1.record()
        torch.cuda.synchronize()
        dev_ms = e0.elapsed_time(e1) / 20
        dh = make_gcm_inputs(ncol, nlay, "cloudy")
        api.rrtmg_lw_from_dict(dh)
        t0 = time.perf_counter()
        for _ in range(5):
            api.rrtmg_lw_from_dict(dh)
        host_ms = 1e3 * (time.perf_counter() - t0) / 5
        rows.append(dict(ncol=ncol, device_ms=round(dev_ms, 4), device_columns_per_s=round(ncol / dev_ms * 1e3), host_ms=round(host_ms, 3),
                         host_columns_per_s=round(ncol / host_ms * 1e3)))
        print(rows[-1], flush=True)
    # the same small chunks aggregated (rrtmg_lw_hip_queue_*): 1024 chunks of 64 columns in one pass
    import numpy as np
    chunks = []
    for c in range(1024):
        chunks.append(make_gcm_inputs(64, nlay, "cloudy", col0=64 * c))
    queue = {}
    for rep in range(2):
        t0 = time.perf_counter()
        q = api.ChunkQueue(nlay, 2, 0, 2, 3, 1)
        outs = [q.add(c) for c in chunks]
        t1 = time.perf_counter()
        q.flush()
        t2 = time.perf_counter()
        queue = dict(chunks=1024, columns_per_chunk=64, add_ms=round(1e3 * (t1 - t0), 1), flush_ms=round(1e3 * (t2 - t1), 1),
                     columns_per_s=round(65536 / (t2 - t0)))
    one = api.rrtmg_lw_from_dict(make_gcm_inputs(64, nlay, "cloudy", col0=64 * 7))
    assert all(np.array_equal(outs[7][k], one[k]) for k in ("uflx", "dflx", "hr"))
    print("queue", queue, flush=True)
    # generator paths: fused McICA entry, 65536 columns, icld = 2
    ncol = 65536
    d = make_gcm_inputs(ncol, nlay, "cloudy", backend="torch", device=dev)
    out = output_views(torch.zeros((output_rows(nlay, 0), ncol), dtype=torch.float64, device=dev), nlay, 0)
    gen = {}
    for irng in (0, 1):
        api.rrtmg_lw_mcica_subcol_device(d, out, 140, irng, icld=2, stream=stream)
        api.check(stream)
        t0 = time.perf_counter()
        for _ in range(3):
            api.rrtmg_lw_mcica_subcol_device(d, out, 140, irng, icld=2, stream=stream)
        api.check(stream)
        gen[irng] = round(1e3 * (time.perf_counter() - t0) / 3, 2)
    print(json.dumps(dict(small_calls=rows, queue=queue, mcica_65536_columns_ms=dict(kissvec_irng0=gen[0], mersenne_twister_irng1=gen[1]))))
    api.finalize()


if __name__ == "__main__":
    main()
