#!/usr/bin/env python3
"""End-to-end rate of the host-pointer entry (what the Fortran shim calls) on the benchmark columns: pageable and pinned arrays,
persistent in both cases (a host model's arrays live for the whole run).  usage: python tools/e2e_timing.py [config] [ncol] [nlay]
Environment read by the library: RRTMG_LW_HOST_THREADS, RRTMG_LW_HOST_BATCH."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rrtmg_lw_amd import api
from rrtmg_lw_amd.synth import make_gcm_inputs

cfg = sys.argv[1] if len(sys.argv) > 1 else "cloudy"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
nlay = int(sys.argv[3]) if len(sys.argv) > 3 else 72
api.rrtmg_lw_ini(1004.0, kdata=api.STANDIN_KDATA, device=0)
d = make_gcm_inputs(n, nlay, cfg, col0=0)
out = api._out_arrays(n, nlay, d["idrv"])
for v in out.values():
    v[...] = 0.0                                   # pages touched: persistent arrays


def timed(reps=3):
    api.rrtmg_lw_from_dict(d, out=out)
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        api.rrtmg_lw_from_dict(d, out=out)
        ts.append(time.perf_counter() - t)
    return min(ts)


res = dict(config=cfg, columns=n, nlay=nlay, host_threads=os.environ.get("RRTMG_LW_HOST_THREADS"), host_batch=os.environ.get("RRTMG_LW_HOST_BATCH"))
t = timed()
res["pageable"] = dict(ms=round(1e3 * t, 2), columns_per_s=round(n / t))
arrs = [v for v in list(d.values()) + list(out.values()) if isinstance(v, np.ndarray) and v.flags.f_contiguous and v.dtype == np.float64]
for v in arrs:
    api.host_register(v)
t = timed()
res["pinned"] = dict(ms=round(1e3 * t, 2), columns_per_s=round(n / t))
static = [d[k] for k in ("co2vmr", "ch4vmr", "n2ovmr", "o2vmr", "cfc11vmr", "cfc12vmr", "cfc22vmr", "ccl4vmr", "tauaer", "emis")]
for v in static:
    api.host_static(v)                            # arrays a host model sets once: rows scanned by the first call only
t = timed()
res["pinned_static"] = dict(ms=round(1e3 * t, 2), columns_per_s=round(n / t))
if os.environ.get("RRTMG_LW_STAGE_TIMING"):
    api.rrtmg_lw_from_dict(d, out=out)
for v in static:
    api.host_changed(v, keep=False)
for v in arrs:
    api.host_unregister(v)
print(json.dumps(res))
