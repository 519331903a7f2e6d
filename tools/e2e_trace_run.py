#!/usr/bin/env python3
"""One host-pointer call under a tracer (rocprofv3 --kernel-trace --memory-copy-trace): usage python3 tools/e2e_trace_run.py [pinned|pageable] [ncol]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rrtmg_lw_amd import api
from rrtmg_lw_amd.synth import make_gcm_inputs
mode = sys.argv[1] if len(sys.argv) > 1 else "pinned"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
api.rrtmg_lw_ini(1004.0, kdata=api.STANDIN_KDATA, device=0)
d = make_gcm_inputs(n, 72, "cloudy", col0=0)
out = api._out_arrays(n, 72, d["idrv"])
for v in out.values():
    v[...] = 0.0
arrs = [v for v in list(d.values()) + list(out.values()) if isinstance(v, np.ndarray) and v.flags.f_contiguous and v.dtype == np.float64]
if mode == "pinned":
    for v in arrs:
        api.host_register(v)
api.rrtmg_lw_from_dict(d, out=out)
api.rrtmg_lw_from_dict(d, out=out)
