import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rrtmg_lw_amd import api
from rrtmg_lw_amd.synth import make_gcm_inputs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
api.rrtmg_lw_ini(1004.0, kdata=api.STANDIN_KDATA, device=0)
d = make_gcm_inputs(n, 72, "cloudy", col0=0)
out = api._out_arrays(n, 72, d["idrv"])
for v in out.values(): v[...] = 0.0
arrs = [v for v in list(d.values()) + list(out.values()) if isinstance(v, np.ndarray) and v.flags.f_contiguous and v.dtype == np.float64]
for v in arrs: api.host_register(v)
static = [d[k] for k in ("co2vmr", "ch4vmr", "n2ovmr", "o2vmr", "cfc11vmr", "cfc12vmr", "cfc22vmr", "ccl4vmr", "tauaer", "emis")]
def t1():
    t = time.perf_counter(); api.rrtmg_lw_from_dict(d, out=out); return 1e3 * (time.perf_counter() - t)
api.rrtmg_lw_from_dict(d, out=out)
res = {"pinned": [], "static": []}
for r in range(6):
    res["pinned"].append(t1())
    for v in static: api.host_static(v)
    t1()
    res["static"].append(t1())
    for v in static: api.host_changed(v, keep=False)
for k, v in res.items():
    print(k, "min %.2f med %.2f" % (min(v), sorted(v)[len(v)//2]), ["%.1f" % x for x in v])
