#!/usr/bin/env python3
"""End-to-end rate of the McICA host-pointer entries on the benchmark columns (pageable arrays): the reference's two-call sequence -
mcica_subcol_lw, then rrtmg_lw with the (140, ncol, nlay) sub-column arrays - against the fused generator + solver entry, which never
materialises them.  usage: python tools/e2e_timing_mcica.py [ncol] [nlay]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rrtmg_lw_amd import api
from rrtmg_lw_amd.synth import make_gcm_inputs

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
nlay = int(sys.argv[2]) if len(sys.argv) > 2 else 72
api.rrtmg_lw_ini(1004.0, kdata=api.STANDIN_KDATA, device=0)
d = make_gcm_inputs(n, nlay, "cloudy", col0=0)
sub = api.mcica_subcol_lw(64, nlay, 2, 140, 0, *[np.asfortranarray(d[k][:64]) for k in ("play", "cldfr", "cicewp", "cliqwp", "reice", "reliq")],
                          np.asfortranarray(d["taucld"][:, :64, :]))
ng = sub["cldfmcl"].shape[0]
sub = dict(cldfmcl=np.zeros((ng, n, nlay), order="F"), ciwpmcl=np.zeros((ng, n, nlay), order="F"), clwpmcl=np.zeros((ng, n, nlay), order="F"),
           taucmcl=np.zeros((ng, n, nlay), order="F"), reicmcl=np.zeros((n, nlay), order="F"), relqmcl=np.zeros((n, nlay), order="F"))
for v in sub.values():
    v[...] = 0.0                                   # pages touched: the host model's sub-column arrays persist
gen = lambda: api.mcica_subcol_lw(n, nlay, 2, 140, 0, d["play"], d["cldfr"], d["cicewp"], d["cliqwp"], d["reice"], d["reliq"], d["taucld"], out=sub)


def best(f, reps=3):
    f()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return min(ts)


res = dict(columns=n, nlay=nlay)
t_gen = best(gen)
dd = dict(d); dd.update({k: sub[k] for k in ("cldfmcl", "ciwpmcl", "clwpmcl", "taucmcl", "reicmcl", "relqmcl")})
t_arr = best(lambda: api.rrtmg_lw_mcica_from_dict(dd, icld=2))
t_fused = best(lambda: api.rrtmg_lw_mcica_subcol_from_dict(d, 140, 0, icld=2))
res["generator_to_host_arrays"] = dict(ms=round(1e3 * t_gen, 1), columns_per_s=round(n / t_gen))
res["rrtmg_lw_with_subcolumn_arrays"] = dict(ms=round(1e3 * t_arr, 1), columns_per_s=round(n / t_arr))
res["two_call_sequence"] = dict(ms=round(1e3 * (t_gen + t_arr), 1), columns_per_s=round(n / (t_gen + t_arr)))
res["fused_entry"] = dict(ms=round(1e3 * t_fused, 1), columns_per_s=round(n / t_fused))
print(json.dumps(res))
