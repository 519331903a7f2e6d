#!/usr/bin/env python3
"""Soak of the host-pointer entry: random column counts, random subsets of the arrays page-locked, back-to-back calls - every result must
equal, bit for bit, the same columns of one reference call (columns are independent and results do not depend on the batching).
usage: python tools/soak_host_entry.py [seconds] [max columns] [mcica|arrays]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rrtmg_lw_amd import api
from rrtmg_lw_amd.synth import make_gcm_inputs

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
nmax = int(sys.argv[2]) if len(sys.argv) > 2 else 70000
mcica = len(sys.argv) > 3 and sys.argv[3] == "mcica"       # the fused generator + solver entry (kissvec: a column's sub-columns are its own)
arrays = len(sys.argv) > 3 and sys.argv[3] == "arrays"     # mcica_subcol_lw into host arrays, then rrtmg_lw with the (140, ncol, nlay) sub-column arrays
nlay = 60
api.rrtmg_lw_ini(1004.0, kdata=api.STANDIN_KDATA, device=0)
if int(os.environ.get("SOAK_NDEV", "1")) > 1:          # several (virtual) devices on GPU 0: the host entries split their columns over them
    api.init_devices([0] * int(os.environ["SOAK_NDEV"]), kdata=api.STANDIN_KDATA)
rng = np.random.default_rng(1)
full = make_gcm_inputs(nmax, nlay, "aer_idrv", col0=7)
for k in ("co2vmr", "o2vmr"):                       # some rows uniform, some not
    full[k] = np.asfortranarray(np.full((nmax, nlay), float(np.asarray(full[k])[0, 0])))
def solve_arrays(d, out=None):
    sub = api.mcica_subcol_lw(d["ncol"], d["nlay"], 2, 140, 0, d["play"], d["cldfr"], d["cicewp"], d["cliqwp"], d["reice"], d["reliq"], d["taucld"])
    dd = dict(d)
    dd.update({k: sub[k] for k in ("cldfmcl", "ciwpmcl", "clwpmcl", "taucmcl", "reicmcl", "relqmcl")})
    return api.rrtmg_lw_mcica_from_dict(dd, icld=2)


solve = (lambda d, out=None: api.rrtmg_lw_mcica_subcol_from_dict(d, 140, 0, icld=2)) if mcica else solve_arrays if arrays else (lambda d, out=None: api.rrtmg_lw_from_dict(d, icld=2, out=out))
ref = solve(full)
names = ("uflx", "dflx", "hr", "uflxc", "dflxc", "hrc", "duflx_dt", "duflxc_dt")
t0, calls, cols = time.time(), 0, 0
while time.time() - t0 < secs:
    n = int(rng.integers(1, nmax + 1)) if rng.random() < 0.7 else int(rng.integers(1, 400))
    c0 = int(rng.integers(0, nmax - n + 1))
    d = dict(full)
    d["ncol"] = n
    for k, v in full.items():
        if isinstance(v, np.ndarray):
            d[k] = np.asfortranarray(v[:, c0:c0 + n, :] if (v.ndim == 3 and v.shape[0] == 16) else v[c0:c0 + n])
    out = api._out_arrays(n, nlay, d["idrv"])
    arrs = [v for v in list(d.values()) + list(out.values()) if isinstance(v, np.ndarray) and v.flags.f_contiguous and v.dtype == np.float64 and v.nbytes >= 4096]
    pinned = [v for v in arrs if rng.random() < 0.5]
    for v in pinned:
        api.host_register(v)
    reps = int(rng.integers(1, 4))
    if os.environ.get("SOAK_VERBOSE"):
        print(f"call n={n} c0={c0} reps={reps} pinned={[k for k, v in d.items() if any(v is w for w in pinned)]}", flush=True)
    for _ in range(reps):
        got = solve(d, out)
        for k in names:
            if not np.array_equal(got[k], ref[k][c0:c0 + n]):
                bad = np.argwhere(got[k] != ref[k][c0:c0 + n])
                print(f"MISMATCH {k}: n={n} c0={c0} pinned={len(pinned)}/{len(arrs)} first differing element {bad[0]} of {len(bad)}")
                sys.exit(1)
        calls += 1
        cols += n
    for v in pinned:
        api.host_unregister(v)
print(f"soak ok: {calls} calls, {cols} columns in {time.time() - t0:.0f} s")
