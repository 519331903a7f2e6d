import os, sys, numpy as np
sys.path.insert(0, ".")
from rrtmg_lw_amd import api
from rrtmg_lw_amd.synth import make_gcm_inputs
mode = sys.argv[1]
nmax, nlay = 40000, 60
api.rrtmg_lw_ini(1004.0, kdata=api.STANDIN_KDATA, device=0)
full = make_gcm_inputs(nmax, nlay, "aer_idrv", col0=7)
for k in ("co2vmr", "o2vmr"):
    full[k] = np.asfortranarray(np.full((nmax, nlay), float(np.asarray(full[k])[0, 0])))
n, c0 = 30207, 9308
d = dict(full); d["ncol"] = n
for k, v in full.items():
    if isinstance(v, np.ndarray):
        d[k] = np.asfortranarray(v[:, c0:c0 + n, :] if (v.ndim == 3 and v.shape[0] == 16) else v[c0:c0 + n])
names = [k for k, v in d.items() if isinstance(v, np.ndarray) and v.flags.f_contiguous and v.dtype == np.float64 and v.nbytes >= 4096]
if mode == "none": pin = []
elif mode == "all": pin = names
elif mode.startswith("only:"): pin = mode[5:].split(",")
else: pin = names[::2]
print("pinned:", pin, flush=True)
for k in pin: api.host_register(d[k])
r = api.rrtmg_lw_mcica_subcol_from_dict(d, 140, 0, icld=2)
print("ok", float(r["uflx"][0, 0]), flush=True)
