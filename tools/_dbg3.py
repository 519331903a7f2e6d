import os, sys, numpy as np
sys.path.insert(0, ".")
from rrtmg_lw_amd import api
from rrtmg_lw_amd.synth import make_gcm_inputs
n = int(sys.argv[1]); nlay = int(sys.argv[2]); cfg = sys.argv[3]
api.rrtmg_lw_ini(1004.0, kdata=api.STANDIN_KDATA, device=0)
d = make_gcm_inputs(n, nlay, cfg, col0=7)
print("calling fused entry", n, nlay, cfg, flush=True)
r = api.rrtmg_lw_mcica_subcol_from_dict(d, 140, 0, icld=2)
print("ok", float(r["uflx"][0, 0]), flush=True)
