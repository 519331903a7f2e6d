import sys, os
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from rrtmg_lw_amd import api
from oracle.bindings import Oracle
from test_hip_parity import _special_cloud_inputs
api.rrtmg_lw_ini(1004.0, kdata=api.STANDIN_KDATA, device=0)
orc = Oracle()
ncol, nlay = 300, 60
d = _special_cloud_inputs(ncol, nlay, "toplayer")
for icld in (1, 2):
    got = api.rrtmg_lw_from_dict(d, icld=icld)
    ref = orc.rrtmg_lw(ncol, nlay, icld, d["idrv"], d)
    for k in ("uflx", "dflx", "uflxc", "dflxc", "hr", "hrc"):
        df = np.abs(got[k] - ref[k])
        c = np.unravel_index(df.argmax(), df.shape)
        print(icld, k, "max", df.max(), "at col,lev", c, "top3 levels max:", df[:, -3:].max(axis=0), "val", ref[k][c])
    c = np.unravel_index(np.abs(got["hr"] - ref["hr"]).argmax(), got["hr"].shape)[0]
    print(" col", c, "cldfr top", d["cldfr"][c, -1], "dp top", d["plev"][c, -2] - d["plev"][c, -1])
    for k in ("uflx", "dflx"):
        print("  ", k, (got[k][c, -3:] - ref[k][c, -3:]))
