"""Accuracy and speed of a build with narrower cell codes (-DRRLW_CODE_BITS=16|24; RRTMG_LW_HIP_LIB selects the library):
max |d flux| / |d heating rate| of the HIP path against (a) the reference-generated stress fixtures tests/golden/ref_stress_*.npz,
(b) the oracle on 24 576 cloudy 72-layer columns, (c) the oracle on 6 144 aerosol / dF/dT 137-layer columns.  One markdown row per case.
usage (GPU box):  RRTMG_LW_HIP_LIB=$PWD/exp/lib_full_code16.so python tools/code_bits_table.py 16"""
import glob, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (one HIP runtime per process)
from rrtmg_lw_amd import api
from rrtmg_lw_amd.synth import make_gcm_inputs, make_stress_inputs
from oracle.bindings import Oracle

tag = sys.argv[1] if len(sys.argv) > 1 else "32"
api.rrtmg_lw_ini(1004.0, kdata=api.STANDIN_KDATA, device=0)
orc = Oracle()


def diffs(got, ref, d=None):
    dfl = max(float(np.abs(got[k] - ref[k]).max()) for k in ("uflx", "dflx", "uflxc", "dflxc"))
    dhr = max(float(np.abs(got[k] - ref[k]).max()) for k in ("hr", "hrc"))
    rel = max(float((np.abs(got[k] - ref[k]) / np.maximum(np.abs(ref[k]), 1.0)).max()) for k in ("hr", "hrc"))
    return dfl, dhr, rel


rows = []
worst = [0.0, 0.0, 0.0]
for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "ref_stress_*.npz"))):
    z = np.load(f)
    ncol, nlay, icld = int(z["ncol"]), int(z["nlay"]), int(z["icld"])
    d = make_stress_inputs(str(z["kind"]), ncol, nlay, col0=int(z["col0"]))
    got = api.rrtmg_lw_from_dict(d, icld=icld, idrv=0)
    ref = {k: z[k] for k in ("uflx", "dflx", "hr", "uflxc", "dflxc", "hrc")}
    r = diffs(got, ref)
    worst = [max(a, b) for a, b in zip(worst, r)]
rows.append(("stress fixtures (reference Fortran outputs; CO2 x8, N2O x3, T -/+70 K, all-upper / all-lower)", *worst))
for cfg, nlay, ncol in (("cloudy", 72, 24576), ("aer_idrv", 137, 6144), ("cloudy_deep", 72, 8192)):
    d = make_gcm_inputs(ncol, nlay, cfg, col0=1000)
    got = api.rrtmg_lw_from_dict(d)
    ref = orc.rrtmg_lw(ncol, nlay, d["icld"], d["idrv"], d)
    rows.append((f"{ncol} columns '{cfg}', {nlay} layers vs the oracle", *diffs(got, ref)))
for name, a, b, c in rows:
    print(f"| {tag} | {name} | {a:.2e} | {b:.2e} | {c:.2e} |", flush=True)
api.finalize()
