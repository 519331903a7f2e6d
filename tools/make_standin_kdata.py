#!/usr/bin/env python3
"""Generate the STAND-IN absorption-coefficient blob  rrtmg_lw_amd/data/standin.kdata.bin.

The reference mount lists its two k-data sources (src/rrtmg_lw_k_g.f90, data/rrtmg_lw.nc) in
.MISSING_LARGE_BLOBS: the real coefficients are not available in this build environment.  This
script synthesises tables with the *same names, shapes and orders of magnitude* so that every code
path (g-point reduction, every gather, both optical-depth regimes of the RT sweep) is exercised and
so that the product, the C oracle and the flang-built reference algorithm can be compared on
identical inputs.  Fluxes computed from these tables are NOT physical and are never compared with
the reference's golden OUTPUT_RRTM files; those tests activate only when real k-data is supplied
(see rrtmg_lw_amd/kdata.py).

Properties: deterministic (splitmix64 of band/array/element), every element distinct, smooth-ish in
pressure/temperature/mixture, spanning ~1e-6..3e1 along g; Planck fractions positive and summing to
one over the 16 original g-points of each band/mixture.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from rrtmg_lw_amd.blob import write_blob  # noqa: E402
from rrtmg_lw_amd.kspec import KSPEC, blob_name, shape_of  # noqa: E402

WT = np.array([0.1527534276, 0.1491729617, 0.1420961469, 0.1316886544, 0.1181945205, 0.1019300893,
               0.0832767040, 0.0626720116, 0.0424925000, 0.0046269894, 0.0038279891, 0.0030260086,
               0.0022199750, 0.0014140010, 0.0005330000, 0.0000750000])


def splitmix_uniform(seed, n):
    """n uniforms in [0,1) from splitmix64(seed + i)."""
    z = (np.uint64(seed) + np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15))
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) / float(1 << 53)


# log10 magnitude at g=1 and g=16 per array family
_RANGE = {
    "kao": (-5.5, 1.2), "kbo": (-5.0, 1.5),
    "selfrefo": (-4.0, 0.0), "forrefo": (-5.0, -1.0),
}
_MINOR = (-6.5, -0.5)
_VEC = (-1.0, 2.0)


def make_array(band, idx, name, bounds, kind, gdim):
    shp = shape_of(bounds)
    n = int(np.prod(shp))
    seed = (band * 1000003 + idx * 7919) & 0xFFFFFFFF
    u = splitmix_uniform(seed, n).reshape(shp, order="F")
    g_axis = 0 if gdim == 0 else len(shp) - 1
    gshape = [1] * len(shp)
    gshape[g_axis] = 16
    gfrac = (np.arange(16) / 15.0).reshape(gshape)
    if kind == "f":
        f = WT.reshape(gshape) * (0.6 + 0.8 * u)
        return f / f.sum(axis=g_axis, keepdims=True)
    if name in _RANGE:
        lo, hi = _RANGE[name]
    elif len(shp) == 1:
        lo, hi = _VEC
    else:
        lo, hi = _MINOR
    lo += 0.15 * ((band * 5) % 7 - 3) / 3.0
    logk = lo + (hi - lo) * gfrac ** 1.4
    # smooth dependence on the non-g axes (pressure / temperature / mixture), plus +-12% jitter
    for ax, m in enumerate(shp):
        if ax == g_axis or m == 1:
            continue
        s = [1] * len(shp)
        s[ax] = m
        x = (np.arange(m) / max(m - 1, 1)).reshape(s)
        logk = logk + (0.25 + 0.1 * ax) * (x - 0.5) * (1.0 - 0.5 * gfrac)
    return 10.0 ** logk * (0.88 + 0.24 * u)


def main():
    out = {}
    for band in range(1, 17):
        for idx, (name, bounds, kind, gdim) in enumerate(KSPEC[band]):
            out[blob_name(band, name)] = make_array(band, idx, name, bounds, kind, gdim)
    out["meta.standin"] = np.array([1], dtype=np.int32)
    dst = os.path.join(os.path.dirname(HERE), "rrtmg_lw_amd", "data", "standin.kdata.bin")
    write_blob(dst, out)
    print("wrote", dst, os.path.getsize(dst), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
