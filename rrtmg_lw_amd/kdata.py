"""Converters for the absorption-coefficient ("k-data") distributions of RRTMG_LW into the RRLWBLOB form that
`rrtmg_lw_hip_init` (and the oracle) read.

    python -m rrtmg_lw_amd.kdata  <rrtmg_lw_k_g.f90 | rrtmg_lw.nc>  data/rrtmg_lw.kdata.bin

Two sources exist upstream and both are stripped from the reference mount (.MISSING_LARGE_BLOBS):
  * src/rrtmg_lw_k_g.f90  - sixteen `lw_kgbNN` routines of array-constructor assignments, filling the arrays
    declared in modules/rrlw_kg01.f90 ... rrlw_kg16.f90 (shapes: rrtmg_lw_amd/kspec.py);
  * data/rrtmg_lw.nc      - netCDF with eight variables, read by src/rrtmg_lw_read_nc.f90 with the start/count
    slices reproduced below (always gPointSetNumber = 1, the 16-g set; the 140-g reduction is done at init).
Neither converter could be run against a real file in this environment; both are exercised on synthetic files of the
same structure (tests/test_kdata_converters.py).
"""
from __future__ import annotations

import sys

import numpy as np

from .blob import write_blob
from .f90data import parse_f90_data
from .kspec import KSPEC, blob_name, shape_of

# absorber order of the netCDF file: modules/rrlw_ncpar.f90:15-27
NC_ABSORBERS = ("N2", "CCL4", "CFC11", "CFC12", "CFC22", "H2O", "CO2", "O3", "N2O", "CO", "CH4", "O2")
_MINOR_GAS = {"mn2": "N2", "mn2o": "N2O", "mo3": "O3", "mco2": "CO2", "mco": "CO", "mo2": "O2"}
_VECTOR_GAS = {"ccl4o": "CCL4", "cfc11adjo": "CFC11", "cfc12o": "CFC12", "cfc22adjo": "CFC22"}


def from_k_g_f90(path):
    """Parse the data-statement distribution.  Every array of every band must be fully assigned."""
    text = open(path).read()
    out = {}
    for band in range(1, 17):
        shapes = {name: bounds for name, bounds, _, _ in KSPEC[band]}
        arrays, _ = parse_f90_data(text, shapes, routine=f"lw_kgb{band:02d}")
        for name, a in arrays.items():
            if np.isnan(a).any():
                raise ValueError(f"band {band}: {name} not completely assigned in {path}")
            out[blob_name(band, name)] = a
    return out


def from_netcdf(path):
    """Read rrtmg_lw.nc following src/rrtmg_lw_read_nc.f90 (e.g. :60-104 for band 1, :576-587 for the CFC vectors).
    netCDF dimensions appear reversed w.r.t. the Fortran reader: (GPointSet, band, ..., fastest)."""
    from scipy.io import netcdf_file          # classic netCDF-3; netCDF-4/HDF5 files cannot be read here
    f = netcdf_file(path, "r", mmap=False)
    v = f.variables
    frac_lo, frac_up = v["PlanckFractionLowerAtmos"][:], v["PlanckFractionUpperAtmos"][:]
    key_lo, key_up = v["KeySpeciesAbsorptionCoefficientsLowerAtmos"][:], v["KeySpeciesAbsorptionCoefficientsUpperAtmos"][:]
    self_, for_ = v["H20SelfAbsorptionCoefficients"][:], v["H20ForeignAbsorptionCoefficients"][:]
    min_lo, min_up = v["AbsorptionCoefficientsLowerAtmos"][:], v["AbsorptionCoefficientsUpperAtmos"][:]
    gs = 0                                     # gPointSetNumber = 1
    out = {}
    for band in range(1, 17):
        b = band - 1
        for name, bounds, kind, gdim in KSPEC[band]:
            shp = shape_of(bounds)
            if name in ("fracrefao", "fracrefbo"):
                src = frac_lo if name == "fracrefao" else frac_up      # (gset, band, key, g)
                nkey = shp[1] if len(shp) == 2 else 1
                a = src[gs, b, :nkey, :16].T                            # -> (g, key)
                a = a.reshape(shp, order="F") if len(shp) == 1 else a
            elif name in ("kao", "kbo"):
                src = key_lo if name == "kao" else key_up               # (gset, band, g, p, Tdiff, key)
                nkey = shp[0] if len(shp) == 4 else 1
                npr = shp[-2]
                a = np.transpose(src[gs, b, :16, :npr, :5, :nkey], (3, 2, 1, 0))   # (key, Tdiff, p, g)
                a = a.reshape(shp, order="F") if len(shp) == 3 else a
            elif name == "selfrefo":
                a = self_[gs, b, :16, :10].T                            # (gset, band, g, Tself)
            elif name == "forrefo":
                a = for_[gs, b, :16, :4].T
            elif name in _VECTOR_GAS:
                ab = NC_ABSORBERS.index(_VECTOR_GAS[name])             # (gset, band, absorber, g, T, key)
                a = min_lo[gs, b, ab, :16, 0, 0]
            else:
                region, gas = name.split("_")                           # kao_mn2o -> ("kao", "mn2o")
                src = min_lo if region == "kao" else min_up
                ab = NC_ABSORBERS.index(_MINOR_GAS[gas])
                nkey = shp[0] if len(shp) == 3 else 1
                a = np.transpose(src[gs, b, ab, :16, :19, :nkey], (2, 1, 0))       # (key, T, g)
                a = a.reshape(shp, order="F") if len(shp) == 2 else a
            a = np.asarray(a, dtype=np.float64)
            if a.shape != shp:
                raise ValueError(f"band {band} {name}: got {a.shape}, expected {shp}")
            out[blob_name(band, name)] = a
    f.close()
    return out


def convert(src, dst):
    arrays = from_netcdf(src) if src.endswith((".nc", ".cdf")) else from_k_g_f90(src)
    write_blob(dst, arrays)
    return len(arrays)


if __name__ == "__main__":
    if len(sys.argv) != 3:
        raise SystemExit(__doc__)
    print("wrote", convert(sys.argv[1], sys.argv[2]), "arrays to", sys.argv[2])
