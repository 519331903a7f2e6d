"""The two k-data converters (rrtmg_lw_amd/kdata.py) on synthetic files with the structure of the real distributions:
a data-statement Fortran file in the style of rrtmg_lw_k_g.f90 and a classic netCDF file with the variables and
dimension order that src/rrtmg_lw_read_nc.f90 reads.  Round trip: stand-in blob -> file -> converter -> same arrays."""
import os

import numpy as np

from rrtmg_lw_amd.blob import read_blob
from rrtmg_lw_amd.kdata import NC_ABSORBERS, from_k_g_f90, from_netcdf
from rrtmg_lw_amd.kspec import KSPEC, blob_name

DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rrtmg_lw_amd", "data")


def _write_k_g_f90(path, k):
    with open(path, "w") as f:
        f.write("! synthetic file in the layout of rrtmg_lw_k_g.f90\n")
        for band in range(1, 17):
            f.write(f"      subroutine lw_kgb{band:02d}\n      use rrlw_kg{band:02d}\n      implicit none\n      save\n")
            for name, bounds, kind, gdim in KSPEC[band]:
                a = k[blob_name(band, name)]
                if a.ndim == 1:
                    slices = [((":",), a)]
                else:
                    lo = [b[0] for b in bounds]
                    slices = []
                    for idx in np.ndindex(*a.shape[1:]):
                        sl = (":",) + tuple(str(i + l) for i, l in zip(idx, lo[1:]))
                        slices.append((sl, a[(slice(None),) + idx]))
                for sl, vals in slices:
                    f.write(f"      {name}({', '.join(sl)}) = (/ &\n")
                    body = [f"{x:.17e}_rb" for x in vals]
                    for i in range(0, len(body), 4):
                        tail = ", &" if i + 4 < len(body) else " /)"
                        f.write("     & " + ",".join(body[i:i + 4]) + tail + "\n")
            f.write(f"      end subroutine lw_kgb{band:02d}\n\n")


def test_k_g_f90_roundtrip(tmp_path):
    k = read_blob(os.path.join(DATA, "standin.kdata.bin"))
    p = tmp_path / "rrtmg_lw_k_g.f90"
    _write_k_g_f90(p, k)
    got = from_k_g_f90(str(p))
    assert len(got) == len(k) - 1
    for name, a in got.items():
        assert np.array_equal(a, k[name]), name


def test_netcdf_roundtrip(tmp_path):
    from scipy.io import netcdf_file
    k = read_blob(os.path.join(DATA, "standin.kdata.bin"))
    p = str(tmp_path / "rrtmg_lw.nc")
    f = netcdf_file(p, "w")
    dims = dict(GPointSet=2, band=16, GPoint=16, keylower=9, keyupper=5, Tdiff=5, plower=13, pupper=47, Tself=10,
                Tforeign=4, T=19, Absorber=12)
    for n, s in dims.items():
        f.createDimension(n, s)
    var = lambda n, d: f.createVariable(n, "d", d)
    fl = var("PlanckFractionLowerAtmos", ("GPointSet", "band", "keylower", "GPoint"))
    fu = var("PlanckFractionUpperAtmos", ("GPointSet", "band", "keyupper", "GPoint"))
    kl = var("KeySpeciesAbsorptionCoefficientsLowerAtmos", ("GPointSet", "band", "GPoint", "plower", "Tdiff", "keylower"))
    ku = var("KeySpeciesAbsorptionCoefficientsUpperAtmos", ("GPointSet", "band", "GPoint", "pupper", "Tdiff", "keyupper"))
    sf = var("H20SelfAbsorptionCoefficients", ("GPointSet", "band", "GPoint", "Tself"))
    ff = var("H20ForeignAbsorptionCoefficients", ("GPointSet", "band", "GPoint", "Tforeign"))
    ml = var("AbsorptionCoefficientsLowerAtmos", ("GPointSet", "band", "Absorber", "GPoint", "T", "keylower"))
    mu = var("AbsorptionCoefficientsUpperAtmos", ("GPointSet", "band", "Absorber", "GPoint", "T", "keyupper"))
    for v in (fl, fu, kl, ku, sf, ff, ml, mu):
        v[:] = -1.0
    gas = {"mn2": "N2", "mn2o": "N2O", "mo3": "O3", "mco2": "CO2", "mco": "CO", "mo2": "O2"}
    vec = {"ccl4o": "CCL4", "cfc11adjo": "CFC11", "cfc12o": "CFC12", "cfc22adjo": "CFC22"}
    for band in range(1, 17):
        b = band - 1
        for name, bounds, kind, gdim in KSPEC[band]:
            a = k[blob_name(band, name)]
            if name.startswith("fracref"):
                dst = fl if name == "fracrefao" else fu
                a2 = a.reshape(16, -1)
                dst[0, b, :a2.shape[1], :] = a2.T
            elif name in ("kao", "kbo"):
                dst = kl if name == "kao" else ku
                a4 = a if a.ndim == 4 else a.reshape((1,) + a.shape)
                dst[0, b, :, :, :, :a4.shape[0]] = np.transpose(a4, (3, 2, 1, 0))
            elif name == "selfrefo":
                sf[0, b] = a.T
            elif name == "forrefo":
                ff[0, b] = a.T
            elif name in vec:
                ml[0, b, NC_ABSORBERS.index(vec[name]), :, 0, 0] = a
            else:
                region, g = name.split("_")
                dst = ml if region == "kao" else mu
                a3 = a if a.ndim == 3 else a.reshape((1,) + a.shape)
                dst[0, b, NC_ABSORBERS.index(gas[g]), :, :, :a3.shape[0]] = np.transpose(a3, (2, 1, 0))
    f.close()
    got = from_netcdf(p)
    assert len(got) == len(k) - 1
    for name, a in got.items():
        assert np.array_equal(a, k[name]), name
