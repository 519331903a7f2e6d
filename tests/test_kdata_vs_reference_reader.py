"""kdata.from_netcdf against the reference's OWN netCDF reader, read as text.

The converter's earlier test wrote its file with a mirror of the converter, so a misread of the reader - dimension order, absorber
index, gPointSetNumber - would have passed.  Here the statements of src/rrtmg_lw_read_nc.f90 are parsed where they lie
(`nf90_inq_varid(ncid, "Variable", varID)`, `nf90_get_var(ncid, varID, target, start = (/.../), count = (/.../))`,
`call getAbsorberIndex('GAS', ab)`), the dimension parameters and the absorber list from modules/rrlw_ncpar.f90, and every read is
carried out on a synthetic file of distinct random numbers with nf90_get_var's semantics: the hyperslab start / count in Fortran
dimension order (the reverse of the file's C order), its elements assigned to the target array in array-element order.  Sixteen bands,
every array of kspec.KSPEC: from_netcdf must return exactly those numbers.  Runs where /root/reference is mounted (this container).
"""
import os
import re

import numpy as np
import pytest

from rrtmg_lw_amd.kdata import from_netcdf
from rrtmg_lw_amd.kspec import KSPEC, blob_name, shape_of

REF = "/root/reference"
READER = os.path.join(REF, "src", "rrtmg_lw_read_nc.f90")
NCPAR = os.path.join(REF, "modules", "rrlw_ncpar.f90")
pytestmark = pytest.mark.skipif(not (os.path.exists(READER) and os.path.exists(NCPAR)), reason="reference sources not mounted")


def _join_continuations(text):
    text = re.sub(r"!.*", "", text)                       # comments
    return re.sub(r"&\s*\n\s*&?", " ", text)


def _ncpar():
    t = _join_continuations(open(NCPAR).read())
    par = {m.group(1).lower(): int(m.group(2)) for m in re.finditer(r"\b(\w+)\s*=\s*(\d+)\b", t)}
    names = [n.strip() for n in re.findall(r"'([A-Za-z0-9 ]+)'", t[t.index("AbsorberNames"):t.index("/)")])]
    assert len(names) == par["absorber"] == 12
    return par, names


def _reads():
    """[(band, target, variable, start, count)] with start / count as lists of integers (Fortran order)"""
    par, absorbers = _ncpar()
    t = _join_continuations(open(READER).read())
    out = []
    for m in re.finditer(r"subroutine\s+lw_kgb(\d\d)\b(.*?)end\s+subroutine", t, flags=re.S | re.I):
        band, body = int(m.group(1)), m.group(2)
        env = dict(par)
        for pm in re.finditer(r"parameter\s*::\s*(.*)", body):
            for a in pm.group(1).split(","):
                k, _, v = a.partition("=")
                v = v.strip()
                env[k.strip().lower()] = 16 if re.fullmatch(r"no\d+", v) else int(v)      # no1 .. no16 = 16 (modules/rrlw_kgNN.f90)
        assert env["bandnumber"] == band and env["gpointsetnumber"] == 1
        var = None
        for st in re.finditer(r"getAbsorberIndex\('(\w+)'\s*,\s*ab\)|nf90_inq_varid\(ncid,\s*\"(\w+)\"|"
                              r"nf90_get_var\(ncid,\s*varID,\s*(\w+)(?:\([^)]*\))?,\s*start\s*=\s*\(/(.*?)/\)\s*,\s*count\s*=\s*\(/(.*?)/\)\)", body):
            if st.group(1):
                env["ab"] = absorbers.index(st.group(1)) + 1
            elif st.group(2):
                var = st.group(2)
            else:
                ev = lambda s: [int(x) if x.isdigit() else env[x.lower()] for x in (y.strip().replace("_im", "") for y in s.split(","))]
                out.append((band, st.group(3).lower(), var, ev(st.group(4)), ev(st.group(5))))
    return out


def _write_file(path, rng):
    from scipy.io import netcdf_file
    par, _ = _ncpar()
    f = netcdf_file(path, "w")
    dims = {n: par[n.lower()] for n in ("GPointSet", "band", "GPoint", "keylower", "keyupper", "Tdiff", "plower", "pupper", "Tself",
                                        "Tforeign", "T", "Absorber")}
    for n, s in dims.items():
        f.createDimension(n, s)
    # C order = the reverse of the reader's Fortran order, Appendix D of SURVEY.md
    layout = {
        "PlanckFractionLowerAtmos": ("GPointSet", "band", "keylower", "GPoint"),
        "PlanckFractionUpperAtmos": ("GPointSet", "band", "keyupper", "GPoint"),
        "KeySpeciesAbsorptionCoefficientsLowerAtmos": ("GPointSet", "band", "GPoint", "plower", "Tdiff", "keylower"),
        "KeySpeciesAbsorptionCoefficientsUpperAtmos": ("GPointSet", "band", "GPoint", "pupper", "Tdiff", "keyupper"),
        "H20SelfAbsorptionCoefficients": ("GPointSet", "band", "GPoint", "Tself"),
        "H20ForeignAbsorptionCoefficients": ("GPointSet", "band", "GPoint", "Tforeign"),
        "AbsorptionCoefficientsLowerAtmos": ("GPointSet", "band", "Absorber", "GPoint", "T", "keylower"),
        "AbsorptionCoefficientsUpperAtmos": ("GPointSet", "band", "Absorber", "GPoint", "T", "keyupper"),
    }
    data = {}
    for n, d in layout.items():
        v = f.createVariable(n, "d", d)
        a = rng.random(tuple(dims[x] for x in d))
        v[:] = a
        data[n] = a
    f.close()
    return data


def test_converter_agrees_with_the_reference_reader(tmp_path):
    reads = _reads()
    assert len(reads) == sum(len(v) for v in KSPEC.values())             # one nf90_get_var per array of every band
    path = str(tmp_path / "rrtmg_lw.nc")
    data = _write_file(path, np.random.default_rng(20240607))
    got = from_netcdf(path)
    seen = set()
    for band, target, var, start, count in reads:
        bounds = {n: b for n, b, _, _ in KSPEC[band]}[target]
        shp = shape_of(bounds)
        vf = data[var].transpose()                                         # Fortran index order
        assert len(start) == len(count) == vf.ndim, (band, target)
        slab = vf[tuple(slice(s - 1, s - 1 + c) for s, c in zip(start, count))]
        assert slab.shape == tuple(count) and slab.size == int(np.prod(shp)), (band, target, count, shp)
        want = slab.flatten(order="F").reshape(shp, order="F")            # nf90_get_var fills the target in array-element order
        a = got[blob_name(band, target)]
        assert a.shape == shp and np.array_equal(a, want), (band, target, var, start, count)
        seen.add((band, target))
    assert seen == {(b, n) for b in KSPEC for n, _, _, _ in KSPEC[b]}
