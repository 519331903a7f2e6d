"""Host-side data plumbing: blob container, Fortran data-statement reader, static tables, k-data layout, text I/O."""
import os

import numpy as np
import pytest

from rrtmg_lw_amd.blob import read_blob, write_blob
from rrtmg_lw_amd.f90data import parse_f90_data
from rrtmg_lw_amd.io_rrtm import read_cld, read_input_rrtm, read_output_rrtm
from rrtmg_lw_amd.kspec import KSPEC, NGC, blob_name, shape_of

HERE = os.path.dirname(__file__)
DATA = os.path.join(os.path.dirname(HERE), "rrtmg_lw_amd", "data")
G = os.path.join(HERE, "golden")


def test_blob_roundtrip(tmp_path):
    a = {"x": np.arange(24, dtype=float).reshape(2, 3, 4), "i": np.array([1, 2, 3], dtype=np.int32), "s": np.array([2.5])}
    p = tmp_path / "t.bin"
    write_blob(p, a)
    b = read_blob(p)
    assert list(b) == ["x", "i", "s"]
    assert np.array_equal(b["x"], a["x"]) and b["x"].flags.f_contiguous
    assert b["i"].dtype == np.int32 and np.array_equal(b["i"], a["i"])


def test_f90_data_statements():
    src = '''
      subroutine lw_kgb99
      kao(:, 1, 2) = (/ &
     & 1.0e-01_rb, 2.0e-01_rb, &   ! comment
     & 3.0e-01_rb /)
      kao(2:3, 2, 2) = (/ 7._rb, 8.d0 /)
      scal = 0.25_rb
      fracrefao(:) = (/ 0.5_rb, .5_rb /)
      end subroutine lw_kgb99
      subroutine other
      kao(:, 1, 1) = (/ 9._rb, 9._rb, 9._rb /)
      end subroutine other
    '''
    arr, sc = parse_f90_data(src, {"kao": [(1, 3), (1, 2), (1, 2)], "fracrefao": [(1, 2)]}, scalars=("scal",), routine="lw_kgb99")
    assert np.array_equal(arr["kao"][:, 0, 1], [0.1, 0.2, 0.3])
    assert np.array_equal(arr["kao"][1:, 1, 1], [7.0, 8.0])
    assert np.isnan(arr["kao"][0, 0, 0])          # the other routine was not read
    assert sc["scal"] == 0.25 and np.array_equal(arr["fracrefao"], [0.5, 0.5])
    with pytest.raises(ValueError):
        parse_f90_data("      kao(:,1,1) = (/ 1._rb, 2._rb /)", {"kao": [(1, 3), (1, 2), (1, 2)]})


def test_static_tables():
    s = read_blob(os.path.join(DATA, "lw_static.bin"))
    assert s["totplnk"].shape == (181, 16) and s["chi_mls"].shape == (7, 59)
    assert np.allclose(np.diff(s["preflog"]), -0.2)                    # src/rrtmg_lw_setcoef.f90:442-444
    assert np.allclose(np.log(s["pref"]), s["preflog"], atol=1e-4)
    assert tuple(s["ngc"]) == NGC and s["ngs"][-1] == 140 and s["ngn"].sum() == 256
    assert np.all(np.diff(s["totplnk"], axis=0) > 0)                   # Planck integrals grow with temperature
    assert abs(s["wt"].sum() - 1.0) < 1e-7
    assert np.array_equal(np.bincount(s["ngb"])[1:], s["ngc"])


def test_standin_kdata_matches_kspec():
    k = read_blob(os.path.join(DATA, "standin.kdata.bin"))
    n = 0
    for band, spec in KSPEC.items():
        for name, bounds, kind, gdim in spec:
            a = k[blob_name(band, name)]
            assert a.shape == shape_of(bounds), (band, name)
            assert np.all(a > 0)
            if kind == "f":
                assert np.allclose(a.sum(axis=0), 1.0)
            n += 1
    assert n == len(k) - 1 and "meta.standin" in k


def test_oracle_reduction_layout(oracle):
    """256 -> 140 g-point reduction (src/rrtmg_lw_init.f90:149-192): group sums with rwgt weights / plain sums."""
    k = read_blob(os.path.join(DATA, "standin.kdata.bin"))
    s = read_blob(os.path.join(DATA, "lw_static.bin"))
    band = 10                                   # 6 reduced points: groups of 2,2,2,2,4,4
    ngn = s["ngn"][s["ngs"][band - 2]:s["ngs"][band - 1]]
    assert tuple(ngn) == (2, 2, 2, 2, 4, 4)
    kao = k["b10.kao"]
    ka = oracle.table(band, "ka").reshape((5, 13, 6), order="F")
    wt = s["wt"]
    first = 0
    for igc, cnt in enumerate(ngn):
        w = wt[first:first + cnt] / wt[first:first + cnt].sum()
        assert np.allclose(ka[:, :, igc], (kao[:, :, first:first + cnt] * w).sum(axis=2), rtol=1e-13)
        first += cnt
    fr = oracle.table(band, "fracrefa")
    assert np.allclose(fr.sum(), 1.0) and np.allclose(fr[4], k["b10.fracrefao"][8:12].sum())
    tau, ex, tfn = oracle.luts()                # src/rrtmg_lw_init.f90:125-142
    assert tau[0] == 0 and tau[-1] == 1e10 and ex[-1] == 1e-20 and tfn[-1] == 1.0
    i = 5000
    assert np.isclose(tau[i], (1 / 0.278) * 0.5 / 0.5) and np.isclose(ex[i], np.exp(-tau[i]))


def test_input_and_output_readers():
    c = read_input_rrtm(os.path.join(G, "input_rrtm_MLS-cld-imca1-icld5-idcor1"), os.path.join(G, "in_cld_rrtm-cld7"))
    assert (c["nlayers"], c["icld"], c["imca"], c["idcor"], c["juldat"], c["lat"]) == (51, 5, 1, 1, 1, 45.0)
    assert c["pz"][0] == 1013.0 and abs(c["pwvcm"] - 2.8763) < 1e-3 and c["tbound"] == 294.2
    cl = read_cld(os.path.join(G, "in_cld_rrtm-cld7"))
    assert (cl["inflag"], cl["iceflag"], cl["liqflag"]) == (2, 3, 1) and len(cl["layers"]) == 13
    x = read_input_rrtm(os.path.join(G, "input_rrtm_MLS-clr-xsec"))
    assert np.all(x["wx"][:, 0] > 0) and x["wx"][0, 0] < x["wx"][2, 0]     # CCL4 < CFC12 amounts
    a = read_input_rrtm(os.path.join(G, "input_rrtm_MLS-clr-aer12"), None, os.path.join(G, "in_aer_rrtm-aer12"))
    assert a["iaer"] == 10 and np.count_nonzero(a["tauaer"].sum(axis=1)) == 12
    o = read_output_rrtm(os.path.join(G, "output_rrtm_MLS-clr"))
    assert len(o) == 17 and o[0]["level"][0] == 0 and o[0]["uflx"][0] == 424.7960 and (o[16]["wn1"], o[16]["wn2"]) == (2600.0, 3250.0)
    s = read_input_rrtm(os.path.join(G, "input_rrtm_ICRCCM_sonde"))          # IATM = 1: tests/test_atmpth.py
    assert s["nlayers"] == 31 and s["tbound"] == 290.93 and abs(s["pz"][0] - 973.6) < 1e-9
