"""IATM = 1: the layering of a level sounding, rrtmg_lw_amd/atmpth.py, in place of the reference's 7900-line RRTATM.

 * tests/golden/ref_rrtatm_*.npz - what the reference's own RRTATM (src/rrtatm.f, compiled where it lies by oracle/Makefile behind
   oracle/ref_rrtatm_harness.f90; tools/gen_ref_fixtures.py --rrtatm) returns for the reference's ICRCCM sonde example and for two inputs
   of ours (tools/make_iatm1_cases.py: a built-in model atmosphere; a user profile with every unit letter we support);
 * live against oracle/_ref/libref_rrtatm.so where it exists;
 * the level pressures of the reference's checked-in output_rrtm_ICRCCM_sonde and its k-independent surface emission;
 * on the GPU: the sonde example through the HIP library against the oracle.
"""
import glob
import os

import numpy as np
import pytest

from rrtmg_lw_amd import atmpth
from rrtmg_lw_amd.io_rrtm import read_input_rrtm, read_output_rrtm

G = os.path.join(os.path.dirname(__file__), "golden")
FIX = sorted(glob.glob(os.path.join(G, "ref_rrtatm_*.npz")))
RTOL = 5e-15        # same formulas in the same order: rounding level


def _layers(name, **kw):
    lines = open(os.path.join(G, name)).read().splitlines()
    p = next(i for i, ln in enumerate(lines) if ln.startswith("$")) + 3
    return atmpth.rrtatm(lines, p, **kw)[0]


def _rel(a, b):
    return (np.abs(a - b) / np.maximum(np.abs(b), 1e-300)).max()


@pytest.mark.parametrize("path", FIX, ids=os.path.basename)
def test_layering_matches_reference_rrtatm_fixture(path):
    f = np.load(path)
    m = _layers(str(f["inp"]))
    assert m["nlayers"] == int(f["nlayers"]) and m["nmol"] == int(f["nmol"])
    for k in ("pavel", "tavel", "pz", "tz", "altz", "wbrodl"):
        assert _rel(m[k], f[k]) <= RTOL, k
    assert _rel(m["wkl"][:7], f["wkl"]) <= RTOL
    assert (np.diff(m["pz"]) < 0).all() and (m["wbrodl"] > 0).all()


def test_layering_matches_live_reference_build():
    from oracle import bindings
    if not bindings.reference_rrtatm_available():
        pytest.skip("reference RRTATM build not present")
    for name in ("input_rrtm_ICRCCM_sonde", "input_rrtm_iatm1_units"):
        r = bindings.reference_rrtatm(os.path.join(G, name))
        m = _layers(name)
        assert r["nlayers"] == m["nlayers"]
        for k in ("pavel", "tavel", "pz", "tz", "wbrodl"):
            assert _rel(m[k], r[k]) <= RTOL, (name, k)
        assert _rel(m["wkl"][:7], r["wkl"]) <= RTOL, name


def test_sonde_example_reader_and_golden_levels():
    """The reference's example: 31 layers from 32 boundaries; its own output lists the level pressures (4 digits)."""
    col = read_input_rrtm(os.path.join(G, "input_rrtm_ICRCCM_sonde"))
    gold = read_output_rrtm(os.path.join(G, "output_rrtm_ICRCCM_sonde"))[0]
    assert col["nlayers"] == 31 and len(gold["pz"]) == 32
    assert (np.abs(col["pz"] - gold["pz"]) <= 5.1e-4 * gold["pz"]).all()          # printed with four significant digits
    assert col["tbound"] == 290.93
    # the reference's unset mean molecular weight of air: water vapour given in g/kg vanishes (atmpth.py) ...
    assert col["wkl"][0, :20].max() == 0.0 and col["pwvcm"] < 1e-3
    # ... and with the value its author commented out the sounding holds 1.7 cm of precipitable water
    wet = read_input_rrtm(os.path.join(G, "input_rrtm_ICRCCM_sonde"), airmwt=28.964)
    assert 1.5 < wet["pwvcm"] < 2.0 and (wet["wkl"][0, :20] > 0).all()
    assert np.allclose(wet["pz"], col["pz"]) and wet["coldry"][0] < col["coldry"][0]      # (the vapour displaces dry air)


def test_oracle_runs_the_sonde_example():
    """Surface emission does not depend on the absorption coefficients: level 0 of the reference's output (bar of
    tests/test_golden_planck.py: the real Planck fractions of a band sum to one within ~1e-5)."""
    from oracle.bindings import Oracle
    col = read_input_rrtm(os.path.join(G, "input_rrtm_ICRCCM_sonde"))
    gold = read_output_rrtm(os.path.join(G, "output_rrtm_ICRCCM_sonde"))[0]
    r = Oracle().column(col)
    assert abs(r["totuflux"][0] - gold["uflx"][0]) < 2e-3
    assert r["totdflux"][-1] == 0.0


def test_unsupported_records_fail_loudly(tmp_path):
    lines = open(os.path.join(G, "input_rrtm_iatm1_model6")).read().splitlines()
    p = next(i for i, ln in enumerate(lines) if ln.startswith("$")) + 3
    neg = list(lines)
    neg[p] = neg[p][:10] + "  -20" + neg[p][15:]                  # IBMAX < 0: boundaries in pressure
    with pytest.raises(NotImplementedError, match="IBMAX"):
        atmpth.rrtatm(neg, p)
    auto = list(lines)
    auto[p] = auto[p][:10] + "    0" + auto[p][15:]               # IBMAX = 0: automatic layering
    with pytest.raises(NotImplementedError, match="AUTLAY"):
        atmpth.rrtatm(auto, p)
    with pytest.raises(NotImplementedError, match="cross-sections"):
        atmpth.rrtatm(lines, p, ixsect=1)
    down = list(lines)
    down[p + 1] = "%10.4f%10.4f" % (60.0, 0.0)                    # H1 > H2 at zenith angle 0
    with pytest.raises(ValueError, match="H1 >= H2"):
        atmpth.rrtatm(down, p)


def test_column_amounts_are_consistent():
    """Air column from the hydrostatic relation: sum of all amounts of a layer = dp / (g m) within the model's gravity variation."""
    m = _layers("input_rrtm_iatm1_model6")
    total = m["wbrodl"] + m["wkl"].sum(axis=0)
    dp = (m["pz"][:-1] - m["pz"][1:]) * 1.0e3                      # dyn cm-2
    hydro = dp * 6.02214199e23 / (980.665 * 28.964)
    assert np.abs(total / hydro - 1.0).max() < 0.02
    assert (m["tavel"] < np.maximum(m["tz"][:-1], m["tz"][1:]) + 1e-9).all() and (m["tavel"] > np.minimum(m["tz"][:-1], m["tz"][1:]) - 1e-9).all()


@pytest.mark.gpu
def test_hip_runs_the_sonde_example(hip):
    from oracle.bindings import Oracle
    col = read_input_rrtm(os.path.join(G, "input_rrtm_ICRCCM_sonde"))
    gold = read_output_rrtm(os.path.join(G, "output_rrtm_ICRCCM_sonde"))[0]
    got = hip.run_columns([col], 1, 16)
    ref = Oracle().column(col)
    for k in ("totuflux", "totdflux", "fnet", "htr"):
        d = np.abs(got[k][0] - ref[k]).max()
        assert d <= (0.001 if k == "htr" else 0.01) and d <= 5e-5, k
    assert abs(got["totuflux"][0][0] - gold["uflx"][0]) < 2e-3          # the reference's own number (k-independent)
    wet = read_input_rrtm(os.path.join(G, "input_rrtm_ICRCCM_sonde"), airmwt=28.964)
    gw, rw = hip.run_columns([wet], 1, 16), Oracle().column(wet)
    assert np.abs(gw["totdflux"][0] - rw["totdflux"]).max() <= 5e-5
    assert gw["totdflux"][0][0] > got["totdflux"][0][0] + 10.0          # water vapour adds downward flux (stand-in coefficients: +30 W m-2)
