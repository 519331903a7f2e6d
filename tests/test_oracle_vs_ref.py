"""Pins the C oracle (oracle/rrtmg_lw_oracle.c) against the REFERENCE's own Fortran.

 * against committed fixtures produced by the flang-built reference (tools/gen_ref_fixtures.py) - runs everywhere;
 * live against oracle/_ref/libref_nomcica.so when it is present (this container).
Both use the stand-in k-tables (the real k-data is stripped from the reference mount), so this pins the
ALGORITHM, not the physical fluxes; see tests/test_golden_examples.py for the latter.
"""
import glob
import os

import numpy as np
import pytest

from rrtmg_lw_amd.io_rrtm import read_input_rrtm
from rrtmg_lw_amd.synth import make_gcm_inputs, make_stress_inputs

G = os.path.join(os.path.dirname(__file__), "golden")
# same compiler-independent arithmetic on both sides: agreement is at rounding level, except where a last-bit
# difference moves a quantised LUT index (worth ~1e-6 W m-2)
TOL = 5e-6


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "ref_gcm_*.npz"))), ids=os.path.basename)
def test_gcm_fixture(oracle, path):
    f = np.load(path)
    ncol, nlay, icld = int(f["ncol"]), int(f["nlay"]), int(f["icld"])
    d = make_gcm_inputs(ncol, nlay, str(f["config"]), col0=int(f["col0"]))
    o = oracle.rrtmg_lw(ncol, nlay, icld, d["idrv"], d)
    assert o["icld"] == int(f["icld_out"])
    keys = ("uflx", "dflx", "hr", "uflxc", "dflxc", "hrc") + (("duflx_dt", "duflxc_dt") if d["idrv"] else ())
    for k in keys:
        assert np.abs(o[k] - f[k]).max() <= TOL, k


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "ref_stress_*.npz"))), ids=os.path.basename)
def test_stress_fixture(oracle, path):
    """High CO2 / N2O (minor-gas adjustment branch), temperatures beyond the table ends, laytrop = 0 and = nlay."""
    f = np.load(path)
    ncol, nlay, icld = int(f["ncol"]), int(f["nlay"]), int(f["icld"])
    d = make_stress_inputs(str(f["kind"]), ncol, nlay, col0=int(f["col0"]))
    o = oracle.rrtmg_lw(ncol, nlay, icld, 0, d)
    for k in ("uflx", "dflx", "hr", "uflxc", "dflxc", "hrc"):
        assert np.abs(o[k] - f[k]).max() <= TOL, k
    # the inputs do reach the branches they are meant for
    kind = str(f["kind"])
    lnp = np.log(np.array(d["play"]))
    if kind == "allupper":
        assert (lnp <= 4.56).all()
    if kind == "alllower":
        assert (lnp > 4.56).all()
    if kind == "cold":
        assert np.array(d["tlay"]).min() < 160.0
    if kind == "hot":
        assert np.array(d["tlay"]).max() > 339.0 and np.array(d["tsfc"]).max() > 339.0


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "ref_col_*.npz"))), ids=os.path.basename)
def test_column_fixture(oracle, path):
    f = np.load(path)
    j = lambda n: os.path.join(G, n) if n else None
    col = read_input_rrtm(j(str(f["inp"])), j(str(f["cld"])), j(str(f["aer"])))
    o = oracle.column(col)
    assert o["ncbands"] == int(f["ncbands"])
    assert np.allclose(o["taug"], f["taug"], rtol=1e-12, atol=0)
    assert np.allclose(o["fracs"], f["fracs"], rtol=1e-12, atol=0)
    for k in ("totuflux", "totdflux", "fnet", "htr", "totuclfl", "totdclfl", "fnetc", "htrc", "dtotuflux_dt", "dtotuclfl_dt"):
        assert np.abs(o[k] - f[k]).max() <= TOL, k
    for key in f.files:
        if key.startswith("b") and key.endswith("_up"):
            b = int(key[1:-3])
            ob = oracle.column(col, b, b, 99)
            assert np.abs(ob["totuflux"] - f[f"b{b}_up"]).max() <= TOL
            assert np.abs(ob["totdflux"] - f[f"b{b}_dn"]).max() <= TOL
            assert np.abs(ob["htr"] - f[f"b{b}_htr"]).max() <= TOL


def _oracle_subcolumns(oracle, f, d):
    """Sub-columns of every column with the oracle's generator; the Mersenne-Twister cases are generated column by
    column (one freshly seeded stream each), which is how the fixture was produced with the reference's 1-column generator."""
    ncol, nlay, icld, irng, seed = int(f["ncol"]), int(f["nlay"]), int(f["icld"]), int(f["irng"]), int(f["ims"]) * 140
    alpha = oracle.get_alpha(ncol, nlay, icld, int(f["idcor"]), 2000.0, f["dz"], f["lat"], int(f["juldat"]), d["cldfr"])
    args = lambda sl: (d["play"][sl], d["cldfr"][sl], d["cicewp"][sl], d["cliqwp"][sl], d["reice"][sl], d["reliq"][sl],
                       d["taucld"][:, sl, :], alpha[sl])
    if irng == 0:
        return alpha, oracle.mcica_subcol(ncol, nlay, icld, seed, 0, *args(slice(None)))
    parts = [oracle.mcica_subcol(1, nlay, icld, seed, 1, *args(slice(c, c + 1))) for c in range(ncol)]
    sub = {k: np.asfortranarray(np.concatenate([p[k] for p in parts], axis=1 if parts[0][k].ndim == 3 else 0))
           for k in ("cldfmcl", "ciwpmcl", "clwpmcl", "taucmcl", "reicmcl", "relqmcl")}
    return alpha, sub


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "ref_mcica_*.npz"))), ids=os.path.basename)
def test_mcica_fixture(oracle, path):
    """Generator (get_alpha, kissvec / Mersenne Twister, five overlaps): bit-exact masks; cldprmc + rtrnmc: fluxes."""
    f = np.load(path)
    ncol, nlay, icld = int(f["ncol"]), int(f["nlay"]), int(f["icld"])
    d = make_gcm_inputs(ncol, nlay, str(f["config"]), col0=int(f["col0"]))
    alpha, sub = _oracle_subcolumns(oracle, f, d)
    assert np.array_equal(alpha, f["alpha"])
    mask = np.unpackbits(f["mask"])[:140 * ncol * nlay].reshape((140, ncol, nlay), order="F")
    assert np.array_equal(sub["cldfmcl"], mask.astype(float))
    assert 0 < mask.mean() < 1
    for k3, ks in (("ciwpmcl", "ciwpsum"), ("clwpmcl", "clwpsum"), ("taucmcl", "taucsum")):
        assert np.array_equal(sub[k3].sum(axis=0), f[ks]), k3
    dd = dict(d)
    dd.update(sub)
    o = oracle.rrtmg_lw(ncol, nlay, icld, d["idrv"], dd, mcica=True)
    assert o["icld"] == int(f["icld_out"])
    keys = ("uflx", "dflx", "hr", "uflxc", "dflxc", "hrc") + (("duflx_dt", "duflxc_dt") if d["idrv"] else ())
    for k in keys:
        assert np.abs(o[k] - f[k]).max() <= TOL, k


def test_live_mcica_reference_if_built(oracle):
    from oracle.bindings import Reference
    if not Reference.available("mcica"):
        pytest.skip("oracle/_ref not built (needs /root/reference and flang)")
    ref = Reference("mcica")
    ncol, nlay = 5, 45
    d = make_gcm_inputs(ncol, nlay, "cloudy", col0=321)
    rng = np.random.default_rng(1)
    dz, lat = rng.uniform(100, 1500, (ncol, nlay)), rng.uniform(-90, 90, ncol)
    for icld in (1, 2, 3, 4, 5):
        alpha = oracle.get_alpha(ncol, nlay, icld, 1, 2500.0, dz, lat, 200, d["cldfr"])
        for c in range(ncol):
            assert np.array_equal(alpha[c], ref.get_alpha_1col(nlay, icld, 1, 2500.0, dz[c], lat[c], 200, d["cldfr"][c]))
        sub = oracle.mcica_subcol(ncol, nlay, icld, 280, 0, d["play"], d["cldfr"], d["cicewp"], d["cliqwp"], d["reice"], d["reliq"],
                                  d["taucld"], alpha)
        for c in range(ncol):
            r = ref.mcica_subcol_1col(nlay, icld, 2, 0, d["play"][c], d["cldfr"][c], d["cicewp"][c], d["cliqwp"][c], d["reice"][c],
                                      d["reliq"][c], d["taucld"][:, c, :], alpha[c])
            for k3, k2 in (("cldfmcl", "cldfmc"), ("ciwpmcl", "ciwpmc"), ("clwpmcl", "clwpmc"), ("taucmcl", "taucmc")):
                assert np.array_equal(sub[k3][:, c, :], r[k2]), (icld, c, k3)
        dd = dict(d)
        dd.update(sub)
        a = oracle.rrtmg_lw(ncol, nlay, icld, d["idrv"], dd, mcica=True)
        b = ref.rrtmg_lw(ncol, nlay, icld, d["idrv"], dd)
        for k in ("uflx", "dflx", "hr", "uflxc", "dflxc", "hrc"):
            assert np.abs(a[k] - b[k]).max() <= TOL, (icld, k)


def test_live_column_mcica_reference_if_built(oracle):
    """McICA flavour of the prepared-column entry (one sample of the column driver with imca = 1): reference example input,
    sub-columns from the reference's own generator, cldprmc -> rtrnmc."""
    from oracle.bindings import Reference
    if not Reference.available("mcica"):
        pytest.skip("oracle/_ref not built (needs /root/reference and flang)")
    ref = Reference("mcica")
    col = read_input_rrtm(os.path.join(G, "input_rrtm_MLS-cld-imca1-icld2"), os.path.join(G, "in_cld_rrtm-cld7"), None)
    nl = int(col["nlayers"])
    for ims, irng in ((1, 1), (5, 1), (2, 0)):
        sub = ref.mcica_subcol_1col(nl, int(col["icld"]), ims, irng, col["pavel"], col["cldfrac"], col["ciwp"], col["clwp"], col["rei"],
                                    col["rel"], col["tauc"], np.zeros(nl))
        a, b = oracle.column_mc(col, sub), ref.column_mc(col, sub)
        assert sub["cldfmc"].any()
        for k in ("totuflux", "totdflux", "fnet", "htr", "totuclfl", "totdclfl", "fnetc", "htrc"):
            assert np.abs(a[k] - b[k]).max() <= TOL, (ims, k)


def test_live_reference_if_built(oracle):
    from oracle.bindings import Reference
    if not Reference.available("nomcica"):
        pytest.skip("oracle/_ref not built (needs /root/reference and flang)")
    ref = Reference("nomcica")
    for cfg, nlay, icld in (("cloudy", 72, 2), ("cloudy", 60, 1), ("aer_idrv", 90, 2)):
        d = make_gcm_inputs(12, nlay, cfg, col0=999)
        a = oracle.rrtmg_lw(12, nlay, icld, d["idrv"], d)
        b = ref.rrtmg_lw(12, nlay, icld, d["idrv"], d)
        for k in ("uflx", "dflx", "hr", "uflxc", "dflxc", "hrc", "duflx_dt", "duflxc_dt"):
            assert np.abs(a[k] - b[k]).max() <= TOL, (cfg, k)


@pytest.mark.parametrize("kind", ["inflag0", "overcast", "thin"])
def test_live_reference_special_clouds(oracle, kind):
    """The oracle against the reference's Fortran on the cloud configurations of tests/test_hip_parity.py::
    test_special_cloud_configurations (prescribed optical depths, overcast / equal neighbouring fractions, fractions
    around rtrn's 1e-6 threshold)."""
    from oracle.bindings import Reference
    if not Reference.available("nomcica"):
        pytest.skip("oracle/_ref not built (needs /root/reference and flang)")
    import importlib.util
    spec = importlib.util.spec_from_file_location("thp", os.path.join(os.path.dirname(__file__), "test_hip_parity.py"))
    thp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(thp)
    ref = Reference("nomcica")
    d = thp._special_cloud_inputs(40, 60, kind)
    for icld in (1, 2):
        a = oracle.rrtmg_lw(40, 60, icld, d["idrv"], d)
        b = ref.rrtmg_lw(40, 60, icld, d["idrv"], d)
        for k in ("uflx", "dflx", "hr", "uflxc", "dflxc", "hrc"):
            assert np.abs(a[k] - b[k]).max() <= TOL, (kind, icld, k)
