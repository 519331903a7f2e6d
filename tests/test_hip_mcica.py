"""GPU parity tests of the McICA flavour: sub-column generator (bit-exact masks), cldprmc + rtrnmc through
rrtmg_lw_hip_run_mcica, and the fused generator+solver entry, all against the CPU oracle (which is itself
bit-identical to the reference's Fortran: tests/test_oracle_vs_ref.py).
"""
import numpy as np
import pytest

from rrtmg_lw_amd.synth import make_gcm_inputs

pytestmark = pytest.mark.gpu

FLUX_TOL = 0.01      # W m-2     (BASELINE.json north_star)
HR_TOL = 0.001       # K day-1
TIGHT_FLUX = 5e-5    # see tests/test_hip_parity.py for why this is not 1e-12
TIGHT_HR = 5e-5
SUB = ("cldfmcl", "ciwpmcl", "clwpmcl", "reicmcl", "relqmcl", "taucmcl")


def _compare(got, ref, idrv, tag):
    dflux = max(np.abs(got[k] - ref[k]).max() for k in ("uflx", "dflx", "uflxc", "dflxc"))
    dhr = max(np.abs(got[k] - ref[k]).max() for k in ("hr", "hrc"))
    ddt = 0.0
    if idrv == 1:
        ddt = max(np.abs(got[k] - ref[k]).max() for k in ("duflx_dt", "duflxc_dt"))
    print(f"{tag}: max|dflux|={dflux:.3e} W/m2  max|dhr|={dhr:.3e} K/d  max|d(dF/dT)|={ddt:.3e}")
    assert np.isfinite(got["uflx"]).all() and np.isfinite(got["hr"]).all()
    assert dflux <= FLUX_TOL and dhr <= HR_TOL and ddt <= FLUX_TOL
    assert dflux <= TIGHT_FLUX and dhr <= TIGHT_HR and ddt <= TIGHT_FLUX
    assert got["icld"] == ref["icld"]


def _geometry(ncol, nlay, seed=3):
    rng = np.random.default_rng(seed)
    return rng.uniform(100, 1500, (ncol, nlay)), rng.uniform(-90, 90, ncol)


def _gen_args(d):
    return d["play"], d["cldfr"], d["cicewp"], d["cliqwp"], d["reice"], d["reliq"], d["taucld"]


@pytest.mark.parametrize("icld,idcor,juldat", [(4, 0, 10), (4, 1, 150), (5, 1, 300), (5, 0, 200), (2, 1, 100)])
def test_get_alpha(hip, oracle, icld, idcor, juldat):
    ncol, nlay = 130, 47
    d = make_gcm_inputs(ncol, nlay, "cloudy")
    dz, lat = _geometry(ncol, nlay)
    got = hip.get_alpha(ncol, nlay, icld, idcor, 2500.0, dz, lat, juldat, d["cldfr"])
    ref = oracle.get_alpha(ncol, nlay, icld, idcor, 2500.0, dz, lat, juldat, d["cldfr"])
    np.testing.assert_allclose(got, ref, rtol=4e-16 * 8, atol=0)      # device exp vs libm exp: a few ulp


@pytest.mark.parametrize("irng", [0, 1])
@pytest.mark.parametrize("icld", [1, 2, 3, 4, 5])
def test_subcolumn_generator_is_bit_exact(hip, oracle, icld, irng):
    """kissvec (one stream per column) and Mersenne Twister (one stream over all columns): identical cloud masks,
    water paths and optical depths (src/mcica_subcol_gen_lw.f90:183-703)."""
    ncol, nlay = 70, 40          # 70 > one generator block
    d = make_gcm_inputs(ncol, nlay, "cloudy", col0=11)
    dz, lat = _geometry(ncol, nlay)
    alpha = oracle.get_alpha(ncol, nlay, icld, 1, 2500.0, dz, lat, 100, d["cldfr"])
    for permuteseed in (140, 3 * 140):
        got = hip.mcica_subcol_lw(ncol, nlay, icld, permuteseed, irng, *_gen_args(d), alpha)
        ref = oracle.mcica_subcol(ncol, nlay, icld, permuteseed, irng, *_gen_args(d), alpha)
        assert 0.01 < ref["cldfmcl"].mean() < 0.9
        for k in SUB:
            assert np.array_equal(got[k], ref[k]), (k, icld, irng, permuteseed)
        assert got["irng"] == ref["irng"]


@pytest.mark.parametrize("icld", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("ncol,nlay,permuteseed", [(37, 4, 0), (21, 72, 1), (16, 9, 2), (53, 137, 7 * 140), (20, 203, 3), (17, 192, 5)])
def test_subcolumn_generator_edge_values(hip, oracle, icld, ncol, nlay, permuteseed):
    """The kissvec kernel reaches every sub-column's place in the column's stream by jump-ahead and compares integers with
    thresholds instead of deviates with 1 - cldfrac / alpha: cloud fractions and overlap parameters at and next to the ends of the
    deviates' range (0, below cldmin, 5e-8, 1 - 5e-8, 1; alpha 0, 1e-9, 0.99999995, 1), seeds 0 / 1 / 2 (sequential start of the
    jump), column counts that do not fill a work-group, the shortest column the generator accepts and the reference's tallest
    (mxlay = 203, modules/parrrtm.f90: more than 48 KB of thresholds in LDS; 192 layers are exactly 48 KB)."""
    d = make_gcm_inputs(ncol, nlay, "cloudy", col0=5)
    rng = np.random.default_rng(100 * icld + nlay)
    special = np.array([0.0, 1e-21, 5e-8, 9.3e-8, 1.1e-7, 0.3, 0.5, 1 - 1.1e-7, 1 - 9.3e-8, 1 - 5e-8, 1.0])
    cf = np.where(rng.random((ncol, nlay)) < 0.6, special[rng.integers(0, special.size, (ncol, nlay))], rng.random((ncol, nlay)))
    if ncol > 32:
        cf[:16] = 0.0                      # a work-group of cloud-free columns (no jump, no draws) ...
        cf[16:32, nlay // 3:] = 0.0        # ... and one whose walk ends a third of the way up
    d["cldfr"] = np.asfortranarray(cf)
    aspecial = np.array([0.0, 1e-9, 9.3e-8, 0.2, 0.9, 1 - 1.1e-7, 0.99999995, 1 - 1e-9, 1.0])
    alpha = np.where(rng.random((ncol, nlay)) < 0.6, aspecial[rng.integers(0, aspecial.size, (ncol, nlay))], rng.random((ncol, nlay)))
    alpha = np.asfortranarray(alpha)
    got = hip.mcica_subcol_lw(ncol, nlay, icld, permuteseed, 0, *_gen_args(d), alpha)
    ref = oracle.mcica_subcol(ncol, nlay, icld, permuteseed, 0, *_gen_args(d), alpha)
    assert 0.05 < ref["cldfmcl"].mean() < 0.95
    for k in SUB:
        assert np.array_equal(got[k], ref[k]), (k, icld, ncol, nlay, permuteseed)


@pytest.mark.parametrize("icld,ncol,nlay,seed", [(2, 2100, 70, 11), (5, 1500, 60, 4357), (3, 70000, 5, 1), (1, 2400, 55, 11)])
def test_mersenne_twister_stream_in_chunks(hip, oracle, icld, ncol, nlay, seed):
    """irng = 1 is ONE MT19937 stream over (sub-column, column, layer).  The device cuts every sub-column's slab into chunks of
    >= 65 536 deviates and reaches the state at each chunk's start by jump-ahead (k_mt_jump): slabs of 2 - 3 chunks
    (and, at icld = 3, a slab of one deviate per column), against the oracle's serial stream."""
    d = make_gcm_inputs(ncol, nlay, "cloudy", col0=3)
    rng = np.random.default_rng(seed)
    alpha = np.asfortranarray(rng.random((ncol, nlay)))
    d["cldfr"] = np.asfortranarray(np.where(rng.random((ncol, nlay)) < 0.5, rng.random((ncol, nlay)), 0.0))      # (cloud in short columns too)
    got = hip.mcica_subcol_lw(ncol, nlay, icld, seed, 1, *_gen_args(d), alpha)
    ref = oracle.mcica_subcol(ncol, nlay, icld, seed, 1, *_gen_args(d), alpha)
    assert got["irng"] == ref["irng"] == 1 and 0.05 < ref["cldfmcl"].mean() < 0.95
    for k in ("cldfmcl", "ciwpmcl", "clwpmcl"):
        assert np.array_equal(got[k], ref[k]), (k, icld, ncol, nlay, seed)
    # the same seed again (cached chunk states), then another one
    again = hip.mcica_subcol_lw(ncol, nlay, icld, seed, 1, *_gen_args(d), alpha)
    assert np.array_equal(again["cldfmcl"], ref["cldfmcl"])
    other = hip.mcica_subcol_lw(ncol, nlay, icld, seed + 1, 1, *_gen_args(d), alpha)      # the second cached set
    assert not np.array_equal(other["cldfmcl"], ref["cldfmcl"])
    third = hip.mcica_subcol_lw(ncol, nlay, icld, seed, 1, *_gen_args(d), alpha)
    assert np.array_equal(third["cldfmcl"], ref["cldfmcl"])


def test_subcolumn_generator_flags(hip, oracle):
    ncol, nlay = 64, 33
    d = make_gcm_inputs(ncol, nlay, "cloudy")
    # irng is normalised to 0/1 (:442); alpha may be omitted for icld < 4
    got = hip.mcica_subcol_lw(ncol, nlay, 2, 140, 7, *_gen_args(d))
    assert got["irng"] == 1
    ref = oracle.mcica_subcol(ncol, nlay, 2, 140, 1, *_gen_args(d), np.zeros((ncol, nlay)))
    assert np.array_equal(got["cldfmcl"], ref["cldfmcl"])
    # icld = 0 returns without touching the outputs (:265)
    got = hip.mcica_subcol_lw(ncol, nlay, 0, 140, 0, *_gen_args(d))
    assert not got["cldfmcl"].any()
    with pytest.raises(hip.RrtmgLwError, match="INVALID ICLD"):
        hip.mcica_subcol_lw(ncol, nlay, 6, 140, 0, *_gen_args(d))
    # the kissvec seeds need pressures ordered surface -> top (:463-466)
    bad = dict(d)
    bad["play"] = np.asfortranarray(d["play"][:, ::-1])
    with pytest.raises(hip.RrtmgLwError, match="KISSVEC SEED GENERATOR REQUIRES PMID"):
        hip.mcica_subcol_lw(ncol, nlay, 2, 140, 0, *_gen_args(bad))


def _with_subcolumns(oracle, d, icld, irng=0, permuteseed=140, alpha=None):
    ncol, nlay = d["ncol"], d["nlay"]
    alpha = np.zeros((ncol, nlay)) if alpha is None else alpha
    sc = oracle.mcica_subcol(ncol, nlay, icld, permuteseed, irng, *_gen_args(d), alpha)
    dd = dict(d)
    dd.update({k: sc[k] for k in SUB})
    return dd


@pytest.mark.parametrize("config,nlay,icld,ncol,flags", [
    ("cloudy", 72, 2, 300, None),            # synth default flags (inflag 2, iceflag 3, liqflag 1)
    ("cloudy", 72, 1, 130, (2, 0, 0)),
    ("cloudy", 51, 3, 100, (2, 1, 1)),
    ("cloudy", 72, 2, 100, (2, 2, 1)),
    ("cloudy", 40, 2, 70, (0, 0, 0)),         # inflag 0: optical depths taken from taucmcl
    ("aer_idrv", 137, 2, 130, None),          # aerosol + dF/dT, 137 layers
    ("aer_idrv", 33, 1, 64, (2, 3, 1)),
    ("cloudy", 72, 9, 65, None),              # out-of-range icld is reset to 2 (src/rrtmg_lw_rad.f90:469)
    ("cloudy_orography", 72, 2, 341, None),   # terrain-following grid, ragged last window with a cloudy last column (k_layer<mcica, wide>)
    ("cloudy_orography", 72, 3, 117, None),
])
def test_mcica_entry_matches_oracle(hip, oracle, config, nlay, icld, ncol, flags, sweeps):
    d = make_gcm_inputs(ncol, nlay, config, col0={341: 282237, 117: 844914}.get(ncol, 77))
    if flags is not None:
        d["inflglw"], d["iceflglw"], d["liqflglw"] = flags
        if flags[1] == 0:
            d["reice"] = np.asfortranarray(np.clip(d["reice"], 10.0, 30.0))
        if flags[1] == 1:
            d["reice"] = np.asfortranarray(np.clip(d["reice"], 13.0, 130.0))
    dd = _with_subcolumns(oracle, d, min(max(icld, 1), 3))
    got = hip.rrtmg_lw_mcica_from_dict(dd, icld=icld)
    ref = oracle.rrtmg_lw(ncol, nlay, icld, d["idrv"], dd, mcica=True)
    _compare(got, ref, d["idrv"], f"mcica {config} L{nlay} icld{icld} flags{flags}")
    assert np.abs(ref["dflx"] - ref["dflxc"]).max() > 1.0 or flags == (0, 0, 0)      # clouds actually matter


def test_mcica_entry_ignores_subcolumn_arrays_where_no_subcolumn_has_cloud(hip, oracle):
    """Layers whose sub-column cloud fractions are all below cldmin for a column batch: cldprmc reads nothing else of them
    (src/rrtmg_lw_cldprmc.f90:182-183) and the host-pointer entry does not scan or copy the other sub-column arrays there - junk in
    them changes nothing.  Several batches; a layer that is cloud-free in the first batch only."""
    ncol, nlay = 500, 40
    d = make_gcm_inputs(ncol, nlay, "cloudy", col0=31)
    cf = np.array(d["cldfr"]); cf[:200, 9] = 0.0; cf[200:, 9] = 0.6
    for k in ("cicewp", "cliqwp"):
        a = np.array(d[k]); a[200:, 9] = 15.0; d[k] = np.asfortranarray(a)
    d["cldfr"] = np.asfortranarray(cf)
    dd = _with_subcolumns(oracle, d, 2)
    free = (dd["cldfmcl"] < 1e-20).all(axis=(0, 1))                       # per layer, all sub-columns of all columns
    assert free.any() and not free.all()
    dj = dict(dd)
    for k, v in (("ciwpmcl", 900.0), ("clwpmcl", 800.0), ("taucmcl", 4.0)):
        a = np.array(dd[k]); a[:, :, free] = v; a[:, :200, 9] = v; dj[k] = np.asfortranarray(a)
    for k, v in (("reicmcl", 1e4), ("relqmcl", 0.001)):
        a = np.array(dd[k]); a[:, free] = v; a[:200, 9] = v; dj[k] = np.asfortranarray(a)
    hip.set_batch(200)
    try:
        a = hip.rrtmg_lw_mcica_from_dict(dd, icld=2)
        b = hip.rrtmg_lw_mcica_from_dict(dj, icld=2)
    finally:
        hip.set_batch(262144)
    for k in ("uflx", "dflx", "hr", "uflxc", "dflxc", "hrc"):
        assert np.array_equal(a[k], b[k]), k
    ref = oracle.rrtmg_lw(ncol, nlay, 2, d["idrv"], dj, mcica=True)
    _compare(b, ref, d["idrv"], "junk in cloud-free layers of the sub-column arrays")


@pytest.mark.parametrize("icld", [2, 5])
def test_mcica_cloud_top_changes_from_block_to_block(hip, oracle, icld, sweeps):
    """rtrnmc through the array entry and through the fused generator entry on cloud decks whose top differs from one 64-column
    block to the next (0 / 14 / nlay / 1 ...): the hand-off level between the clear-sky sweeps and the cloud-zone sweep is per group
    of blocks (k_blocksort), the generator's own walk ends per 16 columns."""
    from test_hip_parity import _block_top_inputs, _compare_thin_layers
    nlay = 60
    ncol = 64 * 27 + 9
    d = _block_top_inputs(ncol, nlay, [0, 14, nlay, 1, 30, 0, 14, 59, 2, 45, 14, 14, 7], seed=9,
                          bases=[1, 9, 40, 1, 30, 1, 1, 50, 2, 20, 14, 3, 7])          # (cloud bases change from block to block as well: k_sweepz's lbot)
    dz, lat = _geometry(ncol, nlay)
    alpha = oracle.get_alpha(ncol, nlay, icld, 0, 2500.0, dz, lat, 100, d["cldfr"])
    dd = _with_subcolumns(oracle, d, icld, alpha=alpha)
    ref = oracle.rrtmg_lw(ncol, nlay, icld, d["idrv"], dd, mcica=True)
    got = hip.rrtmg_lw_mcica_from_dict(dd, icld=icld)
    _compare_thin_layers(got, ref, d, d["idrv"], f"mcica block tops, arrays, icld{icld}")
    fused = hip.rrtmg_lw_mcica_subcol_from_dict(d, 140, 0, icld=icld, alpha=alpha)
    for k in ("uflx", "dflx", "hr", "uflxc", "dflxc", "hrc"):
        assert np.array_equal(got[k], fused[k]), k
    assert np.abs(ref["dflx"] - ref["dflxc"]).max() > 1.0


def test_fused_entry_and_generator_fan_out_over_devices(hip, oracle):
    """rrtmg_lw_hip_init_devices with three (virtual) devices: the fused generator + solver host entry and the stand-alone generator split
    their columns over the devices when the generator is kissvec (a stream per column, src/mcica_subcol_gen_lw.f90:463-474) - results and
    masks equal the one-device call bit for bit, for exponential-random overlap as well; with the Mersenne Twister (ONE stream over all
    columns, :497-503) the call stays on the first device and still gives the same numbers."""
    ncol, nlay = 1100, 40
    d = make_gcm_inputs(ncol, nlay, "cloudy", col0=8)
    dz, lat = _geometry(ncol, nlay)
    alpha = oracle.get_alpha(ncol, nlay, 5, 0, 2500.0, dz, lat, 100, d["cldfr"])
    cases = [(2, 0, None), (5, 0, alpha), (2, 1, None)]
    one = [hip.rrtmg_lw_mcica_subcol_from_dict(d, 7, irng, icld=icld, alpha=al) for icld, irng, al in cases]
    one_gen = [hip.mcica_subcol_lw(ncol, nlay, icld, 7, irng, *_gen_args(d), alpha=al) for icld, irng, al in cases]
    try:
        hip.init_devices([0, 0, 0], kdata=hip.STANDIN_KDATA)
        three = [hip.rrtmg_lw_mcica_subcol_from_dict(d, 7, irng, icld=icld, alpha=al) for icld, irng, al in cases]
        three_gen = [hip.mcica_subcol_lw(ncol, nlay, icld, 7, irng, *_gen_args(d), alpha=al) for icld, irng, al in cases]
    finally:
        hip.rrtmg_lw_ini(1004.0, kdata=hip.STANDIN_KDATA, device=0)
    for a, b, (icld, irng, _) in zip(one, three, cases):
        for k in ("uflx", "dflx", "hr", "uflxc", "dflxc", "hrc"):
            assert np.array_equal(a[k], b[k]), (icld, irng, k)
    for a, b, (icld, irng, _) in zip(one_gen, three_gen, cases):
        for k in SUB:
            assert np.array_equal(a[k], b[k]), (icld, irng, k)
    sc = oracle.mcica_subcol(ncol, nlay, 5, 7, 0, *_gen_args(d), alpha)
    assert np.array_equal(three_gen[1]["cldfmcl"], sc["cldfmcl"])


def test_mcica_tallest_column(hip, oracle):
    """nlay = 603 through the McICA array entry and the fused kissvec entry (the generator's thresholds: 603 x 16 x 16 B = 154 KB of LDS)."""
    from test_hip_parity import _compare_thin_layers
    ncol, nlay, icld = 20, 603, 2
    d = make_gcm_inputs(ncol, nlay, "cloudy", col0=40)
    dd = _with_subcolumns(oracle, d, icld)
    ref = oracle.rrtmg_lw(ncol, nlay, icld, d["idrv"], dd, mcica=True)
    got = hip.rrtmg_lw_mcica_from_dict(dd, icld=icld)
    _compare_thin_layers(got, ref, d, d["idrv"], "mcica 603 layers")
    fused = hip.rrtmg_lw_mcica_subcol_from_dict(d, 140, 0, icld=icld)
    for k in ("uflx", "dflx", "hr", "uflxc", "dflxc", "hrc"):
        assert np.array_equal(got[k], fused[k]), k


def test_fused_entry_takes_optical_depths_from_taucld(hip, oracle):
    """inflglw = 0: the fused generator + solver entry must expand the grid-mean taucld exactly as the generator's taucmcl does
    (src/mcica_subcol_gen_lw.f90:664-680) - the host entry has to ship taucld whole."""
    ncol, nlay, icld = 200, 40, 2
    d = make_gcm_inputs(ncol, nlay, "cloudy", col0=21)
    rng = np.random.default_rng(2)
    cf = np.array(d["cldfr"])
    d["taucld"] = np.asfortranarray(3.0 * rng.random((16, ncol, nlay)) * (cf[None, :, :] > 0))
    d["inflglw"], d["iceflglw"], d["liqflglw"] = 0, 0, 0
    dd = _with_subcolumns(oracle, d, icld)
    ref = oracle.rrtmg_lw(ncol, nlay, icld, d["idrv"], dd, mcica=True)
    arrays = hip.rrtmg_lw_mcica_from_dict(dd, icld=icld)
    fused = hip.rrtmg_lw_mcica_subcol_from_dict(d, 140, 0, icld=icld)
    _compare(arrays, ref, d["idrv"], "mcica inflag 0, arrays")
    for k in ("uflx", "dflx", "hr", "uflxc", "dflxc", "hrc"):
        assert np.array_equal(arrays[k], fused[k]), k
    assert np.abs(ref["dflx"] - ref["dflxc"]).max() > 1.0


def test_mcica_icld0_ignores_cloud_arrays(hip, oracle):
    d = make_gcm_inputs(100, 72, "cloudy")
    dd = _with_subcolumns(oracle, d, 2)
    got = hip.rrtmg_lw_mcica_from_dict(dd, icld=0)
    ref = oracle.rrtmg_lw(100, 72, 0, 0, dd, mcica=True)
    _compare(got, ref, 0, "mcica icld=0")
    assert np.array_equal(got["uflx"], got["uflxc"])


def test_mcica_fractional_and_partial_cells(hip, oracle):
    """cldfmcl values other than 0/1 and cells with water but no cloud flag follow cldprmc/rtrnmc literally
    (only cldfmc == 1 contributes optical depth: src/rrtmg_lw_rtrnmc.f90:311)."""
    ncol, nlay = 90, 45
    d = make_gcm_inputs(ncol, nlay, "cloudy")
    dd = _with_subcolumns(oracle, d, 1)
    rng = np.random.default_rng(5)
    cf = dd["cldfmcl"].copy()
    pick = rng.random(cf.shape) < 0.05
    cf[pick] = 0.5
    dd["cldfmcl"] = np.asfortranarray(cf)
    dd["ciwpmcl"] = np.asfortranarray(np.where(rng.random(cf.shape) < 0.3, 0.0, dd["ciwpmcl"]))
    got = hip.rrtmg_lw_mcica_from_dict(dd, icld=1)
    ref = oracle.rrtmg_lw(ncol, nlay, 1, d["idrv"], dd, mcica=True)
    _compare(got, ref, d["idrv"], "mcica fractional cells")


def test_mcica_batching_is_transparent(hip, oracle):
    d = make_gcm_inputs(600, 40, "cloudy", col0=5)
    dd = _with_subcolumns(oracle, d, 2)
    hip.set_batch(131072)
    one = hip.rrtmg_lw_mcica_from_dict(dd)
    hip.set_batch(256)
    many = hip.rrtmg_lw_mcica_from_dict(dd)
    fused_many = hip.rrtmg_lw_mcica_subcol_from_dict(d, 140, 0)
    hip.set_batch(131072)
    fused_one = hip.rrtmg_lw_mcica_subcol_from_dict(d, 140, 0)
    for k in ("uflx", "dflx", "hr", "uflxc", "dflxc", "hrc"):
        assert np.array_equal(one[k], many[k]), k
        assert np.array_equal(fused_one[k], fused_many[k]), k
        assert np.array_equal(one[k], fused_one[k]), k          # mask path == array path, bit for bit


def test_mcica_errors(hip, oracle):
    d = make_gcm_inputs(64, 40, "cloudy")
    dd = _with_subcolumns(oracle, d, 2)
    dd["inflglw"] = 1
    with pytest.raises(hip.RrtmgLwError, match="INFLAG = 1 OPTION NOT AVAILABLE WITH MCICA"):
        hip.rrtmg_lw_mcica_from_dict(dd)
    dd["inflglw"] = 2
    dd["reicmcl"] = np.asfortranarray(np.full((64, 40), 500.0))
    with pytest.raises(hip.RrtmgLwError, match="ICE GENERALIZED EFFECTIVE SIZE OUT OF BOUNDS"):
        hip.rrtmg_lw_mcica_from_dict(dd)


@pytest.mark.parametrize("icld,irng,config,nlay,ncol", [
    (2, 0, "cloudy", 72, 300),
    (1, 0, "cloudy", 72, 100),
    (3, 0, "cloudy", 51, 100),
    (4, 0, "cloudy", 72, 130),
    (5, 0, "aer_idrv", 72, 130),
    (2, 1, "cloudy", 40, 70),        # Mersenne Twister stream
    (5, 1, "aer_idrv", 33, 64),
    (0, 0, "cloudy", 72, 64),
])
def test_fused_generator_and_solver_matches_oracle(hip, oracle, icld, irng, config, nlay, ncol, sweeps):
    """rrtmg_lw_hip_run_mcica_subcol == mcica_subcol_lw followed by the McICA rrtmg_lw."""
    d = make_gcm_inputs(ncol, nlay, config, col0=31)
    dz, lat = _geometry(ncol, nlay)
    alpha = oracle.get_alpha(ncol, nlay, icld, 1, 2500.0, dz, lat, 100, d["cldfr"])
    got = hip.rrtmg_lw_mcica_subcol_from_dict(d, 280, irng, alpha=alpha, icld=icld)
    if icld == 0:
        from oracle.bindings import _subcol_outputs
        dd = dict(d)
        dd.update(_subcol_outputs(ncol, nlay))
    else:
        dd = _with_subcolumns(oracle, d, icld, irng=irng, permuteseed=280, alpha=alpha)
    ref = oracle.rrtmg_lw(ncol, nlay, icld, d["idrv"], dd, mcica=True)
    _compare(got, ref, d["idrv"], f"fused icld{icld} irng{irng} {config} L{nlay}")


import glob  # noqa: E402
import os  # noqa: E402

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "ref_mcica_*.npz"))), ids=os.path.basename)
def test_against_reference_fixture(hip, path, sweeps):
    """HIP generator and McICA solver against outputs of the REFERENCE's own Fortran (tools/gen_ref_fixtures.py):
    masks bit-exact, fluxes within the north-star bars.  The Mersenne-Twister fixtures were generated column by
    column (one freshly seeded stream each, as the reference's column driver does)."""
    f = np.load(path)
    ncol, nlay, icld, irng, seed = int(f["ncol"]), int(f["nlay"]), int(f["icld"]), int(f["irng"]), int(f["ims"]) * 140
    d = make_gcm_inputs(ncol, nlay, str(f["config"]), col0=int(f["col0"]))
    alpha = hip.get_alpha(ncol, nlay, icld, int(f["idcor"]), 2000.0, f["dz"], f["lat"], int(f["juldat"]), d["cldfr"])
    np.testing.assert_allclose(alpha, f["alpha"], rtol=4e-15, atol=0)
    alpha = np.asfortranarray(f["alpha"])        # continue from the reference's alpha so that the masks must agree exactly
    args = lambda sl: (d["play"][sl], d["cldfr"][sl], d["cicewp"][sl], d["cliqwp"][sl], d["reice"][sl], d["reliq"][sl],
                       d["taucld"][:, sl, :], alpha[sl])
    if irng == 0:
        sub = hip.mcica_subcol_lw(ncol, nlay, icld, seed, 0, *args(slice(None)))
    else:
        parts = [hip.mcica_subcol_lw(1, nlay, icld, seed, 1, *args(slice(c, c + 1))) for c in range(ncol)]
        sub = {k: np.asfortranarray(np.concatenate([p[k] for p in parts], axis=1 if parts[0][k].ndim == 3 else 0)) for k in SUB}
    mask = np.unpackbits(f["mask"])[:140 * ncol * nlay].reshape((140, ncol, nlay), order="F")
    assert np.array_equal(sub["cldfmcl"], mask.astype(float))
    for k3, ks in (("ciwpmcl", "ciwpsum"), ("clwpmcl", "clwpsum"), ("taucmcl", "taucsum")):
        assert np.array_equal(sub[k3].sum(axis=0), f[ks]), k3
    dd = dict(d)
    dd.update({k: sub[k] for k in SUB})
    got = hip.rrtmg_lw_mcica_from_dict(dd, icld=icld)
    ref = {k: f[k] for k in ("uflx", "dflx", "hr", "uflxc", "dflxc", "hrc", "duflx_dt", "duflxc_dt")}
    ref["icld"] = int(f["icld_out"])
    _compare(got, ref, d["idrv"], os.path.basename(path))
    if irng == 0:       # the fused entry generates the same sub-columns internally
        fused = hip.rrtmg_lw_mcica_subcol_from_dict(d, seed, 0, alpha=alpha, icld=icld)
        for k in ("uflx", "dflx", "hr", "uflxc", "dflxc", "hrc"):
            assert np.array_equal(fused[k], got[k]), k


def test_column_mode_mcica_samples(hip, oracle):
    """The column driver's imca = 1 loop (src/rrtmg_lw.1col.f90:471-660) for a reference example input: Mersenne-Twister
    sub-columns with seed ims * 140 per sample (bit-exact), cldprmc -> rtrnmc per sample on the prepared column."""
    from rrtmg_lw_amd.io_rrtm import read_input_rrtm
    col = read_input_rrtm(os.path.join(G, "input_rrtm_MLS-cld-imca1-icld2"), os.path.join(G, "in_cld_rrtm-cld5"), None)
    assert int(col["imca"]) == 1
    nl = int(col["nlayers"])
    samples = [1, 2, 3, 50, 200]
    got, subs = hip.column_mcica_samples(col, samples, irng=1)
    for k, ims in enumerate(samples):
        r2 = lambda v: np.asfortranarray(np.asarray(v, dtype=np.float64).reshape((1, nl)))
        osub = oracle.mcica_subcol(1, nl, int(col["icld"]), ims * 140, 1, r2(col["pavel"]), r2(col["cldfrac"]), r2(col["ciwp"]),
                                   r2(col["clwp"]), r2(col["rei"]), r2(col["rel"]),
                                   np.asfortranarray(np.asarray(col["tauc"]).reshape((16, 1, nl), order="F")), np.zeros((1, nl)))
        assert np.array_equal(subs[k]["cldfmc"], osub["cldfmcl"][:, 0, :])
        ref = oracle.column_mc(col, subs[k])
        for key, tol in (("totuflux", FLUX_TOL), ("totdflux", FLUX_TOL), ("totuclfl", FLUX_TOL), ("totdclfl", FLUX_TOL),
                         ("fnet", FLUX_TOL), ("htr", HR_TOL), ("htrc", HR_TOL)):
            d = np.abs(got[key][k] - ref[key]).max()
            assert d <= tol and d <= TIGHT_FLUX, (ims, key, d)
    assert np.ptp(got["totdflux"][:, 0]) > 1.0          # the samples differ: different sub-columns


def test_concurrent_fused_mcica_callers_are_combined(hip):
    """The fused sub-column generator + McICA entry from several threads at once, in chunks of a few columns (an OpenMP host model): calls
    with the kissvec generator that arrive while another is in flight are solved together in one device pass (driver.hip: comb_call, kind
    1) - every column draws from its own stream (src/mcica_subcol_gen_lw.f90:463-474), so a chunk's fluxes equal those of the same columns
    in one big call bit for bit; a chunk with a physics error fails alone with ITS text; calls with another permuteseed are not mixed in;
    the Mersenne Twister (one stream per call) keeps to the lock and gives what a lone call gives."""
    import threading
    ncol, nlay, chunk, icld = 768, 40, 32, 2
    d = make_gcm_inputs(ncol, nlay, "cloudy", col0=300)
    whole = hip.rrtmg_lw_mcica_subcol_from_dict(d, 13, 0, icld=icld)
    whole7 = hip.rrtmg_lw_mcica_subcol_from_dict(d, 7, 0, icld=icld)
    keys = [k for k, v in d.items() if isinstance(v, np.ndarray)]

    def part(c0, c1):
        p = dict(d)
        p["ncol"] = c1 - c0
        for k in keys:
            v = d[k]
            if v.ndim == 1:
                p[k] = np.ascontiguousarray(v[c0:c1])
            elif k == "taucld":
                p[k] = np.asfortranarray(v[:, c0:c1, :])
            else:
                p[k] = np.asfortranarray(v[c0:c1])
        return p

    chunks = [part(c0, min(ncol, c0 + chunk)) for c0 in range(0, ncol, chunk)]
    bad = 5
    r = np.array(chunks[bad]["reliq"]); r[3, 8] = 1.0
    chunks[bad]["reliq"] = np.asfortranarray(r)
    for k, v in (("cldfr", 0.5), ("cliqwp", 10.0)):
        a = np.array(chunks[bad][k]); a[3, 8] = v; chunks[bad][k] = np.asfortranarray(a)
    nthreads = 6
    results, errors = [None] * len(chunks), [None] * len(chunks)
    seeds = [7 if i % 5 == 0 else 13 for i in range(len(chunks))]

    def worker(t):
        for rep in range(2):
            for i in range(t, len(chunks), nthreads):
                try:
                    results[i] = hip.rrtmg_lw_mcica_subcol_from_dict(chunks[i], seeds[i], 0, icld=icld)
                except hip.RrtmgLwError as e:
                    errors[i] = str(e)

    th = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    for i, c0 in enumerate(range(0, ncol, chunk)):
        if i == bad:
            assert errors[i] and "LIQUID EFFECTIVE RADIUS OUT OF BOUNDS" in errors[i]
            continue
        assert errors[i] is None, (i, errors[i])
        ref = whole7 if seeds[i] == 7 else whole
        for k in ("uflx", "dflx", "hr", "uflxc", "dflxc", "hrc"):
            assert np.array_equal(results[i][k], ref[k][c0:c0 + chunk]), (i, k)
    # combined for certain: one large call holds the turn while ten small ones arrive (three attempts on a loaded machine)
    big = part(0, ncol)
    small = [c for i, c in enumerate(chunks[6:18])][:10]
    combined = False
    for attempt in range(3):
        c0_, p0_ = hip.combine_stats()
        started = threading.Event()

        def lead():
            started.set()
            hip.rrtmg_lw_mcica_subcol_from_dict(big, 13, 0, icld=icld)

        def follow(c):
            started.wait()
            hip.rrtmg_lw_mcica_subcol_from_dict(c, 13, 0, icld=icld)

        th = [threading.Thread(target=lead)] + [threading.Thread(target=follow, args=(c,)) for c in small]
        for x in th:
            x.start()
        for x in th:
            x.join()
        c1_, p1_ = hip.combine_stats()
        print(f"  fused McICA: {c1_ - c0_} calls in {p1_ - p0_} passes")
        if p1_ - p0_ < c1_ - c0_:
            combined = True
            break
    assert combined
    # the Mersenne Twister keeps to the lock: a chunk alone, whatever else is going on, gives what it gives alone
    mt1 = hip.rrtmg_lw_mcica_subcol_from_dict(chunks[2], 13, 1, icld=icld)
    mt2 = hip.rrtmg_lw_mcica_subcol_from_dict(chunks[2], 13, 1, icld=icld)
    for k in ("uflx", "dflx", "hr"):
        assert np.array_equal(mt1[k], mt2[k]), k
