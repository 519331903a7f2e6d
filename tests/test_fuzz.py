"""Seeded random sweep over shapes, modes and band ranges - the launch shapes of the group workgroups (k_sweepc / k_sweepz) depend on the
number of bands in range, on idrv and on the mode, the batch size on the column count: every combination below goes through the C ABI and
is compared with the oracle at the regression bar."""
import os

import numpy as np
import pytest

from rrtmg_lw_amd.io_rrtm import read_input_rrtm
from rrtmg_lw_amd.synth import make_gcm_inputs

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _cases(seed, n):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        config = ("clear", "cloudy", "aer_idrv")[rng.integers(0, 3)]
        out.append(dict(config=config, nlay=int(rng.integers(2, 141)), ncol=int(rng.integers(1, 700)),
                        icld=int(rng.integers(0, 4)) if config != "clear" else int(rng.integers(0, 2)),
                        idrv=int(rng.integers(0, 2)), batch=int((64, 128, 256, 4096)[rng.integers(0, 4)]), col0=int(rng.integers(0, 10 ** 6))))
    return out


@pytest.mark.parametrize("c", _cases(20260104, 28), ids=lambda c: "%(config)s-L%(nlay)d-n%(ncol)d-icld%(icld)d-idrv%(idrv)d-b%(batch)d" % c)
def test_random_gcm_calls(hip, oracle, c, sweeps):
    d = make_gcm_inputs(c["ncol"], c["nlay"], c["config"], col0=c["col0"])
    hip.set_batch(c["batch"])
    try:
        got = hip.rrtmg_lw_from_dict(d, icld=c["icld"], idrv=c["idrv"])
    finally:
        hip.set_batch(0)
    ref = oracle.rrtmg_lw(c["ncol"], c["nlay"], c["icld"], c["idrv"], d)
    keys = ("uflx", "dflx", "uflxc", "dflxc") + (("duflx_dt", "duflxc_dt") if c["idrv"] else ())
    dflux = max(np.abs(got[k] - ref[k]).max() for k in keys)
    scale = max(np.abs(ref[k]).max() for k in ("uflx", "dflx"))
    assert dflux <= max(5e-5, 2.5e-7 * scale), dflux
    # heating rates: the bar relative to the layer's own rate where layers get very thin (a 140-layer profile reaches 1e-3 hPa layers)
    for k in ("hr", "hrc"):
        assert (np.abs(got[k] - ref[k]) <= 5e-5 + 1e-6 * np.abs(ref[k])).all(), k
    assert got["icld"] == ref["icld"]


def _terrain_cases(seed, n):
    rng = np.random.default_rng(seed)
    return [dict(nlay=int(rng.integers(8, 141)), ncol=int(rng.integers(1, 1200)), icld=int(rng.integers(0, 4)), idrv=int(rng.integers(0, 2)),
                 batch=int((256, 512, 4096)[rng.integers(0, 3)]), col0=int(rng.integers(0, 10 ** 6)), spread=float(rng.uniform(0.3, 1.0)))
            for _ in range(n)]


@pytest.mark.parametrize("c", _terrain_cases(20260106, 14), ids=lambda c: "L%(nlay)d-n%(ncol)d-icld%(icld)d-idrv%(idrv)d-b%(batch)d-f%(spread).2f" % c)
def test_random_terrain_calls(hip, oracle, c):
    """Terrain-following pressure grids (synth "cloudy_orography", the mountain ranges stretched further by a random factor: surface pressures
    down to 0.3 x 1013 hPa): k_layer's narrow and wide staging windows, the global-memory evaluation beyond both, layers that hold
    tropospheric and stratospheric cells side by side - random layer counts, column counts that end inside a window, batch sizes, cloud modes."""
    d = make_gcm_inputs(c["ncol"], c["nlay"], "cloudy_orography", col0=c["col0"])
    f = np.array(d["plev"])[:, 0] / 1013.0
    g = np.where(f < 0.96, 1.0 - (1.0 - f) * (1.0 - c["spread"]) / 0.45, 1.0)          # (the deepest range, factor 0.55, goes to `spread`)
    g = np.clip(g / f, 0.2, 1.0) if c["spread"] < 0.55 else np.ones_like(f)
    for k in ("play", "plev"):
        d[k] = np.asfortranarray(np.array(d[k]) * g[:, None])
    hip.set_batch(c["batch"])
    try:
        got = hip.rrtmg_lw_from_dict(d, icld=c["icld"], idrv=c["idrv"])
    finally:
        hip.set_batch(0)
    ref = oracle.rrtmg_lw(c["ncol"], c["nlay"], c["icld"], c["idrv"], d)
    keys = ("uflx", "dflx", "uflxc", "dflxc") + (("duflx_dt", "duflxc_dt") if c["idrv"] else ())
    dflux = max(np.abs(got[k] - ref[k]).max() for k in keys)
    scale = max(np.abs(ref[k]).max() for k in ("uflx", "dflx"))
    assert dflux <= max(5e-5, 2.5e-7 * scale), dflux
    # heating rates: a rate is the layer's flux divergence x 8.44 / dp[hPa]; the stretched grids reach layers of a few Pa, so the bars are
    # those of the thin-layer tests (test_hip_parity._compare_thin_layers): every layer's flux divergence within 2.5e-5 W m-2 - i.e. a rate
    # within max(5e-5 K d-1, 2.1e-4 / dp) -, relative 1e-6 on top, and the north-star 1e-3 K d-1 in every layer at least 0.25 hPa thick
    dp = np.array(d["plev"])[:, :-1] - np.array(d["plev"])[:, 1:]
    for k in ("hr", "hrc"):
        err = np.abs(got[k] - ref[k])
        assert (err <= np.maximum(5e-5, 2.5e-5 * 8.4391 / dp) + 1e-6 * np.abs(ref[k])).all(), (k, float(err.max()))
        assert err[dp >= 0.25].max() <= 1e-3, k
    assert got["icld"] == ref["icld"]


def test_random_ragged_windows_wide_against_narrow(hip):
    """Sixty random (first column, width <= 700, cloud mode) calls on the terrain-following field, each with k_layer's wide staging window and
    without: bit for bit the same.  (Round 5: twelve of sixty such calls differed - whole waves of the ragged last window - before the
    libraries were compiled with -mllvm -amdgpu-remove-redundant-endcf=0: profiles/round5_exec_hazard.md.)"""
    rng = np.random.default_rng(5)
    bad = []
    for trial in range(60):
        col0, ncol, icld = int(rng.integers(0, 10 ** 6)), int(rng.integers(65, 700)), int(rng.integers(1, 4))
        d = make_gcm_inputs(ncol, 72, "cloudy_orography", col0=col0)
        outs = {}
        for on in (1, 0):
            prev = hip.set_wide_window(on)
            try:
                outs[on] = hip.rrtmg_lw_from_dict(d, icld=icld, idrv=0)
            finally:
                hip.set_wide_window(prev)
        if any(not np.array_equal(outs[1][k], outs[0][k]) for k in outs[1]):
            bad.append((col0, ncol, icld))
    assert not bad, bad


@pytest.mark.parametrize("seed", range(10))
def test_random_band_ranges(hip, oracle, seed):
    """Prepared-column entry with a random band range, cloud file and idrv: the groups hold only the bands in range."""
    rng = np.random.default_rng(100 + seed)
    # (what the reference's sweeps accept: one band with iout > 0, or a range that starts at band 1 - with iout = 0 its g-point counter
    # starts at 1 whatever istart is, src/rrtmg_lw_rtrn.f90:354-360)
    if rng.integers(0, 2):
        a = b = int(rng.integers(1, 17))
    else:
        a, b = 1, int(rng.integers(1, 17))
    cloudy = bool(rng.integers(0, 2))
    col = read_input_rrtm(os.path.join(G, "input_rrtm_MLS-cld-imca0-icld2" if cloudy else ("input_rrtm_MLS-clr-idrv1", "input_rrtm_TROP-clr")[rng.integers(0, 2)]),
                          os.path.join(G, ("in_cld_rrtm-cld5", "in_cld_rrtm-cld7")[rng.integers(0, 2)]) if cloudy else None)
    icld = int(rng.integers(1, 3)) if cloudy else int(col["icld"])
    got = hip.run_columns([col] * int(rng.integers(1, 70)), a, b, icld=icld)
    ref = oracle.column(col, a, b, 99 if a == b else 0, icld=icld)
    for k in ("totuflux", "totdflux", "totuclfl", "totdclfl", "htr", "htrc") + (("dtotuflux_dt",) if int(col["idrv"]) == 1 else ()):
        assert np.abs(got[k] - ref[k][None, :]).max() <= 5e-5, (a, b, k)


def _mc_cases(seed, n, configs=("cloudy", "aer_idrv"), ncol_max=300):
    rng = np.random.default_rng(seed)
    return [dict(nlay=int(rng.integers(4, 120)), ncol=int(rng.integers(1, ncol_max)), icld=int(rng.integers(1, 6)), idrv=int(rng.integers(0, 2)),
                 seed=int(rng.integers(0, 5000)), batch=int((64, 256, 4096)[rng.integers(0, 3)]), col0=int(rng.integers(0, 10 ** 6)),
                 config=configs[rng.integers(0, len(configs))]) for _ in range(n)]


# (the second list: terrain-following grids - k_layer<mcmask>'s wide staging window, ragged last windows)
@pytest.mark.parametrize("c", _mc_cases(20260105, 12) + _mc_cases(20260107, 8, ("cloudy_orography",), 700), ids=lambda c: "%(config)s-L%(nlay)d-n%(ncol)d-icld%(icld)d-idrv%(idrv)d-seed%(seed)d-b%(batch)d" % c)
def test_random_mcica_calls(hip, oracle, c, sweeps):
    """The fused entry (kissvec generator by jump-ahead -> cldprmc -> rtrnmc) against the oracle's generator + McICA solver: random
    overlap rule, seed (= how far the jump tables reach), layer count (= draws per sub-column), batch size."""
    ncol, nlay = c["ncol"], c["nlay"]
    d = make_gcm_inputs(ncol, nlay, c["config"], col0=c["col0"])
    d["idrv"] = c["idrv"]
    rng = np.random.default_rng(c["seed"])
    alpha = np.asfortranarray(rng.random((ncol, nlay)))
    hip.set_batch(c["batch"])
    try:
        got = hip.rrtmg_lw_mcica_subcol_from_dict(d, c["seed"], 0, alpha=alpha, icld=c["icld"])
    finally:
        hip.set_batch(0)
    sc = oracle.mcica_subcol(ncol, nlay, c["icld"], c["seed"], 0, d["play"], d["cldfr"], d["cicewp"], d["cliqwp"], d["reice"], d["reliq"],
                             d["taucld"], alpha)
    dd = dict(d)
    dd.update({k: sc[k] for k in ("cldfmcl", "ciwpmcl", "clwpmcl", "reicmcl", "relqmcl", "taucmcl")})
    ref = oracle.rrtmg_lw(ncol, nlay, c["icld"], c["idrv"], dd, mcica=True)
    keys = ("uflx", "dflx", "uflxc", "dflxc") + (("duflx_dt", "duflxc_dt") if c["idrv"] else ())
    dflux = max(np.abs(got[k] - ref[k]).max() for k in keys)
    scale = max(np.abs(ref[k]).max() for k in ("uflx", "dflx"))
    assert dflux <= max(5e-5, 2.5e-7 * scale), dflux
    # (a rate is the layer's flux divergence x 8.44 / dp[hPa]: the bar of the thin-layer tests - 2.5e-5 W m-2 of divergence - where a random
    # layer count puts layers of a tenth of a hectopascal under a cloud; 5e-5 K d-1 elsewhere)
    dp = np.array(d["plev"])[:, :-1] - np.array(d["plev"])[:, 1:]
    for k in ("hr", "hrc"):
        assert (np.abs(got[k] - ref[k]) <= np.maximum(5e-5, 2.5e-5 * 8.4391 / dp) + 1e-6 * np.abs(ref[k])).all(), k
