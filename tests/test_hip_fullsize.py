"""Full-size checks on the GPU (BASELINE.json configs 2 and 3) through properties that do not need the oracle to run a million
columns: replicated columns give bitwise identical results, every column of a 1e6-column device-resident call equals the same
column computed in a small stand-alone call (batching, stream pipelining and column offsets are transparent), heating rates are
consistent with the net fluxes, physical bounds hold; the oracle checks a sample."""
import numpy as np
import pytest

from rrtmg_lw_amd.synth import make_gcm_inputs

pytestmark = pytest.mark.gpu


def _device_run(hip, torch, d, nlay, ncol):
    from rrtmg_lw_amd.shard import output_rows, output_views
    dev = torch.device("cuda", 0)
    outbuf = torch.full((output_rows(nlay), ncol), float("nan"), dtype=torch.float64, device=dev)
    out = output_views(outbuf, nlay)
    s = torch.cuda.current_stream().cuda_stream
    hip.rrtmg_lw_device(d, out, stream=s)
    hip.check(s)
    return out


def test_config2_replicated_columns(hip, oracle):
    """10 000 identical 72-layer clear-sky columns (BASELINE configs[1]): every column bitwise equal, and equal to the oracle."""
    one = make_gcm_inputs(1, 72, "clear", col0=0)
    ncol = 10_000
    d = dict(one)
    d["ncol"] = ncol
    for k, v in one.items():
        if isinstance(v, np.ndarray) and v.ndim >= 1:
            cax = 1 if k == "taucld" else 0
            d[k] = np.asfortranarray(np.repeat(v, ncol, axis=cax))
    for icld in (0, 1):            # icld = 1 routes the cloud-free columns through rtrn
        got = hip.rrtmg_lw_from_dict(d, icld=icld)
        ref = oracle.rrtmg_lw(1, 72, icld, one["idrv"], one)
        for k in ("uflx", "dflx", "hr", "uflxc", "dflxc", "hrc"):
            assert (got[k] == got[k][:1]).all(), k
            assert np.abs(got[k][0] - ref[k][0]).max() <= (1e-3 if k.startswith("hr") else 1e-2)
            assert np.abs(got[k][0] - ref[k][0]).max() <= 5e-5


@pytest.mark.parametrize("config", ["cloudy", "cloudy_orography"])
def test_config3_million_columns(hip, oracle, config):
    """1e6 synthetic 72-layer cloudy columns, maximum-random overlap, device resident (BASELINE configs[2]); and the same on a
    terrain-following pressure grid (k_layer's second launch with the wide staging window takes nine workgroups in ten there)."""
    import torch
    ncol, nlay = 1_000_000, 72
    dev = torch.device("cuda", 0)
    slab = 131072
    parts = [make_gcm_inputs(min(slab, ncol - s), nlay, config, col0=s, backend="torch", device=dev) for s in range(0, ncol, slab)]
    d = dict(parts[0])
    d["ncol"] = ncol
    for k, v in parts[0].items():
        if torch.is_tensor(v):
            cat = torch.cat([p[k] for p in parts], dim=1 if k == "taucld" else 0)
            nd = cat.dim()
            d[k] = cat.permute(*reversed(range(nd))).contiguous().permute(*reversed(range(nd))) if nd > 1 else cat.contiguous()
    del parts
    out = _device_run(hip, torch, d, nlay, ncol)
    # every output written, physical bounds
    for k in ("uflx", "dflx", "uflxc", "dflxc", "hr", "hrc"):
        assert torch.isfinite(out[k]).all(), k
    assert (out["uflx"] > 0).all() and (out["dflx"] >= 0).all()
    assert (out["dflx"][nlay] == 0).all() and (out["dflxc"][nlay] == 0).all()          # no downward flux at the top level
    assert (out["uflx"][0] > 100).all() and (out["uflx"][0] < 700).all()               # surface emission of 250..330 K surfaces
    # heating rate = heatfac * d(fnet)/dp exactly as rtrn forms it (src/rrtmg_lw_rtrn.f90:597-604)
    fnet = out["uflx"] - out["dflx"]
    plev = d["plev"].t()                                                                # (nlay+1, ncol)
    heatfac = 9.8066 * 8.64e4 / (1004.0 * 1.e2)
    hr = heatfac * (fnet[:-1] - fnet[1:]) / (plev[:-1] - plev[1:])
    assert torch.allclose(hr, out["hr"], rtol=1e-12, atol=1e-12)
    # clouds matter, and only where there are clouds: cloud-free columns have identical total and clear-sky fluxes
    cloudfree = (d["cldfr"].t() == 0).all(dim=0)
    assert 0.2 < cloudfree.double().mean().item() < 0.4
    assert torch.equal(out["dflx"][:, cloudfree], out["dflxc"][:, cloudfree])
    assert (out["dflx"][0, ~cloudfree] - out["dflxc"][0, ~cloudfree]).abs().max().item() > 10.0
    # any window of columns equals the same columns computed on their own (offsets 0, mid-batch, batch boundary, tail)
    for c0 in (0, 40_000, 65_536 - 100, 999_000):
        n = 300
        dn = make_gcm_inputs(n, nlay, config, col0=c0)
        if config == "cloudy_orography":       # (the torch backend's log / interpolation differ from numpy's in the last bits: the call's own columns)
            for k, v in d.items():
                if torch.is_tensor(v):
                    dn[k] = np.asfortranarray((v[:, c0:c0 + n] if k == "taucld" else v[c0:c0 + n]).cpu().numpy())
        alone = hip.rrtmg_lw_from_dict(dn)
        for k in ("uflx", "dflx", "hr", "uflxc", "dflxc", "hrc"):
            assert np.array_equal(out[k][:, c0:c0 + n].t().cpu().numpy(), alone[k]), (k, c0)
        if c0 == 40_000:
            ref = oracle.rrtmg_lw(n, nlay, dn["icld"], dn["idrv"], dn)
            for k in ("uflx", "dflx", "uflxc", "dflxc"):
                assert np.abs(alone[k] - ref[k]).max() <= 5e-5
            for k in ("hr", "hrc"):
                assert np.abs(alone[k] - ref[k]).max() <= 5e-5


def test_config4_mcica_half_million_columns(hip, oracle):
    """5e5 synthetic 72-layer columns, McICA with the kissvec generator and exponential-random overlap through the fused device
    entry (BASELINE configs[3], half the columns): per-column streams make every window independent of the rest of the call."""
    import torch
    from rrtmg_lw_amd.shard import output_rows, output_views
    ncol, nlay = 500_000, 72
    dev = torch.device("cuda", 0)
    slab = 125_000
    parts = [make_gcm_inputs(slab, nlay, "cloudy", col0=s, backend="torch", device=dev) for s in range(0, ncol, slab)]
    d = dict(parts[0])
    d["ncol"] = ncol
    for k, v in parts[0].items():
        if torch.is_tensor(v):
            cat = torch.cat([p[k] for p in parts], dim=1 if k == "taucld" else 0)
            nd = cat.dim()
            d[k] = cat.permute(*reversed(range(nd))).contiguous().permute(*reversed(range(nd))) if nd > 1 else cat.contiguous()
    del parts
    alpha = torch.full((nlay, ncol), 0.55, dtype=torch.float64, device=dev).t()
    outbuf = torch.full((output_rows(nlay), ncol), float("nan"), dtype=torch.float64, device=dev)
    out = output_views(outbuf, nlay)
    s = torch.cuda.current_stream().cuda_stream
    hip.rrtmg_lw_mcica_subcol_device(d, out, 1, 0, alpha=alpha, icld=5, stream=s)
    hip.check(s)
    for k in ("uflx", "dflx", "uflxc", "dflxc", "hr", "hrc"):
        assert torch.isfinite(out[k]).all(), k
    assert (out["dflx"][nlay] == 0).all()
    al = np.full((300, nlay), 0.55)
    for c0 in (0, 65_536 - 150, 499_700):
        dn = make_gcm_inputs(300, nlay, "cloudy", col0=c0)
        alone = hip.rrtmg_lw_mcica_subcol_from_dict(dn, 1, 0, alpha=al, icld=5)
        for k in ("uflx", "dflx", "hr", "uflxc", "dflxc", "hrc"):
            assert np.array_equal(out[k][:, c0:c0 + 300].t().cpu().numpy(), alone[k]), (k, c0)
    # the oracle on the last window: same masks (bit-exact generator), fluxes within the tight bar
    sub = oracle.mcica_subcol(300, nlay, 5, 1, 0, dn["play"], dn["cldfr"], dn["cicewp"], dn["cliqwp"], dn["reice"], dn["reliq"], dn["taucld"], al)
    dd = dict(dn)
    dd.update({k: sub[k] for k in ("cldfmcl", "ciwpmcl", "clwpmcl", "reicmcl", "relqmcl", "taucmcl")})
    ref = oracle.rrtmg_lw(300, nlay, 5, dn["idrv"], dd, mcica=True)
    for k in ("uflx", "dflx", "uflxc", "dflxc", "hr", "hrc"):
        assert np.abs(alone[k] - ref[k]).max() <= 5e-5, k


def test_config5_aerosol_derivative_137_layers(hip, oracle):
    """3e5 columns of 137 layers with aerosol optical depth and idrv = 1 (BASELINE configs[4], a third of the columns)."""
    import torch
    ncol, nlay = 300_000, 137
    dev = torch.device("cuda", 0)
    parts = [make_gcm_inputs(100_000, nlay, "aer_idrv", col0=s, backend="torch", device=dev) for s in range(0, ncol, 100_000)]
    d = dict(parts[0])
    d["ncol"] = ncol
    for k, v in parts[0].items():
        if torch.is_tensor(v):
            cat = torch.cat([p[k] for p in parts], dim=1 if k == "taucld" else 0)
            nd = cat.dim()
            d[k] = cat.permute(*reversed(range(nd))).contiguous().permute(*reversed(range(nd))) if nd > 1 else cat.contiguous()
    del parts
    assert d["idrv"] == 1
    out = _device_run(hip, torch, d, nlay, ncol)
    for k in ("uflx", "dflx", "uflxc", "dflxc", "hr", "hrc", "duflx_dt", "duflxc_dt"):
        assert torch.isfinite(out[k]).all(), k
    # dF/dT: positive, largest at the surface, decreasing upward (it is attenuated by every layer: rtrn :507-517)
    assert (out["duflx_dt"][0] > 0).all()
    assert (out["duflx_dt"][:-1] >= out["duflx_dt"][1:] - 1e-12).all()
    for c0 in (0, 199_900):
        dn = make_gcm_inputs(200, nlay, "aer_idrv", col0=c0)
        alone = hip.rrtmg_lw_from_dict(dn)
        for k in ("uflx", "dflx", "hr", "uflxc", "dflxc", "hrc", "duflx_dt", "duflxc_dt"):
            assert np.array_equal(out[k][:, c0:c0 + 200].t().cpu().numpy(), alone[k]), (k, c0)
    ref = oracle.rrtmg_lw(200, nlay, dn["icld"], 1, dn)
    for k in ("uflx", "dflx", "uflxc", "dflxc", "hr", "hrc", "duflx_dt", "duflxc_dt"):
        assert np.abs(alone[k] - ref[k]).max() <= 5e-5, k
