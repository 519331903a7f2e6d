"""Synthetic-input generator (numpy and torch backends agree, shards are reproducible) and the multi-GPU plumbing:
column blocks + the packed output block reassembled by ONE all-gather, exercised with world_size 2 on gloo (CPU).
The per-rank "compute" in the gloo test is the CPU oracle - the point is the sharding/packing/collective logic that
bench.py uses over RCCL, not the kernels."""
import os
import socket

import numpy as np
import pytest

from rrtmg_lw_amd.shard import column_block, flux_names, output_rows, output_views, unpack_gathered
from rrtmg_lw_amd.synth import base_profile, make_gcm_inputs


def test_base_profile_regrid():
    p = base_profile(72)
    assert p["plev"].shape == (73,) and abs(p["plev"][0] - 1013.0) < 1e-9 and abs(p["plev"][-1] - 0.067) < 1e-9
    assert np.all(np.diff(p["plev"]) < 0) and np.all((p["play"] < p["plev"][:-1]) & (p["play"] > p["plev"][1:]))
    assert 150 < p["tlay"].min() and p["tlay"].max() < 320 and p["vmr"].shape == (7, 72)


@pytest.mark.parametrize("config", ["clear", "cloudy", "aer_idrv"])
def test_numpy_and_torch_backends_agree(config):
    import torch
    a = make_gcm_inputs(37, 72, config, col0=123456)
    b = make_gcm_inputs(37, 72, config, col0=123456, backend="torch", device=torch.device("cpu"))
    for k, v in a.items():
        if isinstance(v, np.ndarray):
            t = b[k]
            assert tuple(t.shape) == v.shape
            assert np.array_equal(t.numpy(), v), k
            # column index fastest in memory
            assert t.stride()[0 if k != "taucld" else 1] in (1, 16), k
    assert (a["icld"], a["idrv"]) == (b["icld"], b["idrv"])


def test_orography_variant():
    """"cloudy_orography": the same clouds as "cloudy" on a terrain-following pressure grid - 30 % of the columns in mountain ranges (runs
    of 8-64 columns, surface-pressure factor 0.55-0.95), temperatures read at the column's own pressures.  The reference-pressure index jp
    of setcoef (src/rrtmg_lw_setcoef.f90:276-284) then spans three values and more inside 256 consecutive columns of a layer; the torch
    backend agrees with numpy up to the last bits of log / interpolation; shards are reproducible."""
    import torch
    a = make_gcm_inputs(1024, 72, "cloudy_orography", col0=5000)
    b = make_gcm_inputs(1024, 72, "cloudy_orography", col0=5000, backend="torch", device=torch.device("cpu"))
    for k, v in a.items():
        if isinstance(v, np.ndarray):
            assert np.allclose(b[k].numpy(), v, rtol=1e-12, atol=1e-12), k
    c = make_gcm_inputs(1024, 72, "cloudy", col0=5000)
    assert np.array_equal(a["cldfr"], c["cldfr"]) and np.array_equal(a["cliqwp"], c["cliqwp"])
    psf = a["plev"][:, 0] / 1013.0
    assert 0.2 < (psf < 0.96).mean() < 0.4 and psf.min() < 0.6 and psf.max() > 1.02
    assert np.all(np.diff(a["plev"], axis=1) < 0) and np.all(a["tlay"] > 150) and np.all(a["tlay"] < 330)
    jp = np.floor(36.0 - 5.0 * (np.log(a["play"]) + 0.04)).astype(int).reshape(4, 256, 72)
    assert ((jp.max(axis=1) - jp.min(axis=1)) >= 2).mean() > 0.5
    part = make_gcm_inputs(100, 72, "cloudy_orography", col0=5300)
    for k in ("play", "plev", "tlay", "tlev", "tsfc", "cldfr"):
        assert np.array_equal(a[k][300:400], part[k]), k


def test_shards_are_reproducible():
    full = make_gcm_inputs(100, 40, "cloudy", col0=0)
    part = make_gcm_inputs(30, 40, "cloudy", col0=50)
    for k in ("tlay", "h2ovmr", "cldfr", "cliqwp", "tsfc", "emis"):
        assert np.array_equal(full[k][50:80], part[k]), k
    assert np.array_equal(full["taucld"][:, 50:80], part["taucld"])
    assert (full["cldfr"].sum(axis=1) == 0).mean() > 0.15          # ~30 % of the columns are cloud-free


def test_column_blocks_cover_everything():
    for ncol, world in ((1000000, 8), (10, 4), (7, 8), (1, 1)):
        seen = []
        for r in range(world):
            c0, n, per = column_block(ncol, world, r)
            seen += list(range(c0, c0 + n))
            assert n <= per
        assert seen == list(range(ncol))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, ncol, nlay, config, q):
    """One rank of the sharded step, through the class bench.py uses (shard.ShardedStep): `per` = ceil(ncol / world) columns on every
    rank (the last rank continues past ncol), the packed block sized by idrv, one asynchronous all_gather_into_tensor per step.  Three
    steps, so that both blocks are reused and a gather is waited for before its block is refilled; the oracle stands in for the device entry."""
    import torch
    import torch.distributed as dist
    from oracle.bindings import Oracle
    from rrtmg_lw_amd.shard import ShardedStep
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per = (ncol + world - 1) // world
    d = make_gcm_inputs(per, nlay, config, col0=rank * per)
    idrv = d["idrv"]
    o = Oracle().rrtmg_lw(per, nlay, d["icld"], idrv, d)
    st = ShardedStep(nlay, idrv, per, world, device=None, gather=True)
    assert st.rows == output_rows(nlay, idrv) and set(st.outs[0]) == set(flux_names(idrv)) | {"hr", "hrc"}
    calls = []

    def solve(out):
        calls.append(1)
        scale = 1.0 if len(calls) == 3 else 0.5                # only the last step carries the true values
        for k in out:
            out[k][:, :] = torch.from_numpy(np.ascontiguousarray(o[k].T)) * scale

    for _ in range(3):
        last = st.step(solve)
    st.drain()
    if rank == 0:
        res = st.result(last, ncol)
        q.put({k: v.numpy() for k, v in res.items()})
    dist.destroy_process_group()


@pytest.mark.parametrize("ncol,nlay,config", [(22, 30, "cloudy"), (23, 30, "cloudy"), (21, 20, "aer_idrv")])
def test_two_rank_gloo_allgather_matches_single_process(oracle, ncol, nlay, config):
    """Even split, a column count the world size does not divide, and idrv = 1 (the block then carries duflx_dt / duflxc_dt)."""
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, ncol, nlay, config, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    d = make_gcm_inputs(ncol, nlay, config, col0=0)
    ref = oracle.rrtmg_lw(ncol, nlay, d["icld"], d["idrv"], d)
    keys = ("uflx", "dflx", "uflxc", "dflxc", "hr", "hrc") + (("duflx_dt", "duflxc_dt") if d["idrv"] else ())
    assert set(got) == set(keys)
    for k in keys:
        assert got[k].shape == ref[k].shape
        assert np.array_equal(got[k], ref[k]), k


def test_packed_block_drops_derivative_rows_without_idrv():
    assert output_rows(72, 0) == 4 * 73 + 2 * 72 == 436          # 436 rows per column at 72 layers
    assert output_rows(72, 1) == 6 * 73 + 2 * 72 == 582
