"""The HIP path inside a process group (north star: columns shard over the GPUs of a node, ONE all-gather reassembles the
flux arrays): two ranks - two processes, each with its own library state, both on the one GPU a test box has - run the device-pointer
entries on their column blocks through the class bench.py uses (shard.ShardedStep), the packed blocks are gathered (gloo, host copies:
the box has one GPU and RCCL wants one per rank) and the result equals the single-process call over all columns bit for bit."""
import os
import socket

import numpy as np
import pytest

from rrtmg_lw_amd.synth import make_gcm_inputs

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, ncol, nlay, config, mcica, q):
    import torch
    import torch.distributed as dist
    from rrtmg_lw_amd import api
    from rrtmg_lw_amd.shard import ShardedStep, flux_names, output_rows
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)                       # every rank on the box's one GPU
    api.rrtmg_lw_ini(1004.0, kdata=api.STANDIN_KDATA, device=0)
    per = (ncol + world - 1) // world
    d = make_gcm_inputs(per, nlay, config, col0=rank * per, backend="torch", device=dev)     # (the last rank continues past ncol)
    idrv = d["idrv"]
    st = ShardedStep(nlay, idrv, per, world, device=dev, gather="host")
    assert st.rows == output_rows(nlay, idrv) and set(st.outs[0]) == set(flux_names(idrv)) | {"hr", "hrc"}
    stream = torch.cuda.current_stream().cuda_stream

    def solve(out):
        if mcica:
            api.rrtmg_lw_mcica_subcol_device(d, out, 11, 0, icld=mcica, stream=stream)
        else:
            api.rrtmg_lw_device(d, out, stream=stream)
        api.check(stream)

    api.set_batch(256)                                   # several internal batches per step
    for _ in range(3):                                   # both blocks reused, a gather waited for before its block is refilled
        last = st.step(solve)
    st.drain()
    if rank == 0:
        q.put({k: v.numpy() for k, v in st.result(last, ncol).items()})
    dist.barrier()
    api.finalize()
    dist.destroy_process_group()


@pytest.mark.parametrize("ncol,nlay,config,mcica", [(700, 40, "cloudy", 0), (1001, 72, "aer_idrv", 0), (777, 40, "cloudy", 2)])
def test_two_ranks_on_one_gpu_equal_one_process(hip, ncol, nlay, config, mcica):
    """Even split; a ragged column count with aerosol and idrv = 1 (the block then carries d(flux)/dT); the fused sub-column generator +
    McICA solver (kissvec: a stream per column, so a rank's block draws what the whole call draws for those columns)."""
    import torch
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, ncol, nlay, config, mcica, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # the same columns in ONE call of this process (host-pointer entry: another route to the same kernels)
    d = make_gcm_inputs(ncol, nlay, config, col0=0)
    idrv = d["idrv"]
    one = hip.rrtmg_lw_mcica_subcol_from_dict(d, 11, 0, icld=mcica) if mcica else hip.rrtmg_lw_from_dict(d)
    keys = ("uflx", "dflx", "uflxc", "dflxc", "hr", "hrc") + (("duflx_dt", "duflxc_dt") if idrv else ())
    assert set(got) == set(keys)
    for k in keys:
        assert got[k].shape == one[k].shape, k
        assert np.array_equal(got[k], one[k]), k
    assert np.abs(one["uflx"]).max() > 100.0
