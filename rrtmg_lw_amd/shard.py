"""Column sharding across the GPUs of one node and the packed output block that one RCCL all-gather reassembles.

Columns are independent (reference: the `do iplon = 1, ncol` loop, src/rrtmg_lw_rad.nomcica.f90:472), so rank r of N
owns the contiguous block [r*per, (r+1)*per).  Every output of rrtmg_lw is an (ncol, nlay[+1]) array with the column
index fastest, i.e. nlay[+1] rows of ncol values; the rows of all outputs are stacked into one (rows, ncol_local)
buffer so that a single all_gather_into_tensor moves everything.
"""
from __future__ import annotations

FLUX_NAMES = ("uflx", "dflx", "uflxc", "dflxc")
DERIV_NAMES = ("duflx_dt", "duflxc_dt")          # part of the block only when idrv = 1 (a third of the rows otherwise dead weight in the gather)
RATE_NAMES = ("hr", "hrc")


def flux_names(idrv=1):
    return FLUX_NAMES + (DERIV_NAMES if idrv else ())


def column_block(ncol, world, rank):
    """(first column, number of columns, columns per rank) of `rank`; the last ranks may be short or empty."""
    per = (ncol + world - 1) // world
    col0 = min(rank * per, ncol)
    return col0, max(0, min(per, ncol - col0)), per


def output_rows(nlay, idrv=1):
    """Rows of the packed block: 4 (idrv = 0) or 6 (idrv = 1) flux arrays of nlay + 1 levels, 2 heating-rate arrays of nlay layers."""
    return len(flux_names(idrv)) * (nlay + 1) + len(RATE_NAMES) * nlay


def output_views(buf, nlay, idrv=1):
    """Slice a (rows, ncol) buffer (torch tensor or numpy array) into the named output arrays (six without dF/dT, eight with)."""
    out, r = {}, 0
    for nm in flux_names(idrv):
        out[nm] = buf[r:r + nlay + 1]
        r += nlay + 1
    for nm in RATE_NAMES:
        out[nm] = buf[r:r + nlay]
        r += nlay
    return out


def unpack_gathered(gathered, nlay, ncol, idrv=1):
    """gathered: (world, rows, per) -> dict of (ncol, nlay[+1]) arrays in global column order (columns past ncol - the padding of
    the last ranks when ncol is not a multiple of the world size - are dropped)."""
    world, rows, per = gathered.shape
    assert rows == output_rows(nlay, idrv)
    out = {}
    views = [output_views(gathered[r], nlay, idrv) for r in range(world)]
    for nm in flux_names(idrv) + RATE_NAMES:
        parts = [v[nm].T for v in views]                  # (per, nlev) each
        if hasattr(parts[0], "numpy") and not hasattr(parts[0], "__array_interface__"):
            import torch
            out[nm] = torch.cat(parts, dim=0)[:ncol]
        else:
            import numpy as np
            out[nm] = np.concatenate(parts, axis=0)[:ncol]
    return out


class ShardedStep:
    """One rank's side of the sharded step (bench.py, and the two-rank gloo test runs the same code): the rank's packed output block,
    filled by `solve`, and the single all-gather that reassembles the outputs of all ranks (north star).  Two blocks alternate: the
    gather of step k - asynchronous, on the collective's own stream - overlaps the kernels of step k + 1, and a block is handed to
    `solve` again only after the gather that last read it has completed.

        st = ShardedStep(nlay, idrv, per, world, device)          # per = ceil(ncol / world) columns on EVERY rank
        k = st.step(lambda out: rrtmg_lw_device(d, out, ...))     # out: dict of views into block k (output_views)
        st.drain(); res = st.result(k, ncol)                      # dict of (ncol, nlay[+1]) tensors in global column order
    """

    def __init__(self, nlay, idrv, per, world, device=None, gather=True):
        """gather: True - the collective runs on the blocks where they are (RCCL on device blocks, gloo on host blocks); "host" - the
        blocks live on `device`, a host copy of each is gathered (a process group without a device backend: the two-rank test on one GPU);
        False - no collective (one rank, or the caller gathers)."""
        import torch
        self.nlay, self.idrv, self.per, self.world = nlay, idrv, per, world
        self.rows = output_rows(nlay, idrv)
        self.outbufs = [torch.zeros((self.rows, per), dtype=torch.float64, device=device) for _ in range(2)]
        self.outs = [output_views(b, nlay, idrv) for b in self.outbufs]
        self.do_gather = bool(gather) and world >= 1
        self.via_host = gather == "host"
        gdev = None if self.via_host else device
        self.hostbufs = [torch.empty((self.rows, per), dtype=torch.float64) for _ in range(2)] if self.via_host else None
        self.gathered = [torch.empty((world * self.rows, per), dtype=torch.float64, device=gdev) if self.do_gather else None for _ in range(2)]
        self.pending = [None, None]
        self.count = 0

    def step(self, solve):
        import torch.distributed as dist
        k = self.count & 1
        self.count += 1
        if self.pending[k] is not None:          # the gather that last read this block must be done before it is overwritten
            self.pending[k].wait()
            self.pending[k] = None
        solve(self.outs[k])
        if self.do_gather:
            src = self.outbufs[k]
            if self.via_host:                    # (a blocking copy on the current stream: `solve` enqueued its kernels there)
                self.hostbufs[k].copy_(src)
                src = self.hostbufs[k]
            self.pending[k] = dist.all_gather_into_tensor(self.gathered[k], src, async_op=True)
        return k

    def drain(self):
        for k in range(2):
            if self.pending[k] is not None:
                self.pending[k].wait()
                self.pending[k] = None

    def result(self, k, ncol):
        """outputs of all ranks after the gather of block k (drain first): global column order, padding columns dropped"""
        return unpack_gathered(self.gathered[k].view(self.world, self.rows, self.per), self.nlay, ncol, self.idrv)
