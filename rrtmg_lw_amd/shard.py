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
