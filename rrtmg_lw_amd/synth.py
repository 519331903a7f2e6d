"""Synthetic GCM-interface inputs for the benchmark / parity configurations of BASELINE.json.

The reference ships no multi-column or 72/137-layer case, so the benchmark columns are
generated: the 51-layer mid-latitude-summer example profile (rrtmg_lw_amd/data/mls_base.bin) regridded to
``nlay`` layers, optionally perturbed per column with a counter-based PRNG (splitmix64 keyed by the *global*
column index, so any shard of the columns can be generated independently and reproducibly, on the host with
numpy or directly in HBM with torch).

configs
  "clear"    identical columns, icld = 0                                    (BASELINE config 2)
  "cloudy"   perturbed columns, clouds in layers 6-14, icld = 2,
             inflag 2 / iceflag 3 / liqflag 1                                (config 3)
  "aer_idrv" cloudy + aerosol optical depth in layers 1-12, idrv = 1        (config 5, usually nlay = 137)
Cloud-field variants of "cloudy" (same columns, other vertical extents; they measure how the step time depends on where
the clouds are - the sweeps treat the layers above a column block's highest cloud as clear sky):
  "cloudy_deep"     every cloudy column's deck reaches from layer 6 to a top drawn uniformly from layers 14 .. 45
  "cloudy_towers"   2 % of the columns reach layer 45, in runs of 32 consecutive columns (a convective system ~800 km wide on
                    a 0.25-degree grid whose columns are stored longitude-fastest); the rest as "cloudy" (layers 6-14)
  "cloudy_scatter"  the same 2 %, each tower column on its own (no spatial coherence at all: the worst case for any
                    per-block treatment)
Pressure-grid variant of "cloudy" (same clouds; it measures what k_layer's staging window - the two reference-pressure planes of a
256-column workgroup's layer - costs on a terrain-following grid, where neighbouring columns of one model level sit at
different pressures, reference src/rrtmg_lw_setcoef.f90:276-284: `jp` per layer):
  "cloudy_orography"  surface-pressure factor 0.97-1.03 for 70 % of the columns; the rest lie in mountain ranges - runs of 8-64
                    consecutive columns (run length fixed per block of 64 columns) with a factor between 0.55 and 0.95 that
                    is common to the run up to +-0.03 - and the layer temperatures follow the terrain (the profile is read at
                    the column's own pressures)
"""
from __future__ import annotations

import os

import numpy as np

from .blob import read_blob

NBND = 16
SEED = 20240607
CLOUDY_CONFIGS = ("cloudy", "aer_idrv", "cloudy_deep", "cloudy_towers", "cloudy_scatter", "cloudy_orography")
TOWER_TOP = 45          # highest cloud layer (1-based) of the deep / tower variants
TOWER_RUN = 32          # consecutive columns of one tower system ("cloudy_towers")
_BASE = None


def _base():
    global _BASE
    if _BASE is None:
        _BASE = read_blob(os.path.join(os.path.dirname(__file__), "data", "mls_base.bin"))
    return _BASE


def base_profile(nlay):
    """MLS profile regridded to nlay layers: interface pressures log-spaced 1013 -> 0.067 hPa,
    temperature and mixing ratios linear in ln p."""
    b = _base()
    lnp_lev = np.linspace(np.log(1013.0), np.log(0.067), nlay + 1)
    plev = np.exp(lnp_lev)
    play = 0.5 * (plev[:-1] + plev[1:])
    # np.interp wants increasing x: use -ln p
    tlev = np.interp(-lnp_lev, -np.log(b["pz"]), b["tz"])
    tlay = np.interp(-np.log(play), -np.log(b["pavel"]), b["tavel"])
    vmr = np.stack([np.interp(-np.log(play), -np.log(b["pavel"]), b["vmr"][m]) for m in range(7)])
    return dict(plev=plev, play=play, tlev=tlev, tlay=tlay, vmr=vmr, tsfc=float(b["tbound"][0]))


# aerosol optical depth per band for layers 1..12 (pattern of run_examples_std_atm/in_aer_rrtm-aer12)
_AER_LOW = np.array([0.0005, 0.0020, 0.0040, 0.0060, 0.0080, 0.0100, 0.0100, 0.0080, 0.0060, 0.0040, 0.0020, 0.0005])
_AER_BANDFAC = np.array([1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1], dtype=float)


class _NP:
    """numpy backend"""
    name = "numpy"

    def __init__(self):
        self.f64 = np.float64

    def arange(self, a, b):
        return np.arange(a, b, dtype=np.uint64)

    def uniform(self, key):
        with np.errstate(over="ignore"):
            z = key.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(SEED)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
        return (z >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))

    def asarray(self, a):
        return np.asarray(a, dtype=np.float64)

    def zeros(self, shape):
        return np.zeros(shape, dtype=np.float64)

    def where(self, c, a, b):
        return np.where(c, a, b)

    def floor(self, a):
        return np.floor(a)

    def log(self, a):
        return np.log(a)

    def interp_rows(self, x, xp_, fp_):
        """piecewise-linear fp(xp) (xp increasing, 1-D) at every element of x"""
        return np.interp(x, xp_, fp_)

    def col_major(self, a):
        """(ncol, ...) logical array -> Fortran-contiguous storage"""
        return np.asfortranarray(a)


class _Torch:
    """torch backend: arrays live on `device`; logical shape (ncol, nlay) is stored column-fastest."""
    name = "torch"

    def __init__(self, device):
        import torch
        self.t = torch
        self.device = device

    def arange(self, a, b):
        return self.t.arange(a, b, dtype=self.t.int64, device=self.device)

    def _lsr(self, z, n):
        return (z >> n) & ((1 << (64 - n)) - 1)

    def uniform(self, key):
        t = self.t
        # two's-complement int64 arithmetic wraps exactly like uint64
        def c(v):
            return v - (1 << 64) if v >= (1 << 63) else v
        z = key * c(0x9E3779B97F4A7C15) + SEED
        z = (z ^ self._lsr(z, 30)) * c(0xBF58476D1CE4E5B9)
        z = (z ^ self._lsr(z, 27)) * c(0x94D049BB133111EB)
        z = z ^ self._lsr(z, 31)
        return self._lsr(z, 11).to(t.float64) * (1.0 / (1 << 53))

    def asarray(self, a):
        return self.t.as_tensor(np.asarray(a, dtype=np.float64), device=self.device)

    def zeros(self, shape):
        return self.t.zeros(shape, dtype=self.t.float64, device=self.device)

    def where(self, c, a, b):
        return self.t.where(c, a, b)

    def floor(self, a):
        return self.t.floor(a)

    def log(self, a):
        return self.t.log(a)

    def interp_rows(self, x, xp_, fp_):
        """piecewise-linear fp(xp) (xp increasing, 1-D numpy) at every element of the tensor x - np.interp's arithmetic, clamped ends"""
        t = self.t
        xs = t.as_tensor(np.asarray(xp_, dtype=np.float64), device=self.device)
        fs = t.as_tensor(np.asarray(fp_, dtype=np.float64), device=self.device)
        i = t.clamp(t.searchsorted(xs, x.contiguous(), right=True) - 1, 0, xs.numel() - 2)
        x0, x1, f0, f1 = xs[i], xs[i + 1], fs[i], fs[i + 1]
        w = t.clamp((x - x0) / (x1 - x0), 0.0, 1.0)
        return f0 + w * (f1 - f0)

    def col_major(self, a):
        # store so that the first logical index is fastest in memory: permute-reverse, make contiguous
        nd = a.dim()
        if nd == 1:
            return a.contiguous()
        return a.permute(*reversed(range(nd))).contiguous().permute(*reversed(range(nd)))


def make_gcm_inputs(ncol, nlay, config="clear", col0=0, backend="numpy", device=None):
    """Returns a dict with every rrtmg_lw (non-McICA) input, shaped as the Fortran interface declares them
    (src/rrtmg_lw_rad.nomcica.f90:219-276) and stored column-fastest, plus icld/idrv/inflglw/iceflglw/liqflglw."""
    xp = _NP() if backend == "numpy" else _Torch(device)
    bp = base_profile(nlay)
    perturbed = config in CLOUDY_CONFIGS
    col = xp.arange(col0, col0 + ncol)                      # global column ids

    def u(stream, lay=None):
        """uniform[0,1): per column (lay None) or per (column, layer) -> shape (ncol,) / (ncol, nlay)"""
        if lay is None:
            return xp.uniform(col * 65536 + stream)
        lays = xp.arange(0, lay)
        return xp.uniform((col[:, None] * 1024 + lays[None, :]) * 64 + stream)

    ones = xp.zeros((ncol, 1)) + 1.0
    d = {}
    d["play"] = ones * xp.asarray(bp["play"])[None, :]
    d["plev"] = ones * xp.asarray(bp["plev"])[None, :]
    tlay = ones * xp.asarray(bp["tlay"])[None, :]
    tlev = ones * xp.asarray(bp["tlev"])[None, :]
    q = ones * xp.asarray(bp["vmr"][0])[None, :]
    tsfc = xp.zeros((ncol,)) + bp["tsfc"]
    if perturbed:
        dT = (u(1) * 20.0 - 10.0)[:, None]
        tlay = tlay + dT + (u(2, nlay) * 2.0 - 1.0)
        # interfaces: same column offset (keeps tlev between adjacent tlay up to the layer noise)
        tlev = tlev + dT
        q = q * (0.3 + 1.2 * u(3))[:, None]
        tsfc = tlay[:, 0] + (u(4) * 7.0 - 2.0)
        psf = (0.95 + 0.08 * u(14))[:, None]                # surface-pressure factor (sigma-like grid)
        if config == "cloudy_orography":
            # mountain ranges: the block of 64 columns fixes the run length (8, 16, 32 or 64), the run decides whether it is a range (30 %)
            # and how high (factor 0.55-0.95); the columns of a run differ by +-0.03.  Elsewhere 0.97-1.03.
            blk = col // 64
            run_len = 8 * (1 << (xp.floor(xp.uniform(blk * 65536 + 17) * 4.0)).astype(col.dtype) if backend == "numpy" else
                           1 << xp.floor(xp.uniform(blk * 65536 + 17) * 4.0).to(col.dtype))
            run = col // run_len
            ur = xp.uniform(run * 65536 + 18)
            high = 0.55 + 0.40 * xp.uniform(run * 65536 + 19) + 0.06 * (u(14) - 0.5)
            flat = 0.97 + 0.06 * u(14)
            psf = xp.where(ur < 0.30, high, flat)[:, None]
        d["play"] = d["play"] * psf
        d["plev"] = d["plev"] * psf
        if config == "cloudy_orography":
            # temperatures follow the terrain: the base profile read at the column's own pressures (not at the model level's nominal one)
            nlnp = -np.log(bp["play"])
            tlay = xp.interp_rows(-xp.log(d["play"]), nlnp, bp["tlay"]) + dT + (u(2, nlay) * 2.0 - 1.0)
            tlev = xp.interp_rows(-xp.log(d["plev"]), -np.log(bp["plev"]), bp["tlev"]) + dT
            tsfc = tlay[:, 0] + (u(4) * 7.0 - 2.0)
    d["tlay"], d["tlev"], d["tsfc"], d["h2ovmr"] = tlay, tlev, tsfc, q
    d["co2vmr"] = ones * xp.asarray(bp["vmr"][1])[None, :]
    d["o3vmr"] = ones * xp.asarray(bp["vmr"][2])[None, :]
    d["n2ovmr"] = ones * xp.asarray(bp["vmr"][3])[None, :]
    d["ch4vmr"] = ones * xp.asarray(bp["vmr"][5])[None, :]
    d["o2vmr"] = ones * xp.asarray(bp["vmr"][6])[None, :]
    zl = xp.zeros((ncol, nlay))
    d["cfc11vmr"] = zl + 2.6e-10
    d["cfc12vmr"] = zl + 5.0e-10
    d["cfc22vmr"] = zl + 1.5e-10
    d["ccl4vmr"] = zl + 1.0e-10
    emis = xp.zeros((ncol, NBND)) + 1.0
    if perturbed:
        emis = emis - 0.04 * u(5)[:, None]
    d["emis"] = emis

    cldfr, clwp, ciwp = zl + 0.0, zl + 0.0, zl + 0.0
    rel, rei = zl + 10.0, zl + 30.0
    if perturbed:
        lo, hi = 5, min(14, nlay)                         # layers 6..14 (1-based)
        cloudy_col = (u(6) >= 0.30)[:, None]
        top = None                                        # per-column top (number of layers below it) where it differs from `hi`
        if config in ("cloudy_deep", "cloudy_towers", "cloudy_scatter"):
            hmax = min(TOWER_TOP, nlay)
            if config == "cloudy_deep":
                top = hi + xp.floor(u(15) * (hmax - hi + 1))
            else:
                key = xp.arange(col0, col0 + ncol)
                if config == "cloudy_towers":
                    key = key // TOWER_RUN
                tower = xp.uniform(key * 65536 + 16) < 0.02
                top = xp.where(tower, xp.zeros((ncol,)) + hmax, xp.zeros((ncol,)) + hi)
            hi = hmax
        cf = u(7, nlay)[:, lo:hi]
        if top is not None:                               # layers at or above the column's top stay clear
            lays = xp.asarray(np.arange(lo, hi, dtype=np.float64))
            cf = xp.where(lays[None, :] < top[:, None], cf, cf * 0.0)
        # half of the cloudy cells are zeroed so clear gaps occur inside the cloud deck (exercises istcld)
        cf = xp.where(u(8, nlay)[:, lo:hi] < 0.25, cf * 0.0, cf)
        cf = xp.where(cloudy_col, cf, cf * 0.0)
        cldfr[:, lo:hi] = cf
        clwp[:, lo:hi] = u(9, nlay)[:, lo:hi] * 60.0
        ciwp[:, lo:hi] = u(10, nlay)[:, lo:hi] * 20.0
        rel[:, lo:hi] = 5.0 + 15.0 * u(11, nlay)[:, lo:hi]
        rei[:, lo:hi] = 15.0 + 85.0 * u(12, nlay)[:, lo:hi]
    d["cldfr"], d["cliqwp"], d["cicewp"], d["reliq"], d["reice"] = cldfr, clwp, ciwp, rel, rei
    d["taucld"] = xp.zeros((NBND, ncol, nlay))
    tauaer = xp.zeros((ncol, nlay, NBND))
    if config == "aer_idrv":
        k = min(12, nlay)
        prof = xp.asarray(_AER_LOW[:k])[None, :, None] * xp.asarray(_AER_BANDFAC)[None, None, :]
        tauaer[:, :k, :] = prof * (0.5 + u(13))[:, None, None]
    d["tauaer"] = tauaer

    out = {k: xp.col_major(v) for k, v in d.items()}
    out.update(ncol=ncol, nlay=nlay, inflglw=2, iceflglw=3, liqflglw=1,
               icld=2 if perturbed else 0, idrv=1 if config == "aer_idrv" else 0)
    return out


# ------------------------------------------------------------------------------------------------------------------
# Stress variants: inputs that reach the branches the benchmark columns never take (numpy only; deterministic).
#   "highgas"   CO2 x {2, 4, 8} and N2O x {2, 2.5, 3} by column: the `ratx > thr` adjustment of the minor-gas amounts
#               (reference src/rrtmg_lw_taumol.f90:547-554, 1352-1359, 1478-1487, 1718-1725, 2494-2501)
#   "cold"      every temperature 70 K lower: Planck-table index and jt / indself / indminor clamps at the cold end
#   "hot"       every temperature 70 K higher: the clamps at the warm end (reference src/rrtmg_lw_setcoef.f90:174-178, 294-305)
#   "allupper"  the whole column above 95.6 hPa (laytrop = 0), pressures down to 0.006 hPa (jp clamps at 58)
#   "alllower"  the whole column below that level (laytrop = nlay)
# ------------------------------------------------------------------------------------------------------------------
STRESS_KINDS = ("highgas", "cold", "hot", "allupper", "alllower")


def make_stress_inputs(kind, ncol, nlay, config="cloudy", col0=0):
    d = make_gcm_inputs(ncol, nlay, config, col0=col0)
    c = np.arange(ncol)
    if kind == "highgas":
        fco2 = np.array([2.0, 4.0, 8.0, 1.0])[c % 4][:, None]
        fn2o = np.array([1.0, 2.0, 2.5, 3.0])[(c // 2) % 4][:, None]
        d["co2vmr"] = np.asfortranarray(np.array(d["co2vmr"]) * fco2)
        d["n2ovmr"] = np.asfortranarray(np.array(d["n2ovmr"]) * fn2o)
    elif kind in ("cold", "hot"):
        dt = -70.0 if kind == "cold" else 70.0
        for k in ("tlay", "tlev", "tsfc"):
            d[k] = np.asfortranarray(np.array(d[k]) + dt)
    elif kind in ("allupper", "alllower"):
        # ln p of every level mapped linearly onto [ln 1013, ln 110] (all lower) or [ln 90, ln 0.006] (all upper)
        lo, hi = (np.log(1013.0), np.log(110.0)) if kind == "alllower" else (np.log(90.0), np.log(0.006))
        s0, s1 = np.log(1013.0), np.log(0.067)
        for k in ("play", "plev"):
            p = np.array(d[k])
            f = p / base_profile(nlay)[k][None, :]            # the per-column surface-pressure factor of the base set
            x = (np.log(base_profile(nlay)[k]) - s0) / (s1 - s0)
            d[k] = np.asfortranarray(np.exp(lo + x * (hi - lo))[None, :] * f)
    else:
        raise ValueError(kind)
    return d
