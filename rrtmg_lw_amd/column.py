"""Standalone column mode:  python -m rrtmg_lw_amd.column INPUT_RRTM [--cld IN_CLD_RRTM] [--aer IN_AER_RRTM] [-o OUTPUT_RRTM]

The sequence of the reference's column driver (src/rrtmg_lw.1col.f90:440-704) over the HIP library: read the profile
(readprof / readcld / readaer -> rrtmg_lw_amd.io_rrtm), run the prepared-column entry for the spectral intervals the IOUT
switch asks for - IOUT = 0 the 10-3250 cm-1 total, 1..16 one band, 99 the total followed by the sixteen bands (:452-466,
:689-696) - with IMCA = 1 as the mean over the driver's 200 Mersenne-Twister samples (:457-459, :471-480, :641-660) and with
IDRV = 1 the adjustment of the upward fluxes by DTBOUND (:585-610), and write OUTPUT_RRTM in the driver's record formats.
IATM = 1 inputs (a level sounding; the reference calls RRTATM) are layered by rrtmg_lw_amd/atmpth.py.  There is no CPU path: a GPU and
the built library are required.
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

from . import api
from .io_rrtm import WAVENUM1, WAVENUM2, read_input_rrtm, write_output_rrtm  # noqa: F401

NMCA = 200          # src/rrtmg_lw.1col.f90:459
GRAV, SECDY = 9.8066, 8.6400e4          # src/rrtmg_lw_init.f90:243-251


def band_sequence(iout):
    """(istart, iend) of every output block the driver writes for this IOUT (src/rrtmg_lw.1col.f90:452-466, :689-696)."""
    if iout < 0:
        return []
    if 0 < iout <= 40:
        return [(iout, iout)]
    seq = [(1, 16)]
    if iout == 99:
        seq += [(b, b) for b in range(1, 17)]
    return seq


def run_case(col, cpdair=1004.0):
    """All output blocks of one prepared column: list of dict(istart, iend, pz, uflx, dflx, fnet, htr) (total-sky, as the driver prints)."""
    nl = int(col["nlayers"])
    pz = np.asarray(col["pz"], dtype=np.float64)
    heatfac = 1.0e-2 * GRAV * SECDY / cpdair                                   # src/rrtmg_lw_init.f90:298
    blocks = []
    alpha = None
    if int(col["imca"]) == 1 and int(col["icld"]) in (4, 5):
        r2 = lambda v: np.asarray(v, dtype=np.float64).reshape((1, nl))
        alpha = api.get_alpha(1, nl, int(col["icld"]), int(col["idcor"]), float(col["decorr_con"]), r2(col["dz"]),
                              np.array([float(col["lat"])]), int(col["juldat"]), r2(col["cldfrac"]))[0]
    subs = None
    for istart, iend in band_sequence(int(col["iout"])):
        if int(col["imca"]) == 1:
            # the same 200 sub-column samples serve every spectral interval (seed ims * 140, :248-251 of the 1col generator)
            if subs is None:
                res, subs = api.column_mcica_samples(col, list(range(1, NMCA + 1)), irng=1, alpha=alpha)
                if (istart, iend) != (1, 16):
                    res = api.run_columns_mcica([col] * NMCA, subs, istart, iend)
            else:
                res = api.run_columns_mcica([col] * NMCA, subs, istart, iend)
            o = {k: v.mean(axis=0) for k, v in res.items()}
        else:
            o = {k: v[0] for k, v in api.run_columns([col], istart, iend).items()}
        up, dn, net, htr = o["totuflux"].copy(), o["totdflux"], o["fnet"].copy(), o["htr"].copy()
        if int(col["idrv"]) == 1:                                                # :585-610
            up = up + o["dtotuflux_dt"] * float(col["dtbound"])
            net = up - dn
            htr = np.zeros(nl + 1)
            htr[:nl] = heatfac * (net[:-1] - net[1:]) / (pz[:-1] - pz[1:])
        blocks.append(dict(istart=istart, iend=iend, pz=pz, uflx=up, dflx=dn, fnet=net, htr=htr))
    return blocks


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m rrtmg_lw_amd.column", description=__doc__.split("\n\n")[0])
    ap.add_argument("input", help="INPUT_RRTM file (IATM = 0 or 1)")
    ap.add_argument("--cld", help="IN_CLD_RRTM file (needed when the input's ICLD > 0)")
    ap.add_argument("--aer", help="IN_AER_RRTM file (needed when the input's IAER = 10)")
    ap.add_argument("-o", "--output", default="OUTPUT_RRTM")
    ap.add_argument("--kdata", help="absorption-coefficient blob (default: data/rrtmg_lw.kdata.bin, else the stand-in with a warning)")
    ap.add_argument("--cpdair", type=float, default=1004.0, help="specific heat of dry air handed to rrtmg_lw_ini (the driver uses 1004.0)")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--airmwt", type=float, default=0.0,
                    help="IATM = 1: mean molecular weight of air for amounts given in g/kg; the reference leaves it unset (0: such amounts "
                         "vanish, as in its output_rrtm_ICRCCM_sonde), its commented-out value is 28.964")
    a = ap.parse_args(argv)
    for f in (a.cld, a.aer):
        if f and not os.path.exists(f):
            raise SystemExit(f"{f} not found")
    col = read_input_rrtm(a.input, a.cld, a.aer, airmwt=a.airmwt)
    api.rrtmg_lw_ini(a.cpdair, kdata=a.kdata, device=a.device)
    try:
        blocks = run_case(col, a.cpdair)
    finally:
        standin = api.kdata_is_standin()
        api.finalize()
    footer = ["  Modules and versions used in this calculation:", "",
              "     rrtmg_lw_amd: librrtmg_lw_hip.so (MI355X), prepared-column entry rrtmg_lw_hip_run_columns"
              + ("_mcica, mean of 200 samples" if int(col["imca"]) == 1 else ""),
              "     absorption coefficients: " + ("STAND-IN tables (non-physical fluxes)" if standin else "rrtmg_lw.kdata.bin")]
    write_output_rrtm(a.output, blocks, footer)
    print(f"{a.output}: {len(blocks)} block(s), {int(col['nlayers'])} layers", file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
