#!/usr/bin/env python3
"""Build rrtmg_lw_amd/data/lw_static.bin from the reference's *data* statements.

The numbers (Planck integrals, MLS reference atmosphere, cloud absorption coefficients, g-point
reduction maps, band limits) are physical input data of RRTMG_LW v5.0 (BSD-3).  They are read from
the data statements in /root/reference/src at build time in this container and stored as a binary
table blob; no reference source text is kept.  Sources:

  pref, preflog, tref, chi_mls      src/rrtmg_lw_setcoef.f90:437-597   (lwatmref)
  totplnk, totplk16                 src/rrtmg_lw_setcoef.f90:600-1303  (lwavplank)
  totplnkderiv, totplk16deriv       src/rrtmg_lw_setcoef.f90:1306-2009 (lwavplankderiv)
  abscld1, absice0..3, absliq0..1   src/rrtmg_lw_init.f90:2037-2675    (lwcldpr)
  ngc, ngs, ngm, ngn, ngb, wt       src/rrtmg_lw_init.f90:303-382      (lwcmbdat)
  wavenum1/2, delwave, nspa, nspb   src/rrtmg_lw_init.f90:215-228      (lwdatinit)

Run:  python tools/extract_static_tables.py [/root/reference]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from rrtmg_lw_amd.blob import write_blob  # noqa: E402
from rrtmg_lw_amd.f90data import parse_f90_data  # noqa: E402


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    setcoef = open(os.path.join(ref, "src/rrtmg_lw_setcoef.f90")).read()
    init = open(os.path.join(ref, "src/rrtmg_lw_init.f90")).read()

    out = {}
    a, _ = parse_f90_data(setcoef, {"pref": [(1, 59)], "preflog": [(1, 59)], "tref": [(1, 59)],
                                    "chi_mls": [(1, 7), (1, 59)]}, routine="lwatmref")
    out.update(a)
    a, _ = parse_f90_data(setcoef, {"totplnk": [(1, 181), (1, 16)], "totplk16": [(1, 181)]},
                          routine="lwavplank")
    out.update(a)
    a, _ = parse_f90_data(setcoef, {"totplnkderiv": [(1, 181), (1, 16)], "totplk16deriv": [(1, 181)]},
                          routine="lwavplankderiv")
    out.update(a)
    a, s = parse_f90_data(init, {"absice0": [(1, 2)], "absice1": [(1, 2), (1, 5)],
                                 "absice2": [(1, 43), (1, 16)], "absice3": [(1, 46), (1, 16)],
                                 "absliq1": [(1, 58), (1, 16)]},
                          scalars=("abscld1", "absliq0"), routine="lwcldpr")
    out.update(a)
    out["abscld1"] = np.array([s["abscld1"]])
    out["absliq0"] = np.array([s["absliq0"]])
    a, _ = parse_f90_data(init, {"ngc": [(1, 16)], "ngs": [(1, 16)], "ngm": [(1, 256)],
                                 "ngn": [(1, 140)], "ngb": [(1, 140)], "wt": [(1, 16)]},
                          routine="lwcmbdat")
    for k in ("ngc", "ngs", "ngm", "ngn", "ngb"):
        a[k] = a[k].astype(np.int32)
    out.update(a)
    a, _ = parse_f90_data(init, {"wavenum1": [(1, 16)], "wavenum2": [(1, 16)], "delwave": [(1, 16)],
                                 "nspa": [(1, 16)], "nspb": [(1, 16)]}, routine="lwdatinit")
    for k in ("nspa", "nspb"):
        a[k] = a[k].astype(np.int32)
    out.update(a)

    for k, v in out.items():
        if np.isnan(np.asarray(v, dtype=float)).any():
            raise SystemExit(f"{k}: unassigned elements after parsing")
    dst = os.path.join(os.path.dirname(HERE), "rrtmg_lw_amd", "data", "lw_static.bin")
    write_blob(dst, out)
    print("wrote", dst, {k: tuple(np.shape(v)) for k, v in out.items()})


if __name__ == "__main__":
    main()
