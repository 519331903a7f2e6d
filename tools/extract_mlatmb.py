#!/usr/bin/env python3
"""Build rrtmg_lw_amd/data/mlatmb.npz from the *data* statements of the reference's six built-in model atmospheres
(BLOCK DATA MLATMB, src/rrtatm.f:1807-2912: altitudes, pressures, temperatures, mixing ratios of the seven main
molecules and the air density at 50 levels, 21 trace-gas profiles) and the molecular weights (BLOCK DATA ATMCON,
src/rrtatm.f:1798-1809).  The numbers are physical input data (AFGL atmospheric constituent profiles); they are read
in this container and stored as a binary table, no reference source text is kept.

Run:  python tools/extract_mlatmb.py [/root/reference]
"""
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TRACE = ("ANO", "SO2", "ANO2", "ANH3", "HNO3", "OH", "HF", "HCL", "HBR", "HI", "CLO", "OCS", "H2CO", "HOCL", "AN2", "HCN",
         "CH3CL", "H2O2", "C2H2", "C2H6", "PH3")


def data_blocks(text):
    """name -> list of floats for every fixed-form `DATA NAME / v, v, ... /` statement (repeat counts N*v expanded; the
    MXZ50*0.0 padding dropped)."""
    out = {}
    lines = [ln[:72] for ln in text.splitlines() if ln[:1] not in "Cc*!"]
    joined = []
    for ln in lines:
        if len(ln) > 5 and ln[5] not in " 0" and joined:       # continuation
            joined[-1] += " " + ln[6:]
        else:
            joined.append(ln[6:] if len(ln) > 6 else "")
    for st in joined:
        m = re.match(r"\s*DATA\s+(\w+)\s*/(.*)/\s*$", st.strip(), re.S)
        if not m:
            continue
        vals = []
        for tok in m.group(2).split(","):
            tok = tok.strip()
            if not tok or tok.upper().startswith("MXZ50"):
                continue
            if "*" in tok:
                n, v = tok.split("*")
                vals += [float(v.replace("D", "E"))] * int(n)
            else:
                try:
                    vals.append(float(tok.replace("D", "E")))
                except ValueError:
                    vals = None
                    break
        if vals:
            out[m.group(1).upper()] = vals
    return out


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    src = open(os.path.join(ref, "src", "rrtatm.f")).read().splitlines()
    a = next(i for i, ln in enumerate(src) if ln[:1] == " " and "BLOCK DATA MLATMB" in ln)
    b = next(i for i in range(a, len(src)) if src[i].strip().upper().startswith("END"))
    blk = data_blocks("\n".join(src[a:b]))
    alt = np.array(blk["ALT"][:50])
    pm = np.array([blk[f"P{m}"][:50] for m in range(1, 7)])
    tm = np.array([blk[f"T{m}"][:50] for m in range(1, 7)])
    amol = np.array([[blk[f"AMOL{m}{k}"][:50] for k in range(1, 9)] for m in range(1, 7)])     # (model, species 1..8, level)
    trac = np.array([blk[t][:50] for t in TRACE])
    a2 = next(i for i, ln in enumerate(src) if ln[:1] == " " and "BLOCK DATA ATMCON" in ln)
    b2 = next(i for i in range(a2, len(src)) if src[i].strip().upper().startswith("END"))
    amwt = np.array(data_blocks("\n".join(src[a2:b2]))["AMWT"])
    assert alt.shape == (50,) and pm.shape == (6, 50) and amol.shape == (6, 8, 50) and trac.shape == (21, 50) and amwt.size == 39
    dst = os.path.join(os.path.dirname(HERE), "rrtmg_lw_amd", "data", "mlatmb.npz")
    np.savez_compressed(dst, alt=alt, pm=pm, tm=tm, amol=amol, trac=trac, amwt=amwt)
    print("wrote", dst, os.path.getsize(dst), "bytes; O2 (model 6, level 1):", amol[5, 6, 0], "amwt[:7]", amwt[:7])


if __name__ == "__main__":
    main()
