#!/usr/bin/env python3
"""Debugging aid (profiles/round5_exec_hazard.md): read k_layer's cell codes back after a call with the narrow staging window only and after
one with the wide window, and list the (record, layer, column) triples that differ.  Needs a tuning build with the read-back entry:
    tools/build_variant.sh dump "-DRRLW_TUNE -DRRLW_DBG_DUMP"
    RRTMG_LW_ALLOW_TUNE_BUILD=1 RRTMG_LW_HIP_LIB=$PWD/exp/lib_dump.so python tools/dbg_codes.py [col0:ncol ...]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
from rrtmg_lw_amd import api
from rrtmg_lw_amd.synth import make_gcm_inputs
api.rrtmg_lw_ini(1004.0, kdata=api.STANDIN_KDATA, device=0)
L = api.lib()
NG = [10, 12, 16, 14, 16, 8, 12, 8, 12, 6, 8, 8, 4, 2, 2, 2]
NQ = [(g + 3) // 4 for g in NG]; QS = np.concatenate([[0], np.cumsum(NQ)])
def qband(q): return int(np.searchsorted(QS, q, side="right"))
def dump(which, nrec, words):
    info = (C.c_int * 2)()
    buf = np.zeros(1 << 24, dtype=np.uint32)
    L.rrtmg_lw_hip_debug_scratch(C.c_int(which), C.c_void_p(buf.ctypes.data), C.c_size_t(buf.nbytes), info)
    ncb, nlay = info[0], info[1]
    n = nrec * nlay * ncb * words
    return buf[:n].reshape(nrec, nlay, ncb, words).copy(), ncb
cases = [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:]] or [(844914, 117), (282237, 341)]
for col0, ncol in cases:
    d = make_gcm_inputs(ncol, 72, "cloudy_orography", col0=col0)
    api.set_column_sort(0)
    res = {}
    for w in (0, 1):
        api.set_wide_window(w); api.rrtmg_lw_from_dict(d, icld=2, idrv=0)
        g, ncb = dump(0, 38, 4); t, _ = dump(1, 38, 4); f, _ = dump(2, 9, 1)
        res[w] = (g[:, :, :ncol], t[:, :, :ncol], f[:, :, :ncol])
    print("ncol", ncol, "ncolb", ncb)
    cf = np.array(d["cldfr"])
    for name, k in (("gas", 0), ("total", 1), ("fw", 2)):
        a, b = res[0][k], res[1][k]
        if name == "total":     # total codes mean something in cloudy cells only
            m = (cf.T > 0)[None, :, :, None]
            df = (a != b) & m
        else:
            df = a != b
        idx = np.argwhere(df.any(axis=3))
        print(name, "differing (record, layer, col) triples:", len(idx))
        if len(idx):
            print("   records", sorted(set(idx[:, 0].tolist())), "bands", sorted(set(qband(q) for q in idx[:, 0])) if name != "fw" else "")
            print("   layers", sorted(set((idx[:, 1] + 1).tolist())))
            cs = sorted(set(idx[:, 2].tolist())); print("   cols", cs[:4], "..", cs[-4:], len(cs))
            for q, l, c in idx[:6]:
                print("     rec", q, "layer", l + 1, "col", c, "narrow", a[q, l, c].view(np.float32) if name != "fw" else a[q, l, c], "wide", b[q, l, c].view(np.float32) if name != "fw" else b[q, l, c])
    # where do the wrong records come from?  look for the same record anywhere in the narrow run's band-5 codes
    a, b = res[0][0], res[1][0]
    idx = np.argwhere((a != b).any(axis=3))
    for q, l, c in idx[:12]:
        hit = np.argwhere((a[14:18] == b[q, l, c]).all(axis=3))
        print("   wide rec", q, "layer", l + 1, "col", c, "found in the narrow run at (rec-14, layer-1, col):", hit[:4].tolist())
