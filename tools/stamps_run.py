"""Diagnostic: cycle shares of k_layer's segments from a -DRRLW_LAYER_STAMPS build (RRTMG_LW_HIP_LIB=exp/lib_<name>.so).
usage: python tools/stamps_run.py [--config cloudy] [--ncol 250000]"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rrtmg_lw_amd import api
from rrtmg_lw_amd.synth import make_gcm_inputs
from rrtmg_lw_amd.shard import output_rows, output_views
ap = argparse.ArgumentParser(); ap.add_argument("--config", default="cloudy"); ap.add_argument("--ncol", type=int, default=250000); ap.add_argument("--nlay", type=int, default=72)
a = ap.parse_args()
dev = torch.device("cuda", 0)
api.rrtmg_lw_ini(1004.0, kdata=api.STANDIN_KDATA, device=0)
parts = [make_gcm_inputs(min(125000, a.ncol - s), a.nlay, a.config, col0=s, backend="torch", device=dev) for s in range(0, a.ncol, 125000)]
d = dict(parts[0]); d["ncol"] = a.ncol
for k, v in parts[0].items():
    if torch.is_tensor(v) and len(parts) > 1:
        cat = torch.cat([p[k] for p in parts], dim=1 if k == "taucld" else 0); nd = cat.dim()
        d[k] = cat.permute(*reversed(range(nd))).contiguous().permute(*reversed(range(nd))) if nd > 1 else cat.contiguous()
out = output_views(torch.zeros((output_rows(a.nlay, d["idrv"]), a.ncol), dtype=torch.float64, device=dev), a.nlay, d["idrv"])
st = torch.cuda.current_stream().cuda_stream
api.rrtmg_lw_device(d, out, stream=st); api.check(st)
buf = (C.c_ulonglong * 9)()
api.lib().rrtmg_lw_hip_debug_stamps(buf, 9)
api.rrtmg_lw_device(d, out, stream=st); api.check(st)
api.lib().rrtmg_lw_hip_debug_stamps(buf, 9)
names = ["barrier A (wait for the workgroup; __syncthreads: + store drain)", "staging: loads -> LDS writes -> barrier B", "rows_prep", "row combination (LDS reads + FMAs)",
         "cells: codes + stores", "prologue: inatm + setcoef + window", "-", "-"]
tot = sum(buf[i] for i in range(8)); n = buf[8]
print(f"waves {n}, cycles per wave {tot / max(n, 1):.0f}")
for i in range(6):
    print(f"  {names[i]:70s} {buf[i] / max(n, 1):9.0f} cycles  {100.0 * buf[i] / max(tot, 1):5.1f} %")
