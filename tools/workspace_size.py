#!/usr/bin/env python3
"""Device memory the library holds after a call of each shape (rrtmg_lw_hip_workspace_bytes), per column of the batch."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from rrtmg_lw_amd import api
from rrtmg_lw_amd.synth import make_gcm_inputs
from rrtmg_lw_amd.shard import output_rows, output_views
dev = torch.device("cuda", 0)
lib = api.lib(); lib.rrtmg_lw_hip_workspace_bytes.restype = ctypes.c_longlong
n = 262144
for nlay, cfg, icld in ((72, "clear", 0), (72, "cloudy", 2), (72, "cloudy", 1), (72, "aer_idrv", 2), (137, "aer_idrv", 2)):
    api.rrtmg_lw_ini(1004.0, kdata=api.STANDIN_KDATA, device=0)          # (a fresh state: the workspace only grows)
    d = make_gcm_inputs(n, nlay, cfg, backend="torch", device=dev)
    buf = torch.zeros((output_rows(nlay, d["idrv"]), n), dtype=torch.float64, device=dev)
    api.rrtmg_lw_device(d, output_views(buf, nlay, d["idrv"]), icld=icld, stream=0); api.check(0)
    b = lib.rrtmg_lw_hip_workspace_bytes()
    print(f"nlay {nlay} config {cfg} icld {icld} idrv {d['idrv']}: {b / 1e9:.2f} GB for a batch of {n} columns = {b / n / 1e3:.1f} KB per column", flush=True)
    del d, buf
api.finalize()
