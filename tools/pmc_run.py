#!/usr/bin/env python3
"""Workload for rocprofv3 counter passes (run as `rocprofv3 --pmc <counters> --kernel-trace -d <dir> -f csv -- python3 tools/pmc_run.py`).

One warm-up call and ONE measured-shape call of the hot path on `--ncol` device-resident columns (default 250000 = two
internal batches of 125000 columns, the launch shape of the 1e6-column bench), preceded by a calibration kernel that moves a known
number of bytes (rrtmg_lw_hip_calibrate_stream) so that tools/pmc_summarize.py can fix the unit/scale of FETCH_SIZE and
WRITE_SIZE in the same session (MI355X_MICROARCH.md, HBM section).
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ncol", type=int, default=250000)
    ap.add_argument("--nlay", type=int, default=72)
    ap.add_argument("--config", default="cloudy")
    ap.add_argument("--mcica", type=int, default=0)
    ap.add_argument("--calib-bytes", type=int, default=1 << 30)
    args = ap.parse_args()
    import torch
    from rrtmg_lw_amd import api
    from rrtmg_lw_amd.shard import output_rows, output_views
    from rrtmg_lw_amd.synth import make_gcm_inputs
    dev = torch.device("cuda", 0)
    api.rrtmg_lw_ini(1004.0, device=0)
    api._check(api.lib().rrtmg_lw_hip_calibrate_stream(ctypes.c_longlong(args.calib_bytes)))
    d = make_gcm_inputs(args.ncol, args.nlay, args.config, backend="torch", device=dev)
    outbuf = torch.zeros((output_rows(args.nlay), args.ncol), dtype=torch.float64, device=dev)
    out = output_views(outbuf, args.nlay)
    stream = torch.cuda.current_stream().cuda_stream
    alpha = None
    if args.mcica:
        alpha = torch.full((args.nlay, args.ncol), 0.6, dtype=torch.float64, device=dev).t()
    for _ in range(2):
        if args.mcica:
            api.rrtmg_lw_mcica_subcol_device(d, out, 140, 0, alpha=alpha, icld=args.mcica, stream=stream)
        else:
            api.rrtmg_lw_device(d, out, stream=stream)
        api.check(stream)
    api.finalize()


if __name__ == "__main__":
    main()
