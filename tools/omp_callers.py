#!/usr/bin/env python3
"""An OpenMP Fortran host model (tests/fortran/drive_omp.f90, flang -fopenmp) calling rrtmg_lw per chunk from all its threads:
wall time of one thread calling chunk after chunk against all threads at once (the combining entry).
usage: python tools/omp_callers.py [--ncol 65536] [--chunk 64] [--threads 16] [--nlay 72] [--config cloudy]"""
import argparse, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
ap = argparse.ArgumentParser()
ap.add_argument("--ncol", type=int, default=65536)
ap.add_argument("--chunk", default="64")
ap.add_argument("--threads", default="16")
ap.add_argument("--nlay", type=int, default=72)
ap.add_argument("--config", default="cloudy")
ap.add_argument("--env", default="", help="extra VAR=value,... for the host model")
args = ap.parse_args()
from rrtmg_lw_amd.synth import make_gcm_inputs
from test_fortran_shim import _compile_omp, _write_nomcica_inputs
tmp = tempfile.mkdtemp()
exe = _compile_omp(tmp, link=True)
d = make_gcm_inputs(args.ncol, args.nlay, args.config, col0=5)
_write_nomcica_inputs(os.path.join(tmp, "in.bin"), d, args.ncol, args.nlay, d["icld"])
for chunk in args.chunk.split(","):
    for th in args.threads.split(","):
        env = dict(os.environ, RRTMG_LW_STATIC_TABLES=os.path.join(ROOT, "rrtmg_lw_amd", "data", "lw_static.bin"),
                   RRTMG_LW_KDATA=os.path.join(ROOT, "rrtmg_lw_amd", "data", "standin.kdata.bin"), OMP_NUM_THREADS=th)
        for kv in filter(None, args.env.split(",")):
            k, v = kv.split("="); env[k] = v
        r = subprocess.run([exe, os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin"), chunk], env=env, cwd=tmp, timeout=900, capture_output=True, text=True)
        print(r.stdout.strip(), flush=True)
        if r.returncode != 0 or args.env:
            print(r.stderr[-3000:], flush=True)
