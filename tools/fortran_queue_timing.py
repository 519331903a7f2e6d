"""Small-call timing with a FORTRAN host (no Python / ctypes between the host model and the library): tests/fortran/drive_queue.f90 calls
rrtmg_lw per chunk of a few columns, then records the same chunks through module rrtmg_lw_queue and solves them in one pass.
usage (GPU box): python tools/fortran_queue_timing.py > gpurun_out/fortran_queue.md"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rrtmg_lw_amd.synth import make_gcm_inputs
from test_fortran_shim import _compile, _write_nomcica_inputs
tmp = tempfile.mkdtemp()
exe = _compile(tmp, link=True, prog="drive_queue")
env = dict(os.environ, RRTMG_LW_STATIC_TABLES=os.path.join(ROOT, "rrtmg_lw_amd", "data", "lw_static.bin"),
           RRTMG_LW_KDATA=os.path.join(ROOT, "rrtmg_lw_amd", "data", "standin.kdata.bin"))
print("| columns | columns per call | calls | rrtmg_lw per chunk, ms | columns/s | rrtmg_lw_queue (one pass), ms | columns/s |")
print("|---|---|---|---|---|---|---|")
for ncol, chunk in ((16384, 16), (16384, 64), (65536, 64), (65536, 256), (65536, 1024)):
    d = make_gcm_inputs(ncol, 72, "cloudy", col0=0)
    _write_nomcica_inputs(os.path.join(tmp, "in.bin"), d, ncol, 72, 2)
    r = subprocess.run([exe, os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin"), str(chunk)], check=True, env=env, cwd=tmp, timeout=900,
                       capture_output=True, text=True).stdout
    import re
    f = dict(re.findall(r"(\w+)=\s*([-+.\dEe]+)", r))
    dms, qms = float(f["direct_ms"]), float(f["queue_ms"])
    print(f"| {ncol} | {chunk} | {f['chunks']} | {dms:.1f} | {ncol / dms * 1e3:,.0f} | {qms:.1f} | {ncol / qms * 1e3:,.0f} |", flush=True)
