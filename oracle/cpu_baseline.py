"""TEST/BENCH INFRASTRUCTURE: times the CPU side on the host cores for bench.py's `cpu_baseline` object.

kind "reference": the reference's own Fortran rrtmg_lw (oracle/_ref/libref_nomcica.so, built from /root/reference
by oracle/Makefile) - one process per core, columns split evenly, because the reference has no threading
(SURVEY.md 1).  kind "port": the C restatement (oracle/liboracle.so) when the reference build is absent.
Must run BEFORE the calling process initialises the GPU (workers are spawned interpreters).
"""
from __future__ import annotations

import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor
import multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def physical_cpus():
    """One hardware thread per physical core of the affinity mask (SMT siblings from sysfs; all CPUs when sysfs does not say)."""
    cpus = sorted(os.sched_getaffinity(0))
    seen, out = set(), []
    for c in cpus:
        try:
            sib = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
        except OSError:
            sib = str(c)
        if sib not in seen:
            seen.add(sib)
            out.append(c)
    return out or cpus


def _pin(cpu):
    """Pin this worker to one CPU (one process per core, pinned).  Returns the CPU, or -1 when the mask cannot be
    narrowed (some containers refuse it)."""
    try:
        os.sched_setaffinity(0, {cpu})
        return cpu
    except OSError:
        return -1


def _worker(args):
    ncol, nlay, config, col0, kind, slot = args
    pinned = _pin(slot)
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle.bindings import Oracle, Reference
    from rrtmg_lw_amd.synth import make_gcm_inputs
    eng = Reference("nomcica") if kind == "reference" else Oracle()
    d = make_gcm_inputs(ncol, nlay, config, col0=col0)
    eng.rrtmg_lw(min(ncol, 8), nlay, d["icld"], d["idrv"], make_gcm_inputs(min(ncol, 8), nlay, config))   # warm-up
    t0 = time.perf_counter()
    eng.rrtmg_lw(ncol, nlay, d["icld"], d["idrv"], d)
    return time.perf_counter() - t0, pinned


def _sample_worker(args):
    ncol, nlay, config, kind = args
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle.bindings import Oracle, Reference
    from rrtmg_lw_amd.synth import make_gcm_inputs
    eng = Reference("nomcica") if kind == "reference" else Oracle()
    d = make_gcm_inputs(ncol, nlay, config, col0=0)
    o = eng.rrtmg_lw(ncol, nlay, d["icld"], d["idrv"], d)
    return {k: o[k] for k in ("uflx", "dflx", "uflxc", "dflxc", "hr", "hrc")}


def usable_cores():
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (the GPU box shows 256
    hardware threads but a cpu.max of 16 CPUs; oversubscribing the quota only adds throttling)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def measure(nlay=72, config="cloudy", cols_per_core=6000, cores=None, sample_cols=0):
    from oracle.bindings import Reference
    kind = "reference" if Reference.available("nomcica") else "port"
    if cores is None:
        cores = usable_cores()
    # one process per PHYSICAL core, spread evenly over the cores of the mask (a 16-CPU quota on a 2 x 64-core box: every 8th core)
    phys = physical_cpus()
    stride = max(1, len(phys) // cores)
    jobs = [(cols_per_core, nlay, config, i * cols_per_core, kind, phys[(i * stride) % len(phys)]) for i in range(cores)]
    t0 = time.perf_counter()
    with ProcessPoolExecutor(max_workers=cores, mp_context=mp.get_context("spawn")) as ex:
        res = list(ex.map(_worker, jobs))
        times = [r[0] for r in res]
        npinned = sum(1 for r in res if r[1] >= 0)
        wall = time.perf_counter() - t0
        # the baseline's own results for the first `sample_cols` columns of the workload: bench.py compares the GPU's with them
        sample = ex.submit(_sample_worker, (sample_cols, nlay, config, kind)).result() if sample_cols > 0 else None
    total = cols_per_core * cores
    tmax = max(times)
    return dict(value=total / tmax, unit="columns/s", cores=cores, kind=kind,
                pinned=npinned == cores,
                sample=f"{total} synthetic {nlay}-layer '{config}' columns ({cols_per_core} per process, one process per core, "
                       f"{'each pinned to its own CPU' if npinned == cores else 'unpinned'}, "
                       f"slowest process {tmax:.2f} s, {sum(times):.1f} core-seconds; single-core rate {cols_per_core / (sum(times) / cores):.0f} columns/s)",
                wall_s=wall, sample_outputs=sample)


if __name__ == "__main__":
    import json
    print(json.dumps(measure(cols_per_core=int(sys.argv[1]) if len(sys.argv) > 1 else 2000,
                             cores=int(sys.argv[2]) if len(sys.argv) > 2 else None)))
