#!/usr/bin/env python3
"""Stress of the combining entry: several Python threads call the host-pointer entry at once with chunks of random sizes (64 .. 3000 columns,
two layer counts, idrv 0 / 1, icld 1 / 2 mixed), on one device and on three virtual ones (fan-out inside a combined pass); every result is
compared bit for bit with the same columns taken from one big call.   usage: python tools/soak_concurrent.py [seconds] [threads]"""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rrtmg_lw_amd import api
from rrtmg_lw_amd.synth import make_gcm_inputs

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
nth = int(sys.argv[2]) if len(sys.argv) > 2 else 8
OUT = ("uflx", "dflx", "hr", "uflxc", "dflxc", "hrc")


def part(d, keys, c0, c1):
    p = dict(d); p["ncol"] = c1 - c0
    for k in keys:
        v = d[k]
        p[k] = np.ascontiguousarray(v[c0:c1]) if v.ndim == 1 else (np.asfortranarray(v[:, c0:c1, :]) if k == "taucld" else np.asfortranarray(v[c0:c1]))
    return p


def run(devices):
    if len(devices) > 1:
        api.init_devices(devices, kdata=api.STANDIN_KDATA)
    else:
        api.rrtmg_lw_ini(1004.0, kdata=api.STANDIN_KDATA, device=devices[0])
    cases = []
    for nlay, cfg, icld in ((40, "cloudy", 2), (40, "aer_idrv", 1), (31, "cloudy", 1)):
        d = make_gcm_inputs(6000, nlay, cfg, col0=7 * nlay)
        whole = api.rrtmg_lw_from_dict(d, icld=icld)
        cases.append((d, [k for k, v in d.items() if isinstance(v, np.ndarray)], whole, icld))
    stop = time.time() + secs / 2
    bad, calls = [], [0] * nth

    def worker(t):
        rng = np.random.default_rng(100 + t)
        while time.time() < stop and not bad:
            d, keys, whole, icld = cases[rng.integers(len(cases))]
            n = int(rng.choice([64, 100, 256, 700, 1500, 3000]))
            c0 = int(rng.integers(0, 6000 - n))
            got = api.rrtmg_lw_from_dict(part(d, keys, c0, c0 + n), icld=icld)
            calls[t] += 1
            for k in OUT + (("duflx_dt", "duflxc_dt") if d["idrv"] == 1 else ()):
                if not np.array_equal(got[k], whole[k][c0:c0 + n]):
                    bad.append((t, k, c0, n, float(np.abs(got[k] - whole[k][c0:c0 + n]).max())))
                    return

    th = [threading.Thread(target=worker, args=(t,)) for t in range(nth)]
    for x in th: x.start()
    for x in th: x.join()
    return sum(calls), bad


c0, p0 = api.combine_stats()
for devs in ([0], [0, 0, 0]):
    n, bad = run(devs)
    c1, p1 = api.combine_stats()
    print(f"devices {devs}: {n} calls from {nth} threads, {c1 - c0} through the combining entry in {p1 - p0} passes; mismatches: {bad[:3]}")
    c0, p0 = c1, p1
    if bad:
        sys.exit(1)
print("soak ok")
