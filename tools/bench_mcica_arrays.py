#!/usr/bin/env python3
"""Rate of the McICA flavour through the reference's own argument list (explicit (140,ncol,nlay) sub-column arrays, device resident),
beside the fused generator+solver entry on the same columns - the comparison quoted in DESIGN.md."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ncol", type=int, default=131072)
    ap.add_argument("--nlay", type=int, default=72)
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    import torch
    from rrtmg_lw_amd import api
    from rrtmg_lw_amd.shard import output_rows, output_views
    from rrtmg_lw_amd.synth import make_gcm_inputs
    dev = torch.device("cuda", 0)
    api.rrtmg_lw_ini(1004.0, device=0)
    n, L = args.ncol, args.nlay
    d = make_gcm_inputs(n, L, "cloudy", backend="torch", device=dev)
    out = output_views(torch.zeros((output_rows(L), n), dtype=torch.float64, device=dev), L)
    sub = {k: torch.zeros((L, n, 140), dtype=torch.float64, device=dev) for k in ("cldfmcl", "ciwpmcl", "clwpmcl", "taucmcl")}
    sub["reicmcl"] = torch.zeros((L, n), dtype=torch.float64, device=dev)
    sub["relqmcl"] = torch.zeros((L, n), dtype=torch.float64, device=dev)
    alpha = torch.full((L, n), 0.6, dtype=torch.float64, device=dev)
    s = torch.cuda.current_stream().cuda_stream

    def timed(fn):
        fn(); api.check(s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        api.check(s)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps

    t_gen = timed(lambda: api.mcica_subcol_device(d, sub, 5, 1, 0, alpha=alpha, stream=s))
    t_arr = timed(lambda: api.rrtmg_lw_mcica_device(d, sub, out, icld=2, stream=s))
    ref = {k: out[k].clone() for k in ("uflx", "dflx", "hr")}
    t_fus = timed(lambda: api.rrtmg_lw_mcica_subcol_device(d, out, 1, 0, alpha=alpha, icld=5, stream=s))
    same = all(torch.equal(ref[k], out[k]) for k in ref)
    print(json.dumps(dict(columns=n, nlay=L, generator_arrays_ms=round(1e3 * t_gen, 2), solver_arrays_ms=round(1e3 * t_arr, 2),
                          arrays_path_columns_per_s=round(n / (t_gen + t_arr), 1), fused_ms=round(1e3 * t_fus, 2),
                          fused_columns_per_s=round(n / t_fus, 1), fused_equals_arrays_bitwise=bool(same))))
    api.finalize()


if __name__ == "__main__":
    main()
