#!/usr/bin/env python3
"""Benchmark of the RRTMG_LW hot path on MI355X:  python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): 1e6 synthetic 72-layer columns,
clouds with maximum-random overlap (icld=2 -> rtrnmr), sharded over the N GPUs of one node in contiguous column
blocks (strong scaling: the total is fixed).  A "step" is one rrtmg_lw call over all columns with every input
already resident in HBM (device-pointer entry of the C ABI); for N > 1 the step ends with the single RCCL
all-gather of the packed flux/heating-rate block that reassembles the outputs on every rank (north_star).
One JSON line is printed by rank 0.

Extra objects in the line:
  roofline      HBM roofline of the PATH: algorithmic bytes of a step (33.0 KB per 72-layer column, SURVEY.md 8d, x columns per rank)
                / the summed kernel time of the step, measured live with HIP events on the launch stream.  The dominant kernel
                (largest total time) is reported inside it with ITS time share of those bytes - the whole column's bytes are not
                attributed to one kernel - and the PMC-measured HBM traffic per launch where a summary is committed under profiles/.
  compute       the instruction side ("so that the low HBM fraction is explained rather than hidden"): vector
                lane-instructions per second of the step against the chip's issue peak, LDS bytes per second against the
                ~150 TB/s the LDS arrays deliver; instruction and LDS counts per column come from the committed PMC pass
                (profiles/pmc_compute.json), times from this run.
  path          per-kernel milliseconds of a step.
  cpu_baseline  the reference's own Fortran (oracle/_ref, "reference") or the C port ("port") on the host cores,
                measured before the GPU is touched, rank 0 at N=1 only.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_COL_72 = 33.0e3      # 29.5 KB in + 3.5 KB out per 72-layer column (non-McICA API)
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
VALU_PEAK_LANE_INSTR = 256 * 4 * 32 * 2.4e9      # 78.6e12: 256 CUs x 4 SIMDs x 32 lanes per clock x 2.4 GHz (f32-class; f64 issues at half of it)
LDS_PEAK_BYTES = 150e12             # MI355X_MICROARCH.md, LDS: ~150 TB/s aggregate for ds_read_b64 / b128


def algo_bytes_per_col(nlay, idrv):
    n_in = 12 * nlay + 2 * (nlay + 1) + 1 + 16 + 5 * nlay + 16 * nlay + 16 * nlay
    n_out = 4 * (nlay + 1) + 2 * nlay + (2 * (nlay + 1) if idrv else 0)
    return 8.0 * (n_in + n_out)


def kernels_sha16():
    """first 16 hex digits of the sha256 of the kernel sources: what the committed PMC files (profiles/pmc_traffic.json) are tied to"""
    import hashlib
    h = hashlib.sha256()
    for f in ("kernels.hip", "driver.hip"):
        h.update(open(os.path.join(ROOT, "rrtmg_lw_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--ncol", type=int, default=1_000_000, help="total columns over all GPUs")
    ap.add_argument("--nlay", type=int, default=72)
    ap.add_argument("--config", default="cloudy", choices=["clear", "cloudy", "aer_idrv", "cloudy_deep", "cloudy_towers", "cloudy_scatter", "cloudy_orography"])
    ap.add_argument("--mcica", type=int, default=0, metavar="ICLD",
                    help="McICA flavour (BASELINE configs[3]): sub-column generator with overlap ICLD (5 = exponential-random) "
                         "+ cldprmc + rtrnmc through the fused device entry; 0 = non-McICA rtrn/rtrnmr")
    ap.add_argument("--batch", type=int, default=0, help="columns per internal batch (0 = library default)")
    ap.add_argument("--no-overlap", action="store_true", help="serialise k_layer and the sweeps of consecutive batches (the library default)")
    ap.add_argument("--overlap", action="store_true", help="run the sweeps of batch i beside k_layer of batch i+1 (tuning; second scratch set)")
    ap.add_argument("--cu-partition", type=int, default=-1, metavar="CUS",
                    help="k_layer of batch i+1 on CUS compute units beside the sweeps of batch i on the others (implies --overlap); 0 = off; -1 = library default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-cols-per-core", type=int, default=6000)
    ap.add_argument("--no-gather", action="store_true", help="skip the RCCL all-gather of the outputs (N > 1)")
    ap.add_argument("--force-gather", action="store_true",
                    help="initialise RCCL and run the all-gather even with one rank (plumbing check under torchrun with N = 1)")
    ap.add_argument("--check", action="store_true", help="compare 256 columns with the CPU oracle before timing")
    ap.add_argument("--n1-prototype", action="store_true",
                    help="measurement only: cloud-free calls through the one-column-per-wavefront prototype k_n1 (profiles/round2_n1_expf.md)")
    ap.add_argument("--host-cols", type=int, default=131072,
                    help="columns of the end-to-end (host-pointer, PCIe-inclusive) measurement after the timed region; 0 = skip")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # (dmabuf IPC: what RCCL needs on this pool; already exported by the image)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world

    # ---- CPU baseline first: worker processes are spawned, which must happen before this process touches the GPU
    cpu, cpu_sample = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.cpu_baseline import measure
        cpu = measure(nlay=args.nlay, config=args.config, cols_per_core=args.cpu_cols_per_core, sample_cols=0 if args.mcica else 256)
        cpu.pop("wall_s", None)
        cpu_sample = cpu.pop("sample_outputs", None)

    import torch
    import torch.distributed as dist
    from rrtmg_lw_amd import api
    from rrtmg_lw_amd.synth import make_gcm_inputs

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the solver has no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or (args.force_gather and "RANK" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    # (explicit choice of the coefficient file: the line's config.kdata says which one ran)
    api.rrtmg_lw_ini(1004.0, kdata=api.REAL_KDATA if os.path.exists(api.REAL_KDATA) else api.STANDIN_KDATA, device=local_rank)
    if args.n1_prototype:
        api.set_n1_prototype(True)
    if args.batch:
        api.set_batch(args.batch)
    if args.no_overlap:
        api.set_overlap(False)
    if args.overlap:
        api.set_overlap(True)
    if args.cu_partition >= 0:
        api.set_cu_partition(args.cu_partition)

    # contiguous column block of this rank.  Every rank works on exactly `per` = ceil(ncol / world) columns (the synthetic generator is
    # keyed by the global column index, so the last rank simply continues past --ncol when it does not divide): equal blocks for the
    # all-gather, no rank-dependent path anywhere.  `value` counts --ncol columns only.
    from rrtmg_lw_amd.shard import column_block, output_rows, output_views
    per = (args.ncol + world - 1) // world
    col0, ncol = rank * per, per
    nlay = args.nlay
    # inputs straight into HBM, generated in slabs to bound the temporaries
    slab = 131072
    parts = [make_gcm_inputs(min(slab, ncol - s), nlay, args.config, col0=col0 + s, backend="torch", device=dev)
             for s in range(0, ncol, slab)]
    d = dict(parts[0])
    d["ncol"] = ncol
    for k, v in parts[0].items():
        if torch.is_tensor(v):
            if len(parts) == 1:
                continue
            cdim = 1 if k == "taucld" else 0            # taucld is (16, ncol, nlay)
            cat = torch.cat([p[k] for p in parts], dim=cdim)
            nd = cat.dim()
            d[k] = cat.permute(*reversed(range(nd))).contiguous().permute(*reversed(range(nd))) if nd > 1 else cat.contiguous()
    del parts
    idrv = d["idrv"]

    # packed output block (rows = uflx, dflx, uflxc, dflxc [, duflx_dt, duflxc_dt with idrv = 1] of nlay+1 levels, hr, hrc of nlay layers) and
    # the single all-gather that reassembles it over the ranks: rrtmg_lw_amd/shard.py, ShardedStep (the two-rank gloo test runs the same class)
    from rrtmg_lw_amd.shard import ShardedStep
    do_gather = use_dist and not args.no_gather
    sharded = ShardedStep(nlay, idrv, per, world, device=dev, gather=do_gather)
    outs = sharded.outs
    out = outs[0]

    stream = torch.cuda.current_stream().cuda_stream
    alpha = None
    if args.mcica:
        # get_alpha's overlap parameter (src/mcica_subcol_gen_lw.f90:160-166) for idcor = 0, decorr_con = 2500 m, layer depths
        # from the hypsometric equation; computed with torch because it is an INPUT of the timed call
        dz = 29.2717 * d["tlay"] * torch.log(d["plev"][:, :-1] / d["plev"][:, 1:])        # R_d / g = 29.27 m K-1
        a = torch.exp(-0.5 * (dz[:, 1:] + dz[:, :-1]) / 2500.0)
        alpha = torch.cat([torch.zeros_like(a[:, :1]), a], dim=1).t().contiguous().t()
        del dz, a

    def solve(o):
        if args.mcica:
            api.rrtmg_lw_mcica_subcol_device(d, o, 1, 0, alpha=alpha, icld=args.mcica, stream=stream)
        else:
            api.rrtmg_lw_device(d, o, stream=stream)

    def step():
        sharded.step(solve)

    drain = sharded.drain

    if args.check and rank == 0:
        import numpy as np
        from oracle.bindings import Oracle
        sharded.count = 0
        step()
        drain()
        api.check(stream)
        n = min(256, ncol)
        dn = make_gcm_inputs(n, nlay, args.config, col0=col0)
        orc = Oracle(kdata=api.REAL_KDATA if os.path.exists(api.REAL_KDATA) else api.STANDIN_KDATA)
        if args.mcica:
            al = np.asfortranarray(alpha[:n].cpu().numpy())
            sub = orc.mcica_subcol(n, nlay, args.mcica, 1, 0, dn["play"], dn["cldfr"], dn["cicewp"], dn["cliqwp"], dn["reice"],
                                   dn["reliq"], dn["taucld"], al)
            dn.update({k: sub[k] for k in ("cldfmcl", "ciwpmcl", "clwpmcl", "reicmcl", "relqmcl", "taucmcl")})
        ref = orc.rrtmg_lw(n, nlay, args.mcica or dn["icld"], dn["idrv"], dn, mcica=bool(args.mcica))
        dfl = max(float(np.abs(out[k][:, :n].T.cpu().numpy() - ref[k]).max()) for k in ("uflx", "dflx", "uflxc", "dflxc"))
        dhr = max(float(np.abs(out[k][:, :n].T.cpu().numpy() - ref[k]).max()) for k in ("hr", "hrc"))
        print(f"# check vs oracle on {n} columns: max|dflux| = {dfl:.3e} W/m2, max|dhr| = {dhr:.3e} K/d", file=sys.stderr)

    for _ in range(args.warmup):
        step()
    drain()
    api.check(stream)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    buf = ctypes.create_string_buffer(1 << 16)
    barrier()
    api.lib().rrtmg_lw_hip_profile_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()                                  # every all-gather of the timed steps completes inside the timed region
    barrier()
    dt = time.perf_counter() - t0
    api.lib().rrtmg_lw_hip_profile_end(buf, len(buf))
    api.check(stream)
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- end to end through the host-pointer entry (what the Fortran shim calls): pageable host arrays in, H2D staging,
    # kernels, D2H, host arrays out.  Reported beside `value`, never as `value`.
    e2e = None
    if rank == 0 and world == 1 and args.host_cols > 0 and not args.mcica:
        import numpy as np
        nh = min(args.host_cols, args.ncol)

        def host_rate(cfg):
            """columns/s of the host-pointer entry on `cfg` columns: pageable numpy arrays, then every array pinned once
            (rrtmg_lw_hip_host_register); input and output arrays persist across the calls, as a host model's do"""
            dh = make_gcm_inputs(nh, nlay, cfg, col0=col0)
            hidrv = dh["idrv"]
            hbytes = algo_bytes_per_col(nlay, hidrv) * nh

            def timed(out=None, reps=3):
                api.rrtmg_lw_from_dict(dh, out=out)                       # warm-up: staging buffers, host threads
                ts = []
                for _ in range(reps):
                    t1 = time.perf_counter()
                    api.rrtmg_lw_from_dict(dh, out=out)
                    ts.append(time.perf_counter() - t1)
                return sorted(ts)[len(ts) // 2]                           # median of three calls (the host's memory rate moves from call to call)

            hout = api._out_arrays(nh, nlay, hidrv)                      # persistent arrays in both legs (a host model's live for the whole run)
            th = timed(out=hout)
            r = dict(value=round(nh / th, 1), unit="columns/s", columns=nh, ms=round(1e3 * th, 2), host_GBps=round(hbytes / th / 1e9, 2))
            pinned = []
            try:
                pinned = [v for v in list(dh.values()) + list(hout.values()) if isinstance(v, np.ndarray) and v.flags.f_contiguous and v.dtype == np.float64]
                for v in pinned:
                    api.host_register(v)
                tp = timed(out=hout)
                r["pinned"] = dict(value=round(nh / tp, 1), ms=round(1e3 * tp, 2), host_GBps=round(hbytes / tp / 1e9, 2))
                # ... and with the arrays a host model sets once declared static (rrtmg_lw_hip_host_static: well-mixed gases, halocarbons,
                # aerosol optical depths, emissivities): their rows are scanned by the first call only.  (A leg of its own: whatever
                # happens in it, the pinned figure stands and the declarations are withdrawn.)
                static = [dh[k] for k in ("co2vmr", "ch4vmr", "n2ovmr", "o2vmr", "cfc11vmr", "cfc12vmr", "cfc22vmr", "ccl4vmr", "tauaer", "emis")]
                try:
                    for v in static:
                        api.host_static(v)
                    ts = timed(out=hout)
                    r["pinned_static"] = dict(value=round(nh / ts, 1), ms=round(1e3 * ts, 2), host_GBps=round(hbytes / ts / 1e9, 2),
                                              static="co2, ch4, n2o, o2, cfc11, cfc12, cfc22, ccl4 mixing ratios, tauaer, emis")
                except Exception as ex:
                    r["pinned_static"] = dict(error=str(ex)[:120])
                finally:
                    for v in static:
                        try:
                            api.host_changed(v, keep=False)
                        except Exception:
                            pass
            except Exception as ex:          # registration is optional
                r["pinned"] = dict(error=str(ex)[:120])
            finally:
                for v in pinned:
                    try:
                        api.host_unregister(v)
                    except Exception:
                        pass
            return r

        e2e = host_rate(args.config)
        e2e["note"] = ("rrtmg_lw_hip_run_nomcica with host arrays: scan + pack | H2D | kernels | D2H | unpack pipelined over the column batches; "
                       "rows holding one value for a batch's columns (zero aerosol / cloud rows, well-mixed gases) are filled on the device instead of "
                       "copied, taucld travels as its band sum (inflglw >= 1), pageable arrays go through pinned staging packed by host threads; "
                       "host_GBps counts the interface's bytes, not the bytes copied - the entry is bound by the host threads' reads of them")
        if args.config != "aer_idrv":
            # the same with aerosol optical depths in layers 1-12 of every band (192 of the 1152 rows non-zero) and idrv = 1
            e2e["with_aerosol_idrv1"] = host_rate("aer_idrv")

    if rank == 0 and cpu is not None and cpu_sample is not None:
        # parity of the timed GPU outputs with the CPU baseline's own results on the first 256 columns of the workload
        import numpy as np
        n = cpu_sample["uflx"].shape[0]
        o = outs[(sharded.count - 1) & 1]
        g = {k: o[k][:, :n].T.cpu().numpy() for k in cpu_sample}
        cpu["gpu_vs_this_baseline"] = dict(
            columns=n, max_abs_dflux_W_m2=float(max(np.abs(g[k] - cpu_sample[k]).max() for k in ("uflx", "dflx", "uflxc", "dflxc"))),
            max_abs_dhr_K_day=float(max(np.abs(g[k] - cpu_sample[k]).max() for k in ("hr", "hrc"))))

    if rank == 0:
        ms_per_step = 1e3 * dt / args.steps
        value = args.ncol / (dt / args.steps)
        kern = {}
        for line in buf.value.decode().splitlines():
            nm, cnt, tot = line.rsplit(" ", 2)
            kern[nm] = (int(cnt), float(tot))
        bpc = algo_bytes_per_col(nlay, idrv) + (8.0 * nlay if args.mcica else 0.0)     # + alpha
        roof = None
        path = None
        compute = None
        if kern:
            ktot = sum(v[1] for v in kern.values())                 # summed kernel milliseconds of the timed steps (this rank)
            # the path's time: the summed kernel time, or the wall time of the timed region where that is shorter (the sweep launches of
            # the four band groups run on four streams and the per-column kernels on a fifth: their HIP-event times overlap)
            tpath = min(ktot, 1e3 * dt)
            pach = bpc * ncol * args.steps / (tpath * 1e-3) / 1e9
            # dominant kernel of the critical path: k_colprep / k_cloudscan / k_cloudlay run on the auxiliary stream underneath the previous
            # batch's k_layer / sweeps (driver.hip: run_pipelined) and are left out of the choice
            crit = {k: v for k, v in kern.items() if not (k.startswith("k_colprep") or k in ("k_cloudscan", "k_cloudlay"))} or kern
            dom = max(crit, key=lambda k: crit[k][1])
            cnt, tot = kern[dom]
            avg_ms = tot / cnt
            cols_per_launch = ncol * args.steps / cnt      # every kernel is launched once per column batch
            dom_d = dict(kernel=dom, avg_launch_ms=round(avg_ms, 4), launches=cnt, columns_per_launch=cols_per_launch,
                         time_share=round(tot / ktot, 4),
                         # the contract's per-kernel figure (whole-column algorithmic bytes / this kernel's time) overstates a kernel that is
                         # a fraction of the path; kept for continuity with round 1, next to the share-weighted one
                         whole_column_bytes_over_kernel_time_GBps=round(bpc * cols_per_launch / (avg_ms * 1e-3) / 1e9, 3),
                         frac_by_kernel_formula=round(bpc * cols_per_launch / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         traffic=None)
            # PMC-measured HBM bytes (profiles/pmc_traffic.json: FETCH_SIZE x 2 + WRITE_SIZE of separate rocprofv3 passes, one entry per
            # configuration): the PATH's bytes per column x this rank's columns per step, and the dominant kernel's per launch
            key = f"{args.config}_L{nlay}" + (f"_mcica{args.mcica}" if args.mcica else "")
            path_traffic = None
            traffic_stale = None
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(pmc):
                try:
                    ent = json.load(open(pmc)).get(key)
                    if ent:
                        # the PMC pass is a committed file, not part of this run: say so when the kernels have changed since it was taken
                        if ent.get("kernels_sha16") != kernels_sha16():
                            traffic_stale = (f"profiles/pmc_traffic.json[{key}] was collected for kernels {ent.get('kernels_sha16', 'of an unrecorded version')}, "
                                             f"this tree has {kernels_sha16()}: re-run tools/pmc_collect.sh")
                            print("# WARNING: " + traffic_stale, file=sys.stderr)
                        path_traffic = ent["bytes_per_column"] * ncol
                        kb = ent["kernels"].get(dom) or ent["kernels"].get(dom.replace(">", ",false>"))
                        if kb:
                            dom_d["traffic"] = round(kb / ent["columns"] * cols_per_launch, 1)
                            dom_d["traffic_source"] = f"profiles/pmc_traffic.json[{key}]"
                except Exception:
                    pass
            roof = dict(bound="hbm", scope="path: all kernels of a step (sweep family = %.0f %% of it)" %
                        (100.0 * sum(v[1] for k, v in kern.items() if k.startswith("k_sweep")) / ktot),
                        achieved=round(pach, 3), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(pach / HBM_PEAK_GBS, 5),
                        algorithmic_bytes_per_column=bpc, kernels_ms_per_step=round(ktot / args.steps, 3), path_ms_per_step=round(tpath / args.steps, 3),
                        traffic=path_traffic, traffic_scope="HBM bytes of ALL kernels of one step on this rank (PMC, per-column figure of the same configuration x columns)" if path_traffic else None,
                        traffic_over_algorithmic=round(path_traffic / (bpc * ncol), 3) if path_traffic else None, traffic_stale=traffic_stale, dominant=dom_d)
            path = dict(kernels_ms_per_step=round(ktot / args.steps, 3),
                        note="HIP-event time per kernel; the sweeps of the four band groups and the per-column kernels run on their own streams, so the sum exceeds the step",
                        kernels={k: round(v[1] / args.steps, 3) for k, v in sorted(kern.items(), key=lambda kv: -kv[1][1])},
                        families={fam: round(sum(v[1] for k, v in kern.items() if k.startswith(fam)) / args.steps, 3)
                                  for fam in ("k_colprep", "k_subcol", "k_cloud", "k_layer", "k_sweepc", "k_sweepz", "k_flux")})
            # instruction side: counts per column from the committed PMC pass of the same configuration, times from this run
            pc = os.path.join(ROOT, "profiles", "pmc_compute.json")
            if os.path.exists(pc):
                try:
                    ent = json.load(open(pc)).get(key)
                    if ent:
                        # (over the path's time, as `roofline`: the summed HIP-event times overlap since the sweeps run on four streams)
                        valu = ent["valu_wave_instr_per_column"] * 64.0 * ncol * args.steps / (tpath * 1e-3)
                        ldsb = ent["lds_bytes_per_column"] * ncol * args.steps / (tpath * 1e-3)
                        compute = dict(valu_lane_instr_per_s=round(valu, 1), valu_peak=VALU_PEAK_LANE_INSTR, valu_frac=round(valu / VALU_PEAK_LANE_INSTR, 4),
                                       valu_frac_of_f64_rate=round(valu / (VALU_PEAK_LANE_INSTR / 2), 4),
                                       lds_bytes_per_s=round(ldsb, 1), lds_peak=LDS_PEAK_BYTES, lds_frac=round(ldsb / LDS_PEAK_BYTES, 4),
                                       valu_wave_instr_per_column=ent["valu_wave_instr_per_column"], lds_bytes_per_column=ent["lds_bytes_per_column"],
                                       source=ent.get("source", "profiles/pmc_compute.json"))
                except Exception:
                    pass
        res = dict(metric="columns/sec, 72-layer profiles" if nlay == 72 else f"columns/sec, {nlay}-layer profiles",
                   value=round(value, 1), unit="columns/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=round(ms_per_step, 3), higher_is_better=True, scaling="strong", vs_baseline=None,
                   dtype="f64 (transmittance / tfn codes and their LDS table f32)", data="synthetic",
                   config=dict(workload=f"{args.ncol} synthetic {nlay}-layer columns, config '{args.config}' "
                                        + (f"(McICA: kissvec sub-column generator icld={args.mcica} + cldprmc + rtrnmc, idrv={idrv}), " if args.mcica else
                                           f"(icld={d['icld']}: {'rtrnmr max-random overlap' if d['icld'] == 2 else 'clear'}, idrv={idrv}), ") +
                                        f"sharded {world}x{per}",
                               ncol_total=args.ncol, nlay=nlay, columns_per_gpu=per, parallelism=f"columns/{world}",
                               gather="rccl all_gather_into_tensor of the packed outputs, overlapped with the next step's kernels" if do_gather else "none",
                               kdata="stand-in (real k-data absent from the reference mount)" if api.kdata_is_standin() else "real",
                               sweeps="k_n1 prototype (measurement)" if args.n1_prototype else "production"),
                   workspace_GB=round(api.workspace_bytes() / 1e9, 3),
                   roofline=roof, compute=compute, path=path, cpu_baseline=cpu, end_to_end=e2e)
        print(json.dumps(res))
    if use_dist:
        dist.destroy_process_group()
    api.finalize()


if __name__ == "__main__":
    main()
