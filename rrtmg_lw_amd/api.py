"""Python host-side mirror of the reference's GCM interface, over the C ABI of librrtmg_lw_hip.so.

    rrtmg_lw_ini(cpdair)                         reference: src/rrtmg_lw_init.f90:47
    rrtmg_lw(ncol, nlay, icld, idrv, play, ...)  reference: src/rrtmg_lw_rad.nomcica.f90:99-108

Same names, argument order and meaning as the Fortran; arrays are numpy float64 in Fortran order (or anything
convertible).  Outputs are returned in a dict instead of being written into caller arrays.  There is no CPU
path: importing works anywhere, but every call raises unless the HIP library is built and a GPU is present.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RRTMG_LW_HIP_LIB", os.path.join(_HERE, "librrtmg_lw_hip.so"))   # env override: tuning builds
STATIC_BLOB = os.path.join(_HERE, "data", "lw_static.bin")
STANDIN_KDATA = os.path.join(_HERE, "data", "standin.kdata.bin")
REAL_KDATA = os.path.join(os.path.dirname(_HERE), "data", "rrtmg_lw.kdata.bin")
NBND, NGPT = 16, 140

LIB_PATH_G256 = os.path.join(_HERE, "librrtmg_lw_hip_g256.so")      # the 256-g-point build (every band keeps its 16 original g-points)

_dp = C.POINTER(C.c_double)
_lib = None
_libs = {}
_gpoints = 140
_initialised = False


class RrtmgLwError(RuntimeError):
    pass


def lib():
    """Load the selected library - librrtmg_lw_hip.so, or librrtmg_lw_hip_g256.so after select_gpoints(256) - and fail loudly when it has
    not been built.  A process that also uses PyTorch must import torch first (one HIP runtime per process: INTEGRATION.md 2)."""
    global _lib
    if _lib is None:
        path = LIB_PATH if _gpoints == 140 else LIB_PATH_G256
        if path not in _libs:
            if not os.path.exists(path):
                raise RrtmgLwError(f"{path} not found - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                                   "(hipcc --offload-arch=gfx950); there is no CPU fallback")
            h = C.CDLL(path)
            h.rrtmg_lw_hip_last_error.restype = C.c_char_p
            h.rrtmg_lw_hip_workspace_bytes.restype = C.c_longlong
            _libs[path] = h
        _lib = _libs[path]
    return _lib


def select_gpoints(n):
    """Choose the g-point model of the calls that follow: 140 (the reference's shipped model) or 256 (every band keeps its 16 original
    g-points: the accuracy mode of modules/parrrtm.f90:40-41,77-110; its McICA arrays hold 256 sub-columns).  Each is its own library with its own
    state: call rrtmg_lw_ini after switching."""
    global _lib, _gpoints
    if n not in (140, 256):
        raise ValueError("gpoints must be 140 or 256")
    _gpoints, _lib = n, None


def gpoints():
    return int(lib().rrtmg_lw_hip_gpoints())


def _check(rc):
    if rc != 0:
        raise RrtmgLwError(f"rrtmg_lw_hip error {rc}: {lib().rrtmg_lw_hip_last_error().decode()}")


def default_kdata():
    """Real absorption coefficients if they have been converted into data/rrtmg_lw.kdata.bin
    (rrtmg_lw_amd/kdata.py); otherwise the synthetic stand-in tables - with a warning, because every flux computed
    from them is non-physical (right shapes and magnitudes only).  Pass kdata=STANDIN_KDATA explicitly (tests, bench)
    or set RRTMG_LW_ALLOW_STANDIN=1 to silence it."""
    if os.path.exists(REAL_KDATA):
        return REAL_KDATA
    if os.environ.get("RRTMG_LW_ALLOW_STANDIN", "") != "1":
        import warnings
        warnings.warn(f"{REAL_KDATA} not found: rrtmg_lw_ini falls back to the STAND-IN absorption coefficients - fluxes and "
                      "heating rates will be non-physical.  Convert the AER data with `python -m rrtmg_lw_amd.kdata "
                      "<rrtmg_lw_k_g.f90 | rrtmg_lw.nc> data/rrtmg_lw.kdata.bin`.", RuntimeWarning, stacklevel=3)
    return STANDIN_KDATA


def rrtmg_lw_ini(cpdair=1004.0, kdata=None, device=None, static=STATIC_BLOB):
    """rrtmg_lw_ini(cpdair): one-time table setup on this process's GPU (reference src/rrtmg_lw_init.f90:47).
    kdata=None: the converted real coefficients, or (with a RuntimeWarning) the stand-in tables; kdata_is_standin() tells."""
    global _initialised
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    kdata = kdata or default_kdata()
    _check(lib().rrtmg_lw_hip_init(static.encode(), kdata.encode(), C.c_double(cpdair), C.c_int(device)))
    _initialised = True


def kdata_is_standin():
    return lib().rrtmg_lw_hip_kdata_is_standin() == 1


def set_batch(n):
    _check(lib().rrtmg_lw_hip_set_batch(C.c_int(int(n))))


def host_register(a):
    """Pin a numpy array that will be passed to the host-pointer entries repeatedly (rrtmg_lw_hip_host_register)."""
    _check(lib().rrtmg_lw_hip_host_register(C.c_void_p(a.ctypes.data), C.c_longlong(a.nbytes)))


def host_is_registered(a):
    """whether the host-pointer entries would copy the numpy array `a` from where it lies (it is inside a range pinned by host_register)"""
    return bool(lib().rrtmg_lw_hip_host_is_registered(C.c_void_p(a.ctypes.data), C.c_longlong(a.nbytes)))


def host_unregister(a):
    _check(lib().rrtmg_lw_hip_host_unregister(C.c_void_p(a.ctypes.data)))


def host_static(a):
    """Declare the numpy array `a` static: the host-pointer entries scan its rows once instead of on every call (rrtmg_lw_hip_host_static);
    call host_changed(a) after modifying it."""
    _check(lib().rrtmg_lw_hip_host_static(C.c_void_p(a.ctypes.data), C.c_longlong(a.nbytes)))


def host_changed(a, keep=True):
    """the static array `a` has new contents (keep=False: the declaration is withdrawn as well)"""
    _check(lib().rrtmg_lw_hip_host_changed(C.c_void_p(a.ctypes.data), C.c_int(1 if keep else 0)))


def combine_stats():
    """(calls, device passes) of the combining entry for concurrent small calls since the library was loaded"""
    c, p = C.c_longlong(0), C.c_longlong(0)
    lib().rrtmg_lw_hip_combine_stats.restype = None
    lib().rrtmg_lw_hip_combine_stats(C.byref(c), C.byref(p))
    return int(c.value), int(p.value)


def set_n1_prototype(on):
    """Measurement only: cloud-free calls through the prototype of the one-column-per-wavefront mapping (k_n1)."""
    _check(lib().rrtmg_lw_hip_set_n1_prototype(C.c_int(1 if on else 0)))


def init_devices(devices, cpdair=1004.0, kdata=None, static=STATIC_BLOB):
    """rrtmg_lw_ini for several GPUs driven by this one process (rrtmg_lw_hip_init_devices): the host-pointer entries split their
    columns over `devices` (HIP ordinals; an ordinal may repeat: virtual devices on one GPU)."""
    global _initialised
    kdata = kdata or default_kdata()
    arr = (C.c_int * len(devices))(*[int(v) for v in devices])
    _check(lib().rrtmg_lw_hip_init_devices(static.encode(), kdata.encode(), C.c_double(cpdair), C.c_int(len(devices)), arr))
    _initialised = True


def last_device_state():
    """index of the library state (rrtmg_lw_hip_init_devices) this thread's last device-pointer entry ran on"""
    return int(lib().rrtmg_lw_hip_last_device_state())


def num_devices():
    return int(lib().rrtmg_lw_hip_num_devices())


def set_overlap(on):
    _check(lib().rrtmg_lw_hip_set_overlap(C.c_int(1 if on else 0)))


def set_cu_partition(layer_cus):
    """k_layer of batch i+1 on `layer_cus` CUs beside the sweeps of batch i on the others (rrtmg_lw_hip_set_cu_partition); 0 = off"""
    _check(lib().rrtmg_lw_hip_set_cu_partition(C.c_int(int(layer_cus))))


def workspace_bytes():
    """device memory the library holds right now (rrtmg_lw_hip_workspace_bytes)"""
    lib().rrtmg_lw_hip_workspace_bytes.restype = C.c_longlong
    return int(lib().rrtmg_lw_hip_workspace_bytes())


def set_one_sweep_max(ncol):
    """cloudy batches of up to `ncol` columns take one sweep launch per band group instead of three (0 = never); returns the previous value"""
    return int(lib().rrtmg_lw_hip_set_one_sweep_max(C.c_int(int(ncol))))


def set_layer_split(on):
    """k_layer's bands of a (window, layer) over several workgroups where the batch does not fill the chip (rrtmg_lw_hip_set_layer_split); returns the previous value"""
    return int(lib().rrtmg_lw_hip_set_layer_split(C.c_int(1 if on else 0)))


def set_split_max(ncol):
    """batches of up to `ncol` columns are swept one band per workgroup (rrtmg_lw_hip_set_split_max; 0 = never); returns the previous value"""
    return int(lib().rrtmg_lw_hip_set_split_max(C.c_int(int(ncol))))


def set_graph_max(ncol):
    """device-resident one-batch calls of up to `ncol` columns are replayed as one graph (rrtmg_lw_hip_set_graph_max; 0 = never); returns the previous value"""
    return int(lib().rrtmg_lw_hip_set_graph_max(C.c_int(int(ncol))))


def graph_stats():
    """(graphs captured, calls replayed from a graph) since initialisation"""
    a, b = C.c_longlong(0), C.c_longlong(0)
    lib().rrtmg_lw_hip_graph_stats(C.byref(a), C.byref(b))
    return int(a.value), int(b.value)


def set_wide_window(on):
    """k_layer's second pass with the wide staging window for workgroups whose columns lie more than one reference-pressure plane apart
    (rrtmg_lw_hip_set_wide_window); returns the previous on / off"""
    return int(lib().rrtmg_lw_hip_set_wide_window(C.c_int(1 if on else 0)))


def set_column_sort(on, min_gain=-1):
    """columns of a cloudy batch (rtrn, rtrnmr, McICA) are taken by cloud top within windows of 256 where that removes >= min_gain block-levels from the
    cloud zone (rrtmg_lw_hip_set_column_sort; min_gain < 0 keeps the threshold); returns the previous on / off"""
    return int(lib().rrtmg_lw_hip_set_column_sort(C.c_int(1 if on else 0), C.c_int(int(min_gain))))


def column_sort_min():
    """the threshold of set_column_sort in force"""
    return int(lib().rrtmg_lw_hip_column_sort_min())


def cu_partition():
    return int(lib().rrtmg_lw_hip_cu_partition())


def _f(a, shape):
    a = np.asfortranarray(a, dtype=np.float64)
    if tuple(a.shape) != tuple(shape):
        raise ValueError(f"array has shape {a.shape}, interface declares {shape}")
    return a


def _p(a):
    return a.ctypes.data_as(_dp)


def _out_ok(a, shape, name):
    """an array the library writes into in place: float64, Fortran-contiguous, writeable, exactly the interface's shape"""
    if not (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags.f_contiguous and a.flags.writeable and tuple(a.shape) == tuple(shape)):
        raise ValueError(f"output array {name!r} must be a writeable float64 Fortran-ordered array of shape {tuple(shape)}")
    return a


def rrtmg_lw(ncol, nlay, icld, idrv, play, plev, tlay, tlev, tsfc, h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr,
             cfc11vmr, cfc12vmr, cfc22vmr, ccl4vmr, emis, inflglw, iceflglw, liqflglw, cldfr, taucld, cicewp,
             cliqwp, reice, reliq, tauaer, out=None):
    """Non-McICA rrtmg_lw with host arrays.  Returns dict(uflx, dflx, hr, uflxc, dflxc, hrc[, duflx_dt, duflxc_dt], icld);
    `out` may hold preallocated (e.g. host_register'ed) Fortran-ordered output arrays."""
    a2 = [_f(x, (ncol, nlay)) for x in (play,)] + [_f(plev, (ncol, nlay + 1)), _f(tlay, (ncol, nlay)),
                                                  _f(tlev, (ncol, nlay + 1)), _f(tsfc, (ncol,))]
    gases = [_f(x, (ncol, nlay)) for x in (h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, cfc11vmr, cfc12vmr, cfc22vmr, ccl4vmr)]
    em = _f(emis, (ncol, NBND))
    cld = [_f(cldfr, (ncol, nlay)), _f(taucld, (NBND, ncol, nlay)), _f(cicewp, (ncol, nlay)), _f(cliqwp, (ncol, nlay)),
           _f(reice, (ncol, nlay)), _f(reliq, (ncol, nlay)), _f(tauaer, (ncol, nlay, NBND))]
    if out is None:
        out = _out_arrays(ncol, nlay, idrv)
    else:
        for k in ("uflx", "dflx", "uflxc", "dflxc") + (("duflx_dt", "duflxc_dt") if idrv == 1 else ()):
            _out_ok(out[k], (ncol, nlay + 1), k)
        for k in ("hr", "hrc"):
            _out_ok(out[k], (ncol, nlay), k)
    icld_c = C.c_int(int(icld))
    null = C.cast(None, _dp)
    args = [C.c_int(ncol), C.c_int(nlay), C.byref(icld_c), C.c_int(int(idrv))]
    args += [_p(x) for x in a2] + [_p(x) for x in gases] + [_p(em)]
    args += [C.c_int(int(inflglw)), C.c_int(int(iceflglw)), C.c_int(int(liqflglw))]
    args += [_p(x) for x in cld]
    args += [_p(out[k]) for k in ("uflx", "dflx", "hr", "uflxc", "dflxc", "hrc")]
    args += [_p(out["duflx_dt"]) if idrv == 1 else null, _p(out["duflxc_dt"]) if idrv == 1 else null]
    _check(lib().rrtmg_lw_hip_run_nomcica(*args))
    out["icld"] = icld_c.value
    return out


_GCM_ORDER = ("play", "plev", "tlay", "tlev", "tsfc", "h2ovmr", "o3vmr", "co2vmr", "ch4vmr", "n2ovmr", "o2vmr",
              "cfc11vmr", "cfc12vmr", "cfc22vmr", "ccl4vmr", "emis")
_CLD_ORDER = ("cldfr", "taucld", "cicewp", "cliqwp", "reice", "reliq", "tauaer")


def rrtmg_lw_from_dict(d, icld=None, idrv=None, out=None):
    """Convenience: call rrtmg_lw with the dictionaries produced by rrtmg_lw_amd.synth.make_gcm_inputs."""
    icld = d["icld"] if icld is None else icld
    idrv = d["idrv"] if idrv is None else idrv
    return rrtmg_lw(d["ncol"], d["nlay"], icld, idrv, *[d[k] for k in _GCM_ORDER], d["inflglw"], d["iceflglw"],
                    d["liqflglw"], *[d[k] for k in _CLD_ORDER], out=out)


def rrtmg_lw_device(d, out, icld=None, idrv=None, stream=None):
    """Device-resident call: `d` holds torch CUDA tensors laid out column-fastest (synth.make_gcm_inputs(backend="torch")),
    `out` a dict of preallocated output tensors (uflx, dflx, hr, uflxc, dflxc, hrc, duflx_dt, duflxc_dt).
    Enqueues on `stream` (an integer hipStream_t handle, e.g. torch.cuda.current_stream().cuda_stream) and returns
    immediately; call check(stream) to synchronise and collect physics errors."""
    icld = d["icld"] if icld is None else icld
    idrv = d["idrv"] if idrv is None else idrv
    icld_c = C.c_int(int(icld))
    ptr = lambda t: C.c_void_p(t.data_ptr())
    args = [C.c_int(d["ncol"]), C.c_int(d["nlay"]), C.byref(icld_c), C.c_int(int(idrv))]
    args += [ptr(d[k]) for k in _GCM_ORDER]
    args += [C.c_int(int(d["inflglw"])), C.c_int(int(d["iceflglw"])), C.c_int(int(d["liqflglw"]))]
    args += [ptr(d[k]) for k in _CLD_ORDER]
    args += [ptr(out[k]) for k in ("uflx", "dflx", "hr", "uflxc", "dflxc", "hrc")]
    args += [ptr(out[k]) if out.get(k) is not None else C.c_void_p(0) for k in ("duflx_dt", "duflxc_dt")]      # needed only with idrv = 1
    args.append(C.c_void_p(stream or 0))
    _check(lib().rrtmg_lw_hip_run_nomcica_device(*args))
    return icld_c.value


def mcica_subcol_device(d, sub, icld, permuteseed, irng, alpha=None, stream=None):
    """Device-resident mcica_subcol_lw: `d` as for rrtmg_lw_device (play, cldfr, cicewp, cliqwp, reice, reliq, taucld), `sub` a dict of
    preallocated torch tensors cldfmcl, ciwpmcl, clwpmcl, taucmcl (g fastest: shape (nlay, ncol, 140) contiguous) and reicmcl, relqmcl."""
    irng_c = C.c_int(int(irng))
    ptr = lambda t: C.c_void_p(t.data_ptr())
    args = [C.c_int(d["ncol"]), C.c_int(d["nlay"]), C.c_int(int(icld)), C.c_int(int(permuteseed)), C.byref(irng_c)]
    args += [ptr(d[k]) for k in ("play", "cldfr", "cicewp", "cliqwp", "reice", "reliq", "taucld")]
    args += [ptr(alpha) if alpha is not None else C.c_void_p(0)]
    args += [ptr(sub[k]) for k in ("cldfmcl", "ciwpmcl", "clwpmcl", "reicmcl", "relqmcl", "taucmcl")]
    args.append(C.c_void_p(stream or 0))
    _check(lib().rrtmg_lw_hip_mcica_subcol_device(*args))
    return irng_c.value


def rrtmg_lw_mcica_device(d, sub, out, icld=None, idrv=None, stream=None):
    """Device-resident McICA rrtmg_lw with explicit sub-column arrays (the reference's argument list, src/rrtmg_lw_rad.f90:99-108)."""
    icld = d["icld"] if icld is None else icld
    idrv = d["idrv"] if idrv is None else idrv
    icld_c = C.c_int(int(icld))
    ptr = lambda t: C.c_void_p(t.data_ptr())
    args = [C.c_int(d["ncol"]), C.c_int(d["nlay"]), C.byref(icld_c), C.c_int(int(idrv))]
    args += [ptr(d[k]) for k in _GCM_ORDER]
    args += [C.c_int(int(d["inflglw"])), C.c_int(int(d["iceflglw"])), C.c_int(int(d["liqflglw"]))]
    args += [ptr(sub[k]) for k in ("cldfmcl", "taucmcl", "ciwpmcl", "clwpmcl", "reicmcl", "relqmcl")] + [ptr(d["tauaer"])]
    args += [ptr(out[k]) for k in ("uflx", "dflx", "hr", "uflxc", "dflxc", "hrc")]
    args += [ptr(out[k]) if out.get(k) is not None else C.c_void_p(0) for k in ("duflx_dt", "duflxc_dt")]      # needed only with idrv = 1
    args.append(C.c_void_p(stream or 0))
    _check(lib().rrtmg_lw_hip_run_mcica_device(*args))
    return icld_c.value


def check(stream=None):
    _check(lib().rrtmg_lw_hip_check(C.c_void_p(stream or 0)))


def run_columns(cols, istart=1, iend=16, icld=None, idrv=None):
    """Prepared-column entry for a list of column dicts (rrtmg_lw_amd.io_rrtm.read_input_rrtm) that share nlayers
    and cloud flags.  Returns dict of arrays (ncol, 0:nlayers)."""
    n = len(cols)
    nl = int(cols[0]["nlayers"])
    icld = int(cols[0]["icld"]) if icld is None else icld
    idrv = int(cols[0]["idrv"]) if idrv is None else idrv
    st = lambda key, shape: _f(np.stack([np.asarray(c[key], dtype=np.float64).reshape(shape[1:], order="F") for c in cols]), shape)
    a = dict(pavel=st("pavel", (n, nl)), tavel=st("tavel", (n, nl)), pz=st("pz", (n, nl + 1)), tz=st("tz", (n, nl + 1)),
             tbound=_f(np.array([float(c["tbound"]) for c in cols]), (n,)), semiss=st("semiss", (n, NBND)),
             coldry=st("coldry", (n, nl)), wkl=st("wkl", (n, 7, nl)), wbrodl=st("wbrodl", (n, nl)), wx=st("wx", (n, 4, nl)),
             pwvcm=_f(np.array([float(c["pwvcm"]) for c in cols]), (n,)), cldfrac=st("cldfrac", (n, nl)),
             tauc=st("tauc", (n, NBND, nl)), ciwp=st("ciwp", (n, nl)), clwp=st("clwp", (n, nl)), rei=st("rei", (n, nl)),
             rel=st("rel", (n, nl)), taua=st("tauaer", (n, nl, NBND)))
    names = ("totuflux", "totdflux", "fnet", "htr", "totuclfl", "totdclfl", "fnetc", "htrc", "dtotuflux_dt", "dtotuclfl_dt")
    out = {k: np.zeros((n, nl + 1), order="F") for k in names}
    c0 = cols[0]
    args = [C.c_int(n), C.c_int(nl), C.c_int(istart), C.c_int(iend), C.c_int(icld), C.c_int(idrv)]
    args += [_p(a[k]) for k in ("pavel", "tavel", "pz", "tz", "tbound", "semiss", "coldry", "wkl", "wbrodl", "wx", "pwvcm")]
    args += [C.c_int(int(c0["inflag"])), C.c_int(int(c0["iceflag"])), C.c_int(int(c0["liqflag"]))]
    args += [_p(a[k]) for k in ("cldfrac", "tauc", "ciwp", "clwp", "rei", "rel", "taua")]
    args += [_p(out[k]) for k in names]
    _check(lib().rrtmg_lw_hip_run_columns(*args))
    return out


# ---------------------------------------------------------------------------------------------------
# McICA flavour: reference src/rrtmg_lw_rad.f90:99-108, src/mcica_subcol_gen_lw.f90:68,183
# ---------------------------------------------------------------------------------------------------
def _out_arrays(ncol, nlay, idrv):
    out = {k: np.empty((ncol, nlay + 1), order="F") for k in ("uflx", "dflx", "uflxc", "dflxc")}
    out["hr"] = np.empty((ncol, nlay), order="F")
    out["hrc"] = np.empty((ncol, nlay), order="F")
    if idrv == 1:
        out["duflx_dt"] = np.empty((ncol, nlay + 1), order="F")
        out["duflxc_dt"] = np.empty((ncol, nlay + 1), order="F")
    return out


def _out_ptrs(out, idrv):
    null = C.cast(None, _dp)
    return [_p(out[k]) for k in ("uflx", "dflx", "hr", "uflxc", "dflxc", "hrc")] + \
           [_p(out["duflx_dt"]) if idrv == 1 else null, _p(out["duflxc_dt"]) if idrv == 1 else null]


def _gcm_arrays(ncol, nlay, play, plev, tlay, tlev, tsfc, gases, emis):
    return [_f(play, (ncol, nlay)), _f(plev, (ncol, nlay + 1)), _f(tlay, (ncol, nlay)), _f(tlev, (ncol, nlay + 1)),
            _f(tsfc, (ncol,))] + [_f(x, (ncol, nlay)) for x in gases] + [_f(emis, (ncol, NBND))]


def rrtmg_lw_mcica(ncol, nlay, icld, idrv, play, plev, tlay, tlev, tsfc, h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr,
                   cfc11vmr, cfc12vmr, cfc22vmr, ccl4vmr, emis, inflglw, iceflglw, liqflglw, cldfmcl, taucmcl, ciwpmcl,
                   clwpmcl, reicmcl, relqmcl, tauaer):
    """McICA rrtmg_lw (src/rrtmg_lw_rad.f90:99-108) with host arrays; sub-column arrays are (140, ncol, nlay)."""
    a = _gcm_arrays(ncol, nlay, play, plev, tlay, tlev, tsfc,
                    (h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, cfc11vmr, cfc12vmr, cfc22vmr, ccl4vmr), emis)
    ng = gpoints()
    cld = [_f(cldfmcl, (ng, ncol, nlay)), _f(taucmcl, (ng, ncol, nlay)), _f(ciwpmcl, (ng, ncol, nlay)),
           _f(clwpmcl, (ng, ncol, nlay)), _f(reicmcl, (ncol, nlay)), _f(relqmcl, (ncol, nlay)), _f(tauaer, (ncol, nlay, NBND))]
    out = _out_arrays(ncol, nlay, idrv)
    icld_c = C.c_int(int(icld))
    args = [C.c_int(ncol), C.c_int(nlay), C.byref(icld_c), C.c_int(int(idrv))] + [_p(x) for x in a]
    args += [C.c_int(int(inflglw)), C.c_int(int(iceflglw)), C.c_int(int(liqflglw))] + [_p(x) for x in cld] + _out_ptrs(out, idrv)
    _check(lib().rrtmg_lw_hip_run_mcica(*args))
    out["icld"] = icld_c.value
    return out


_MC_ORDER = ("cldfmcl", "taucmcl", "ciwpmcl", "clwpmcl", "reicmcl", "relqmcl", "tauaer")


def rrtmg_lw_mcica_from_dict(d, icld=None, idrv=None):
    icld = d["icld"] if icld is None else icld
    idrv = d["idrv"] if idrv is None else idrv
    return rrtmg_lw_mcica(d["ncol"], d["nlay"], icld, idrv, *[d[k] for k in _GCM_ORDER], d["inflglw"], d["iceflglw"],
                          d["liqflglw"], *[d[k] for k in _MC_ORDER])


def get_alpha(ncol, nlay, icld, idcor, decorr_con, dz, lat, juldat, cldfrac):
    """get_alpha (src/mcica_subcol_gen_lw.f90:68): decorrelation-length overlap parameter, (ncol, nlay)."""
    alpha = np.zeros((ncol, nlay), order="F")
    _check(lib().rrtmg_lw_hip_get_alpha(C.c_int(ncol), C.c_int(nlay), C.c_int(int(icld)), C.c_int(int(idcor)),
                                        C.c_double(float(decorr_con)), _p(_f(dz, (ncol, nlay))), _p(_f(lat, (ncol,))),
                                        C.c_int(int(juldat)), _p(_f(cldfrac, (ncol, nlay))), _p(alpha)))
    return alpha


def mcica_subcol_lw(ncol, nlay, icld, permuteseed, irng, play, cldfrac, ciwp, clwp, rei, rel, tauc, alpha=None, out=None):
    """mcica_subcol_lw (src/mcica_subcol_gen_lw.f90:183-185; `iplon` dropped: every column is generated).
    Returns dict(cldfmcl, ciwpmcl, clwpmcl, taucmcl (ngpt,ncol,nlay), reicmcl, relqmcl (ncol,nlay), irng); `out`: such a dict of
    preallocated Fortran-ordered arrays to fill (a host model's sub-column arrays persist)."""
    ng = gpoints()
    z3 = lambda: np.zeros((ng, ncol, nlay), order="F")
    z2 = lambda: np.zeros((ncol, nlay), order="F")
    o = out if out is not None else dict(cldfmcl=z3(), ciwpmcl=z3(), clwpmcl=z3(), reicmcl=z2(), relqmcl=z2(), taucmcl=z3())
    # the C entry writes through these pointers: every output must BE a float64 Fortran-ordered array of the declared shape (_f would
    # silently hand a converted copy to nobody)
    for k, shape in (("cldfmcl", (ng, ncol, nlay)), ("ciwpmcl", (ng, ncol, nlay)), ("clwpmcl", (ng, ncol, nlay)), ("taucmcl", (ng, ncol, nlay)),
                     ("reicmcl", (ncol, nlay)), ("relqmcl", (ncol, nlay))):
        _out_ok(o[k], shape, k)
    irng_c = C.c_int(int(irng))
    ins = [_f(play, (ncol, nlay)), _f(cldfrac, (ncol, nlay)), _f(ciwp, (ncol, nlay)), _f(clwp, (ncol, nlay)),
           _f(rei, (ncol, nlay)), _f(rel, (ncol, nlay)), _f(tauc, (NBND, ncol, nlay))]
    al = _p(_f(alpha, (ncol, nlay))) if alpha is not None else C.cast(None, _dp)
    _check(lib().rrtmg_lw_hip_mcica_subcol(C.c_int(ncol), C.c_int(nlay), C.c_int(int(icld)), C.c_int(int(permuteseed)),
                                           C.byref(irng_c), *[_p(x) for x in ins], al,
                                           *[_p(o[k]) for k in ("cldfmcl", "ciwpmcl", "clwpmcl", "reicmcl", "relqmcl", "taucmcl")]))
    o["irng"] = irng_c.value
    return o


def rrtmg_lw_mcica_subcol(ncol, nlay, icld, idrv, permuteseed, irng, play, plev, tlay, tlev, tsfc, h2ovmr, o3vmr, co2vmr,
                          ch4vmr, n2ovmr, o2vmr, cfc11vmr, cfc12vmr, cfc22vmr, ccl4vmr, emis, inflglw, iceflglw, liqflglw,
                          cldfr, taucld, cicewp, cliqwp, reice, reliq, alpha, tauaer):
    """mcica_subcol_lw followed by the McICA rrtmg_lw in one call; the sub-columns stay on the device as bit masks."""
    a = _gcm_arrays(ncol, nlay, play, plev, tlay, tlev, tsfc,
                    (h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, cfc11vmr, cfc12vmr, cfc22vmr, ccl4vmr), emis)
    cld = [_p(_f(cldfr, (ncol, nlay))), _p(_f(taucld, (NBND, ncol, nlay))), _p(_f(cicewp, (ncol, nlay))), _p(_f(cliqwp, (ncol, nlay))),
           _p(_f(reice, (ncol, nlay))), _p(_f(reliq, (ncol, nlay))),
           _p(_f(alpha, (ncol, nlay))) if alpha is not None else C.cast(None, _dp), _p(_f(tauaer, (ncol, nlay, NBND)))]
    out = _out_arrays(ncol, nlay, idrv)
    icld_c, irng_c = C.c_int(int(icld)), C.c_int(int(irng))
    args = [C.c_int(ncol), C.c_int(nlay), C.byref(icld_c), C.c_int(int(idrv)), C.c_int(int(permuteseed)), C.byref(irng_c)]
    args += [_p(x) for x in a] + [C.c_int(int(inflglw)), C.c_int(int(iceflglw)), C.c_int(int(liqflglw))] + cld + _out_ptrs(out, idrv)
    _check(lib().rrtmg_lw_hip_run_mcica_subcol(*args))
    out["icld"] = icld_c.value
    out["irng"] = irng_c.value
    return out


def rrtmg_lw_mcica_subcol_from_dict(d, permuteseed, irng, alpha=None, icld=None, idrv=None):
    icld = d["icld"] if icld is None else icld
    idrv = d["idrv"] if idrv is None else idrv
    return rrtmg_lw_mcica_subcol(d["ncol"], d["nlay"], icld, idrv, permuteseed, irng, *[d[k] for k in _GCM_ORDER], d["inflglw"],
                                 d["iceflglw"], d["liqflglw"], d["cldfr"], d["taucld"], d["cicewp"], d["cliqwp"], d["reice"],
                                 d["reliq"], alpha, d["tauaer"])


def rrtmg_lw_mcica_subcol_device(d, out, permuteseed, irng, alpha=None, icld=None, idrv=None, stream=None):
    """Device-resident fused generator + solver (torch CUDA tensors, column-fastest); see rrtmg_lw_device."""
    icld = d["icld"] if icld is None else icld
    idrv = d["idrv"] if idrv is None else idrv
    icld_c, irng_c = C.c_int(int(icld)), C.c_int(int(irng))
    ptr = lambda t: C.c_void_p(t.data_ptr())
    args = [C.c_int(d["ncol"]), C.c_int(d["nlay"]), C.byref(icld_c), C.c_int(int(idrv)), C.c_int(int(permuteseed)), C.byref(irng_c)]
    args += [ptr(d[k]) for k in _GCM_ORDER]
    args += [C.c_int(int(d["inflglw"])), C.c_int(int(d["iceflglw"])), C.c_int(int(d["liqflglw"]))]
    args += [ptr(d[k]) for k in ("cldfr", "taucld", "cicewp", "cliqwp", "reice", "reliq")]
    args += [ptr(alpha) if alpha is not None else C.c_void_p(0), ptr(d["tauaer"])]
    args += [ptr(out[k]) for k in ("uflx", "dflx", "hr", "uflxc", "dflxc", "hrc")]
    args += [ptr(out[k]) if out.get(k) is not None else C.c_void_p(0) for k in ("duflx_dt", "duflxc_dt")]      # needed only with idrv = 1
    args.append(C.c_void_p(stream or 0))
    _check(lib().rrtmg_lw_hip_run_mcica_subcol_device(*args))
    return icld_c.value


def run_columns_mcica(cols, subs, istart=1, iend=16, icld=None, idrv=None):
    """Prepared-column McICA entry: cols[k] (rrtmg_lw_amd.io_rrtm.read_input_rrtm dicts) with sub-columns subs[k] =
    dict(cldfmc, taucmc, ciwpmc, clwpmc (140, nlayers), reicmc, relqmc (nlayers)); one (column, sample) pair per entry.
    Returns dict of arrays (n, 0:nlayers)."""
    n = len(cols)
    nl = int(cols[0]["nlayers"])
    icld = int(cols[0]["icld"]) if icld is None else icld
    idrv = int(cols[0]["idrv"]) if idrv is None else idrv
    st = lambda src, key, shape: _f(np.stack([np.asarray(c[key], dtype=np.float64).reshape(shape[1:], order="F") for c in src]), shape)
    a = dict(pavel=st(cols, "pavel", (n, nl)), tavel=st(cols, "tavel", (n, nl)), pz=st(cols, "pz", (n, nl + 1)), tz=st(cols, "tz", (n, nl + 1)),
             tbound=_f(np.array([float(c["tbound"]) for c in cols]), (n,)), semiss=st(cols, "semiss", (n, NBND)),
             coldry=st(cols, "coldry", (n, nl)), wkl=st(cols, "wkl", (n, 7, nl)), wbrodl=st(cols, "wbrodl", (n, nl)), wx=st(cols, "wx", (n, 4, nl)),
             pwvcm=_f(np.array([float(c["pwvcm"]) for c in cols]), (n,)), reicmc=st(subs, "reicmc", (n, nl)), relqmc=st(subs, "relqmc", (n, nl)),
             taua=st(cols, "tauaer", (n, nl, NBND)))
    for k in ("cldfmc", "taucmc", "ciwpmc", "clwpmc"):          # (140, n, nlayers)
        a[k] = _f(np.stack([np.asarray(s_[k], dtype=np.float64) for s_ in subs], axis=1), (gpoints(), n, nl))
    names = ("totuflux", "totdflux", "fnet", "htr", "totuclfl", "totdclfl", "fnetc", "htrc", "dtotuflux_dt", "dtotuclfl_dt")
    out = {k: np.zeros((n, nl + 1), order="F") for k in names}
    c0 = cols[0]
    args = [C.c_int(n), C.c_int(nl), C.c_int(istart), C.c_int(iend), C.c_int(icld), C.c_int(idrv)]
    args += [_p(a[k]) for k in ("pavel", "tavel", "pz", "tz", "tbound", "semiss", "coldry", "wkl", "wbrodl", "wx", "pwvcm")]
    args += [C.c_int(int(c0["inflag"])), C.c_int(int(c0["iceflag"])), C.c_int(int(c0["liqflag"]))]
    args += [_p(a[k]) for k in ("cldfmc", "taucmc", "ciwpmc", "clwpmc", "reicmc", "relqmc", "taua")]
    args += [_p(out[k]) for k in names]
    _check(lib().rrtmg_lw_hip_run_columns_mcica(*args))
    return out


def column_mcica_samples(col, samples, irng=1, alpha=None):
    """The column driver's McICA loop for one prepared column (src/rrtmg_lw.1col.f90:471-660): sub-columns of sample `ims`
    from mcica_subcol_lw with permuteseed = ims * 140, then cldprmc -> rtrnmc; returns the per-sample results (n, 0:nlayers)
    and the generated sub-columns.  The driver's output is the mean over ims = 1..200."""
    nl = int(col["nlayers"])
    r2 = lambda v: np.asfortranarray(np.asarray(v, dtype=np.float64).reshape((1, nl)))
    al = None if alpha is None else r2(alpha)
    subs = []
    for ims in samples:
        g = mcica_subcol_lw(1, nl, int(col["icld"]), ims * gpoints(), irng, r2(col["pavel"]), r2(col["cldfrac"]), r2(col["ciwp"]),
                            r2(col["clwp"]), r2(col["rei"]), r2(col["rel"]),
                            np.asfortranarray(np.asarray(col["tauc"], dtype=np.float64).reshape((NBND, 1, nl), order="F")), al)
        subs.append(dict(cldfmc=g["cldfmcl"][:, 0, :], taucmc=g["taucmcl"][:, 0, :], ciwpmc=g["ciwpmcl"][:, 0, :],
                         clwpmc=g["clwpmcl"][:, 0, :], reicmc=g["reicmcl"][0], relqmc=g["relqmcl"][0]))
    return run_columns_mcica([col] * len(subs), subs), subs


def finalize(selected_only=False):
    """Release the device state of every library this process has loaded (the 140- and the 256-g-point build keep separate states:
    workspace, pinned staging, generator caches, streams); selected_only = True: only that of the library select_gpoints chose."""
    global _initialised
    if selected_only:
        if _lib is not None:
            _lib.rrtmg_lw_hip_finalize()
        return
    for h in _libs.values():
        h.rrtmg_lw_hip_finalize()
    _initialised = False


# ---------------------------------------------------------------------------------------------------
# Aggregation of small calls (include/rrtmg_lw_hip.h: rrtmg_lw_hip_queue_*)
# ---------------------------------------------------------------------------------------------------
class ChunkQueue:
    """Collects rrtmg_lw calls on small column chunks and solves them in one device pass:

        q = ChunkQueue(nlay, icld, idrv, inflglw, iceflglw, liqflglw)
        outs = [q.add(chunk_dict) for chunk_dict in chunks]       # dicts as for rrtmg_lw_from_dict
        q.flush()                                                  # fills every `outs[k]`

    One call of 64 columns costs ~0.5 ms of launch latency on the GPU; a thousand of them queued cost one pass over 64 000 columns."""

    def __init__(self, nlay, icld, idrv, inflglw, iceflglw, liqflglw):
        self.nlay, self.idrv = int(nlay), int(idrv)
        self._keep = []
        _check(lib().rrtmg_lw_hip_queue_begin(C.c_int(int(nlay)), C.c_int(int(icld)), C.c_int(int(idrv)), C.c_int(int(inflglw)),
                                              C.c_int(int(iceflglw)), C.c_int(int(liqflglw))))

    def add(self, d, out=None):
        ncol, nlay = int(d["ncol"]), self.nlay
        shp = dict(play=(ncol, nlay), plev=(ncol, nlay + 1), tlay=(ncol, nlay), tlev=(ncol, nlay + 1), tsfc=(ncol,), emis=(ncol, NBND),
                   taucld=(NBND, ncol, nlay), tauaer=(ncol, nlay, NBND))
        arrs = [_f(d[k], shp.get(k, (ncol, nlay))) for k in _GCM_ORDER + _CLD_ORDER]
        if out is None:
            out = _out_arrays(ncol, nlay, self.idrv)
        null = C.cast(None, _dp)
        args = [C.c_int(ncol), C.cast(None, C.POINTER(C.c_int))] + [_p(a) for a in arrs]
        args += [_p(out[k]) for k in ("uflx", "dflx", "hr", "uflxc", "dflxc", "hrc")]
        args += [_p(out["duflx_dt"]) if self.idrv == 1 else null, _p(out["duflxc_dt"]) if self.idrv == 1 else null]
        _check(lib().rrtmg_lw_hip_queue_add(*args))
        self._keep.append((arrs, out))          # the library holds pointers until the flush
        return out

    def columns(self):
        return int(lib().rrtmg_lw_hip_queue_columns())

    def flush(self):
        try:
            _check(lib().rrtmg_lw_hip_queue_flush())
        finally:
            self._keep = []
