"""TEST INFRASTRUCTURE: ctypes bindings for the C oracle (oracle/liboracle.so) and for the flang-built
reference harness (oracle/_ref/libref_{nomcica,mcica}.so).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module.  All arrays are float64, Fortran order.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
STATIC_BLOB = os.path.join(REPO, "rrtmg_lw_amd", "data", "lw_static.bin")
STANDIN_KDATA = os.path.join(REPO, "rrtmg_lw_amd", "data", "standin.kdata.bin")
NBND, NGPT = 16, 140

_dp = C.POINTER(C.c_double)


def _p(a):
    return a.ctypes.data_as(_dp)


def _f(a, shape=None):
    a = np.asfortranarray(a, dtype=np.float64)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError(f"shape {a.shape} != {shape}")
    return a


GCM_FIELDS_2D = ("play", "plev", "tlay", "tlev")
GAS_FIELDS = ("h2ovmr", "o3vmr", "co2vmr", "ch4vmr", "n2ovmr", "o2vmr", "cfc11vmr", "cfc12vmr", "cfc22vmr", "ccl4vmr")


def _gcm_call(fn, ncol, nlay, icld, idrv, d, mcica, ngpt=NGPT):
    """Shared marshalling for the rrtmg_lw(ncol, nlay, ...) argument list (src/rrtmg_lw_rad*.f90:99-108)."""
    arrs = [_f(d["play"], (ncol, nlay)), _f(d["plev"], (ncol, nlay + 1)), _f(d["tlay"], (ncol, nlay)),
            _f(d["tlev"], (ncol, nlay + 1)), _f(d["tsfc"], (ncol,))]
    arrs += [_f(d[g], (ncol, nlay)) for g in GAS_FIELDS]
    arrs.append(_f(d["emis"], (ncol, NBND)))
    if mcica:
        cld = [_f(d["cldfmcl"], (ngpt, ncol, nlay)), _f(d["taucmcl"], (ngpt, ncol, nlay)),
               _f(d["ciwpmcl"], (ngpt, ncol, nlay)), _f(d["clwpmcl"], (ngpt, ncol, nlay)),
               _f(d["reicmcl"], (ncol, nlay)), _f(d["relqmcl"], (ncol, nlay))]
    else:
        cld = [_f(d["cldfr"], (ncol, nlay)), _f(d["taucld"], (NBND, ncol, nlay)), _f(d["cicewp"], (ncol, nlay)),
               _f(d["cliqwp"], (ncol, nlay)), _f(d["reice"], (ncol, nlay)), _f(d["reliq"], (ncol, nlay))]
    cld.append(_f(d["tauaer"], (ncol, nlay, NBND)))
    out = {k: np.zeros((ncol, nlay + 1), order="F") for k in ("uflx", "dflx", "uflxc", "dflxc", "duflx_dt", "duflxc_dt")}
    out["hr"] = np.zeros((ncol, nlay), order="F")
    out["hrc"] = np.zeros((ncol, nlay), order="F")
    icld_c = C.c_int(icld)
    args = [C.c_int(ncol), C.c_int(nlay), C.byref(icld_c), C.c_int(idrv)]
    args += [_p(a) for a in arrs]
    args += [C.c_int(int(d["inflglw"])), C.c_int(int(d["iceflglw"])), C.c_int(int(d["liqflglw"]))]
    args += [_p(a) for a in cld]
    args += [_p(out[k]) for k in ("uflx", "dflx", "hr", "uflxc", "dflxc", "hrc", "duflx_dt", "duflxc_dt")]
    rc = fn(*args)
    out["icld"] = icld_c.value
    return rc, out


def _column_call(fn, col, istart, iend, iout, icld, idrv, byref_scalars, ngpt=NGPT):
    nl = int(col["nlayers"])
    z = lambda *s: np.zeros(s, order="F")
    outs = [z(nl + 1) for _ in range(10)]
    taug, fracs = z(nl, ngpt), z(nl, ngpt)
    ncb = C.c_int(0)
    a = dict(pavel=_f(col["pavel"], (nl,)), tavel=_f(col["tavel"], (nl,)), pz=_f(col["pz"], (nl + 1,)),
             tz=_f(col["tz"], (nl + 1,)), semiss=_f(col["semiss"], (NBND,)), coldry=_f(col["coldry"], (nl,)),
             wkl=_f(col["wkl"], (7, nl)), wbrodl=_f(col["wbrodl"], (nl,)), wx=_f(col["wx"], (4, nl)),
             cldfrac=_f(col["cldfrac"], (nl,)), tauc=_f(col["tauc"], (NBND, nl)), ciwp=_f(col["ciwp"], (nl,)),
             clwp=_f(col["clwp"], (nl,)), rei=_f(col["rei"], (nl,)), rel=_f(col["rel"], (nl,)),
             taua=_f(col["tauaer"], (nl, NBND)))
    args = [C.c_int(nl), C.c_int(istart), C.c_int(iend), C.c_int(iout), C.c_int(icld), C.c_int(idrv),
            _p(a["pavel"]), _p(a["tavel"]), _p(a["pz"]), _p(a["tz"]), C.c_double(float(col["tbound"])),
            _p(a["semiss"]), _p(a["coldry"]), _p(a["wkl"]), _p(a["wbrodl"]), _p(a["wx"]),
            C.c_double(float(col["pwvcm"])), C.c_int(int(col["inflag"])), C.c_int(int(col["iceflag"])),
            C.c_int(int(col["liqflag"])), _p(a["cldfrac"]), _p(a["tauc"]), _p(a["ciwp"]), _p(a["clwp"]),
            _p(a["rei"]), _p(a["rel"]), _p(a["taua"])]
    args += [_p(o) for o in outs] + [_p(taug), _p(fracs), C.byref(ncb)]
    rc = fn(*args)
    names = ("totuflux", "totdflux", "fnet", "htr", "totuclfl", "totdclfl", "fnetc", "htrc", "dtotuflux_dt", "dtotuclfl_dt")
    res = dict(zip(names, outs))
    res.update(taug=taug, fracs=fracs, ncbands=ncb.value)
    return rc, res


def _column_mc_call(fn, col, sub, istart, iend, iout, icld, idrv, ngpt=NGPT):
    """Prepared column + one set of sub-columns (cldfmc, taucmc, ciwpmc, clwpmc (140,nlayers); reicmc, relqmc (nlayers))."""
    nl = int(col["nlayers"])
    z = lambda *s: np.zeros(s, order="F")
    outs = [z(nl + 1) for _ in range(10)]
    taug, fracs = z(nl, ngpt), z(nl, ngpt)
    ncb = C.c_int(0)
    a = dict(pavel=_f(col["pavel"], (nl,)), tavel=_f(col["tavel"], (nl,)), pz=_f(col["pz"], (nl + 1,)),
             tz=_f(col["tz"], (nl + 1,)), semiss=_f(col["semiss"], (NBND,)), coldry=_f(col["coldry"], (nl,)),
             wkl=_f(col["wkl"], (7, nl)), wbrodl=_f(col["wbrodl"], (nl,)), wx=_f(col["wx"], (4, nl)),
             cldfmc=_f(sub["cldfmc"], (ngpt, nl)), taucmc=_f(sub["taucmc"], (ngpt, nl)), ciwpmc=_f(sub["ciwpmc"], (ngpt, nl)),
             clwpmc=_f(sub["clwpmc"], (ngpt, nl)), reicmc=_f(sub["reicmc"], (nl,)), relqmc=_f(sub["relqmc"], (nl,)),
             taua=_f(col["tauaer"], (nl, NBND)))
    args = [C.c_int(nl), C.c_int(istart), C.c_int(iend), C.c_int(iout), C.c_int(icld), C.c_int(idrv),
            _p(a["pavel"]), _p(a["tavel"]), _p(a["pz"]), _p(a["tz"]), C.c_double(float(col["tbound"])),
            _p(a["semiss"]), _p(a["coldry"]), _p(a["wkl"]), _p(a["wbrodl"]), _p(a["wx"]),
            C.c_double(float(col["pwvcm"])), C.c_int(int(col["inflag"])), C.c_int(int(col["iceflag"])),
            C.c_int(int(col["liqflag"])), _p(a["cldfmc"]), _p(a["taucmc"]), _p(a["ciwpmc"]), _p(a["clwpmc"]),
            _p(a["reicmc"]), _p(a["relqmc"]), _p(a["taua"])]
    args += [_p(o) for o in outs] + [_p(taug), _p(fracs), C.byref(ncb)]
    rc = fn(*args)
    names = ("totuflux", "totdflux", "fnet", "htr", "totuclfl", "totdclfl", "fnetc", "htrc", "dtotuflux_dt", "dtotuclfl_dt")
    res = dict(zip(names, outs))
    res.update(taug=taug, fracs=fracs, ncbands=ncb.value)
    return rc, res


class Oracle:
    """The plain-C restatement (oracle/rrtmg_lw_oracle.c)."""

    def __init__(self, kdata=STANDIN_KDATA, cpdair=1004.0, static=STATIC_BLOB, gpoints=140):
        """gpoints = 256: the build that keeps every band's 16 original g-points (liboracle_g256.so)."""
        assert gpoints in (140, 256)
        path = os.path.join(HERE, "liboracle.so" if gpoints == 140 else "liboracle_g256.so")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} missing - run `make -C oracle liboracle.so` or __graft_entry__.build()")
        self.lib = C.CDLL(path)
        self.lib.orc_errmsg.restype = C.c_char_p
        self.lib.orc_get_table.restype = C.c_long
        rc = self.lib.orc_init(static.encode(), kdata.encode(), C.c_double(cpdair))
        if rc != 0:
            raise RuntimeError(f"orc_init failed: {self.lib.orc_errmsg().decode()}")
        self.ngpt = int(self.lib.orc_ngpt())
        assert self.ngpt == gpoints

    def errmsg(self):
        return self.lib.orc_errmsg().decode()

    def rrtmg_lw(self, ncol, nlay, icld, idrv, d, mcica=False):
        fn = self.lib.orc_rrtmg_lw_mcica if mcica else self.lib.orc_rrtmg_lw_nomcica
        rc, out = _gcm_call(fn, ncol, nlay, icld, idrv, d, mcica, self.ngpt)
        if rc != 0:
            raise RuntimeError(f"oracle: {self.errmsg()}")
        return out

    def column(self, col, istart=1, iend=16, iout=0, icld=None, idrv=None):
        icld = int(col["icld"]) if icld is None else icld
        idrv = int(col["idrv"]) if idrv is None else idrv
        rc, res = _column_call(self.lib.orc_column, col, istart, iend, iout, icld, idrv, False, self.ngpt)
        if rc != 0:
            raise RuntimeError(f"oracle: {self.errmsg()}")
        return res

    def column_mc(self, col, sub, istart=1, iend=16, iout=0, icld=None, idrv=None):
        """One McICA sample of the column driver: cldprmc -> setcoef -> taumol -> rtrnmc on the given sub-columns."""
        icld = int(col["icld"]) if icld is None else icld
        idrv = int(col["idrv"]) if idrv is None else idrv
        rc, res = _column_mc_call(self.lib.orc_column_mc, col, sub, istart, iend, iout, icld, idrv, self.ngpt)
        if rc != 0:
            raise RuntimeError(f"oracle: {self.errmsg()}")
        return res

    def get_alpha(self, ncol, nlay, icld, idcor, decorr_con, dz, lat, juldat, cldfrac):
        a = np.zeros((ncol, nlay), order="F")
        self.lib.orc_get_alpha(C.c_int(ncol), C.c_int(nlay), C.c_int(icld), C.c_int(idcor), C.c_double(decorr_con),
                               _p(_f(dz, (ncol, nlay))), _p(_f(lat, (ncol,))), C.c_int(juldat), _p(_f(cldfrac, (ncol, nlay))), _p(a))
        return a

    def mcica_subcol(self, ncol, nlay, icld, permuteseed, irng, play, cldfrac, ciwp, clwp, rei, rel, tauc, alpha):
        """mcica_subcol_lw with the GCM argument list (src/mcica_subcol_gen_lw.f90:183-185)."""
        o = _subcol_outputs(ncol, nlay, self.ngpt)
        irng_c = C.c_int(irng)
        ins = [_f(play, (ncol, nlay)), _f(cldfrac, (ncol, nlay)), _f(ciwp, (ncol, nlay)), _f(clwp, (ncol, nlay)),
               _f(rei, (ncol, nlay)), _f(rel, (ncol, nlay)), _f(tauc, (NBND, ncol, nlay)), _f(alpha, (ncol, nlay))]
        rc = self.lib.orc_mcica_subcol(C.c_int(ncol), C.c_int(nlay), C.c_int(icld), C.c_int(permuteseed), C.byref(irng_c),
                                       *[_p(a) for a in ins],
                                       *[_p(o[k]) for k in ("cldfmcl", "ciwpmcl", "clwpmcl", "reicmcl", "relqmcl", "taucmcl")])
        if rc != 0:
            raise RuntimeError(f"oracle: {self.errmsg()}")
        o["irng"] = irng_c.value
        return o

    def table(self, band, name):
        n = self.lib.orc_get_table(C.c_int(band), name.encode(), None, C.c_long(0))
        if n < 0:
            raise KeyError(f"band {band}: {name}")
        out = np.zeros(n)
        self.lib.orc_get_table(C.c_int(band), name.encode(), _p(out), C.c_long(n))
        return out

    def luts(self):
        t, e, f = np.zeros(10001), np.zeros(10001), np.zeros(10001)
        self.lib.orc_get_luts(_p(t), _p(e), _p(f))
        return t, e, f


def _subcol_outputs(ncol, nlay, ngpt=NGPT):
    z3 = lambda: np.zeros((ngpt, ncol, nlay), order="F")
    z2 = lambda: np.zeros((ncol, nlay), order="F")
    return dict(cldfmcl=z3(), ciwpmcl=z3(), clwpmcl=z3(), reicmcl=z2(), relqmcl=z2(), taucmcl=z3())


class Reference:
    """The reference's own Fortran, built by oracle/Makefile (only available where /root/reference is, or
    where a prebuilt oracle/_ref travelled with the snapshot)."""

    def __init__(self, flavour="nomcica", kdata=STANDIN_KDATA, cpdair=1004.0):
        path = os.path.join(HERE, "_ref", f"libref_{flavour}.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.flavour = flavour
        self.ngpt = 256 if flavour.endswith("_g256") else NGPT          # "nomcica_g256": the reference's own 256-g-point parameters (oracle/patch_g256.py)
        self.lib = C.CDLL(path)
        kb = kdata.encode()
        self.lib.ref_set_kdata_path(kb, C.c_int(len(kb)))
        self.lib.ref_init(C.c_double(cpdair))

    @staticmethod
    def available(flavour="nomcica"):
        return os.path.exists(os.path.join(HERE, "_ref", f"libref_{flavour}.so"))

    def rrtmg_lw(self, ncol, nlay, icld, idrv, d):
        _, out = _gcm_call(self.lib.ref_rrtmg_lw, ncol, nlay, icld, idrv, d, self.flavour.startswith("mcica"), self.ngpt)
        return out

    def column(self, col, istart=1, iend=16, iout=0, icld=None, idrv=None):
        icld = int(col["icld"]) if icld is None else icld
        idrv = int(col["idrv"]) if idrv is None else idrv
        _, res = _column_call(self.lib.ref_column, col, istart, iend, iout, icld, idrv, True, self.ngpt)
        return res

    def column_mc(self, col, sub, istart=1, iend=16, iout=0, icld=None, idrv=None):
        icld = int(col["icld"]) if icld is None else icld
        idrv = int(col["idrv"]) if idrv is None else idrv
        _, res = _column_mc_call(self.lib.ref_column_mc, col, sub, istart, iend, iout, icld, idrv, self.ngpt)
        return res

    def get_alpha_1col(self, nlay, icld, idcor, decorr_con, dz, lat, juldat, cldfrac):
        a = np.zeros(nlay)
        self.lib.ref_get_alpha_1col(C.c_int(nlay), C.c_int(icld), C.c_int(idcor), C.c_double(decorr_con), _p(_f(dz, (nlay,))),
                                    C.c_double(lat), C.c_int(juldat), _p(_f(cldfrac, (nlay,))), _p(a))
        return a

    def mcica_subcol_1col(self, nlay, icld, ims, irng, play, cldfrac, ciwp, clwp, rei, rel, tauc, alpha):
        """The reference's one-column generator (src/mcica_subcol_gen_lw.1col.f90:171); permuteseed = ims * ngptlw."""
        z2 = lambda: np.zeros((self.ngpt, nlay), order="F")
        o = dict(cldfmc=z2(), ciwpmc=z2(), clwpmc=z2(), reicmc=np.zeros(nlay), relqmc=np.zeros(nlay), taucmc=z2())
        irng_c = C.c_int(irng)
        ins = [_f(play, (nlay,)), _f(cldfrac, (nlay,)), _f(ciwp, (nlay,)), _f(clwp, (nlay,)), _f(rei, (nlay,)), _f(rel, (nlay,)),
               _f(tauc, (NBND, nlay)), _f(alpha, (nlay,))]
        self.lib.ref_mcica_subcol_1col(C.c_int(nlay), C.c_int(icld), C.c_int(ims), C.byref(irng_c), *[_p(a) for a in ins],
                                       *[_p(o[k]) for k in ("cldfmc", "ciwpmc", "clwpmc", "reicmc", "relqmc", "taucmc")])
        o["irng"] = irng_c.value
        return o


def reference_rrtatm_available():
    return os.path.exists(os.path.join(HERE, "_ref", "libref_rrtatm.so"))


def reference_rrtatm(path):
    """The REFERENCE's own RRTATM (oracle/_ref/libref_rrtatm.so, oracle/ref_rrtatm_harness.f90) on the IATM = 1 records of an
    INPUT_RRTM file -> dict(nlayers, pavel, tavel, pz, tz, altz, wkl(7, nlayers), wbrodl, nmol).  The harness skips the records readprof
    itself reads before the call (src/rrtmg_lw.1col.f90:919-957)."""
    import tempfile
    lib = C.CDLL(os.path.join(HERE, "_ref", "libref_rrtatm.so"))
    lines = open(path).read().splitlines()
    p = 0
    while not lines[p].startswith("$"):
        p += 1
    ctl = lines[p + 1].ljust(95)
    idrv, icld = int(ctl[91:92].strip() or 0), int(ctl[94:95].strip() or 0)
    p += 3                                    # '$' record, control record 1.2, surface record 1.4
    if idrv == 1:
        p += 1
    if icld in (4, 5):
        p += 2
    mxl = 603
    nlay, nmol = C.c_long(0), C.c_long(0)
    pavel, tavel, wbrodl = np.zeros(mxl), np.zeros(mxl), np.zeros(mxl)
    pz, tz, altz = np.zeros(mxl + 1), np.zeros(mxl + 1), np.zeros(mxl + 1)
    wkl = np.zeros((mxl, 7))                  # Fortran wkl(7, mxl)
    with tempfile.TemporaryDirectory() as td:
        t6 = os.path.join(td, "TAPE6").encode()
        pb = os.path.abspath(path).encode()
        lib.ref_rrtatm(pb, C.c_long(len(pb)), C.c_long(p), t6, C.c_long(len(t6)), C.c_long(mxl), C.byref(nlay), _p(pavel), _p(tavel),
                       _p(pz), _p(tz), _p(altz), _p(wkl), _p(wbrodl), C.byref(nmol))
    n = nlay.value
    return dict(nlayers=n, pavel=pavel[:n].copy(), tavel=tavel[:n].copy(), pz=pz[:n + 1].copy(), tz=tz[:n + 1].copy(),
                altz=altz[:n + 1].copy(), wkl=wkl[:n].T.copy(), wbrodl=wbrodl[:n].copy(), nmol=nmol.value)
