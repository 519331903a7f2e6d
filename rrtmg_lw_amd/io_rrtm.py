"""Reader and writer for the column driver's text files: INPUT_RRTM, IN_CLD_RRTM, IN_AER_RRTM in, OUTPUT_RRTM in and out.

Follows the record formats of the reference's standalone driver (``readprof`` src/rrtmg_lw.1col.f90:755-1149,
``readcld`` :1152-1208, ``readaer`` :1211-1293, ``xsident`` :1296-1363; output formats :737-746) so that
the examples under run_examples_std_atm can be fed to this package's *prepared-column* entry
(``rrtmg_lw_hip_column``) and compared with the checked-in OUTPUT_RRTM files.  Only IATM=0 inputs are
supported (IATM=1 needs the RRTATM layering program, out of scope - SURVEY.md 2).
"""
from __future__ import annotations

import numpy as np

NBND = 16
AMD = 28.9660          # src/rrtmg_lw.1col.f90:776
AMW = 18.0160          # :777
GRAV = 9.8066          # src/rrtmg_lw_init.f90:243

_XS_ALIAS = {  # xsident, src/rrtmg_lw.1col.f90:1319-1326
    "CCL4": 1,
    "CCL3F": 2, "CFCL3": 2, "CFC11": 2, "F11": 2,
    "CCL2F2": 3, "CF2CL2": 3, "CFC12": 3, "F12": 3,
    "CHCLF2": 4, "CHF2CL": 4, "CFC22": 4, "F22": 4,
}


def _f(s):
    s = s.strip()
    if not s:
        return 0.0
    return float(s.replace("D", "E").replace("d", "e"))


def _i(s):
    s = s.strip()
    return int(s) if s else 0


def _fields(line, widths):
    out, p = [], 0
    for w in widths:
        out.append(line[p:p + w])
        p += w
    return out


def read_cld(path):
    """IN_CLD_RRTM -> dict(inflag, iceflag, liqflag, layers={lay: (frac, d1, d2, d3, d4)}) (readcld)."""
    with open(path) as f:
        lines = f.read().splitlines()
    h = lines[0].ljust(15)
    out = dict(inflag=_i(h[3:5]), iceflag=_i(h[9:10]), liqflag=_i(h[14:15]), layers={})
    for ln in lines[1:]:
        if ln[:1] == "%":
            break
        ln = ln.ljust(55)
        lay = _i(ln[2:5])
        vals = [_f(ln[5 + 10 * k:15 + 10 * k]) for k in range(5)]
        out["layers"][lay] = tuple(vals)
    return out


def read_aer(path, nlayers):
    """IN_AER_RRTM -> tauaer(nlayers, 16) (readaer)."""
    with open(path) as f:
        lines = f.read().splitlines()
    tau = np.zeros((nlayers, NBND))
    p = 0
    naer = _i(lines[p][3:5]); p += 1
    for _ in range(naer):
        nlay = _i(lines[p][2:5]); p += 1
        for _ in range(nlay):
            ln = lines[p].ljust(5 + 7 * NBND); p += 1
            lay = _i(ln[2:5])
            tau[lay - 1, :] = [_f(ln[5 + 7 * k:12 + 7 * k]) for k in range(NBND)]
    return tau


def read_input_rrtm(path, cld_path=None, aer_path=None, airmwt=0.0):
    """Parse one INPUT_RRTM case into the prepared-column dictionary handed to the physics.

    IATM = 1 (a level sounding instead of layer amounts; the reference calls RRTATM, src/rrtmg_lw.1col.f90:998-1002) goes through
    rrtmg_lw_amd/atmpth.py; `airmwt` is that module's switch between the reference's unset mean molecular weight of air (0: amounts given
    in g/kg vanish, as in the reference's own output_rrtm_ICRCCM_sonde) and 28.964.

    Keys mirror readprof's outputs: nlayers, iout, imca, icld, iaer, idrv, pavel, tavel, pz, tz (0:nlayers),
    tbound, dtbound, semiss(16), coldry, wkl(7,nlayers), wbrodl, wx(4,nlayers), pwvcm, dz (m), inflag,
    iceflag, liqflag, cldfrac, tauc(16,nlayers), ciwp, clwp, rei, rel, tauaer(nlayers,16), idcor,
    decorr_con, juldat, lat.
    """
    with open(path) as f:
        lines = f.read().splitlines()
    p = 0
    while not lines[p].startswith("$"):
        p += 1
    p += 1
    ctl = lines[p].ljust(95); p += 1
    iaer, iatm, ixsect = _i(ctl[18:20]), _i(ctl[49:50]), _i(ctl[69:70])
    iout, idrv, imca, icld = _i(ctl[87:90]), _i(ctl[91:92]), _i(ctl[93:94]), _i(ctl[94:95])
    rec = lines[p].ljust(16 + 5 * NBND); p += 1
    tbound, iemiss = _f(rec[0:10]), _i(rec[11:12])
    semis = [_f(rec[15 + 5 * k:20 + 5 * k]) for k in range(NBND)]
    dtbound = 0.0
    if idrv == 1:
        dtbound = _f(lines[p][0:10]); p += 1
    idcor, decorr_con, juldat, lat = 0, 0.0, 0, 0.0
    if icld in (4, 5):
        idcor = _i(lines[p][8:10]); p += 1
        if idcor == 0:
            decorr_con = _f(lines[p][0:10]); p += 1
        elif idcor == 1:
            juldat, lat = _i(lines[p][5:10]), _f(lines[p][10:20]); p += 1
    semiss = np.ones(NBND)
    if iemiss == 1 and semis[0] != 0.0:
        semiss[:] = semis[0]
    elif iemiss == 2:
        for k in range(NBND):
            if semis[k] != 0.0:
                semiss[k] = semis[k]

    if iatm == 1:
        from . import atmpth
        a, p = atmpth.rrtatm(lines, p, ixsect=ixsect, airmwt=airmwt)
        nlayers, nmol = a["nlayers"], a["nmol"]
        pavel, tavel, pz, tz, altz = a["pavel"], a["tavel"], a["pz"], a["tz"], a["altz"]
        wkl = np.zeros((7, nlayers))
        wkl[:min(nmol, 7)] = a["wkl"][:7]
        wbrodl = a["wbrodl"]
        return _finish_column(locals(), cld_path, aer_path)
    rec = lines[p].ljust(10); p += 1
    iform, nlayers, nmol = _i(rec[1:2]), _i(rec[2:5]), _i(rec[5:10])
    if nmol == 0:
        nmol = 7
    w1 = (15, 10, 10, 3, 2, 1, 7, 8, 7, 7, 8, 7) if iform == 1 else (10, 10, 10, 3, 2, 1, 7, 8, 7, 7, 8, 7)
    w2 = (15, 10, 10, 3, 2, 23, 7, 8, 7) if iform == 1 else (10, 10, 10, 3, 2, 23, 7, 8, 7)
    w3 = 15 if iform == 1 else 10
    pavel, tavel = np.zeros(nlayers), np.zeros(nlayers)
    pz, tz, altz = np.zeros(nlayers + 1), np.zeros(nlayers + 1), np.zeros(nlayers + 1)
    wkl = np.zeros((7, nlayers))
    wbrodl = np.zeros(nlayers)
    for l in range(nlayers):
        if l == 0:
            f = _fields(lines[p].ljust(sum(w1)), w1); p += 1
            pavel[0], tavel[0] = _f(f[0]), _f(f[1])
            altz[0], pz[0], tz[0] = _f(f[6]), _f(f[7]), _f(f[8])
            altz[1], pz[1], tz[1] = _f(f[9]), _f(f[10]), _f(f[11])
        else:
            f = _fields(lines[p].ljust(sum(w2)), w2); p += 1
            pavel[l], tavel[l] = _f(f[0]), _f(f[1])
            altz[l + 1], pz[l + 1], tz[l + 1] = _f(f[6]), _f(f[7]), _f(f[8])
        ln = lines[p].ljust(8 * w3); p += 1
        v = [_f(ln[k * w3:(k + 1) * w3]) for k in range(8)]
        wkl[:, l] = v[:7]
        wbrodl[l] = v[7]
        if nmol > 7:
            p += 1   # molecules 8..nmol are not used by RRTMG_LW
    wx0 = None
    ixindx = []
    if ixsect == 1:
        nxmol0 = _i(lines[p][0:5]); p += 1
        names = lines[p].ljust(70); p += 1
        for k in range(min(nxmol0, 7)):
            ixindx.append(_XS_ALIAS.get(names[10 * k:10 * k + 10].strip().upper(), 0))
        if nxmol0 > 7:
            names = lines[p].ljust(80); p += 1
            for k in range(nxmol0 - 7):
                ixindx.append(_XS_ALIAS.get(names[10 * k:10 * k + 10].strip().upper(), 0))
        iformx = _i(lines[p][1:2]); p += 1
        wx3 = 15 if iformx == 1 else 10
        wx0 = np.zeros((max(nxmol0, 7), nlayers))
        for l in range(nlayers):
            p += 1
            ln = lines[p].ljust(8 * wx3); p += 1
            wx0[:7, l] = [_f(ln[k * wx3:(k + 1) * wx3]) for k in range(7)]
            if nxmol0 > 7:
                ln = lines[p].ljust(8 * wx3); p += 1
                wx0[7:nxmol0, l] = [_f(ln[k * wx3:(k + 1) * wx3]) for k in range(nxmol0 - 7)]

    return _finish_column(locals(), cld_path, aer_path)


def _finish_column(v, cld_path, aer_path):
    """Column amounts, precipitable water, cloud and aerosol files: what readprof does after the layer records (or RRTATM)."""
    nlayers, nmol, iout, imca, icld, iaer, idrv = (v[k] for k in ("nlayers", "nmol", "iout", "imca", "icld", "iaer", "idrv"))
    pavel, tavel, pz, tz, altz, wkl, wbrodl = (v[k] for k in ("pavel", "tavel", "pz", "tz", "altz", "wkl", "wbrodl"))
    tbound, dtbound, semiss = v["tbound"], v["dtbound"], v["semiss"]
    idcor, decorr_con, juldat, lat = v["idcor"], v["decorr_con"], v["juldat"], v["lat"]
    wx0, ixindx = v.get("wx0"), v.get("ixindx", [])
    if tbound < 0:
        tbound = tz[0]
    # column amounts: readprof :1014-1060
    imix = 0 if (wkl[:nmol, 0] > 1.0).any() else 1
    imixx = 1 if (wx0 is not None and wx0[0, 0] <= 1.0) else 0
    coldry = np.zeros(nlayers)
    wx = np.zeros((4, nlayers))
    amttl = wvttl = 0.0
    for l in range(nlayers):
        summol = 0.0
        for m in range(1, 7):
            summol += wkl[m, l]
        if imix == 1:
            coldry[l] = wbrodl[l] / (1.0 - summol)
            wkl[:, l] = coldry[l] * wkl[:, l]
        else:
            coldry[l] = wbrodl[l] + summol
        amttl += coldry[l] + wkl[0, l]
        wvttl += wkl[0, l]
        if wx0 is not None:
            for ix, idx in enumerate(ixindx):
                if idx != 0:
                    wx[idx - 1, l] = (coldry[l] * wx0[ix, l] if imixx == 1 else wx0[ix, l]) * 1.0e-20
    wvsh = (AMW * wvttl) / (AMD * amttl)
    pwvcm = wvsh * (1.0e3 * pz[0]) / (1.0e2 * GRAV)

    out = dict(nlayers=nlayers, iout=iout, imca=imca, icld=icld, iaer=iaer, idrv=idrv,
               pavel=pavel, tavel=tavel, pz=pz, tz=tz, tbound=tbound, dtbound=dtbound,
               semiss=semiss, coldry=coldry, wkl=wkl, wbrodl=wbrodl, wx=wx, pwvcm=pwvcm,
               dz=(altz[1:] - altz[:-1]) * 1000.0, idcor=idcor, decorr_con=decorr_con,
               juldat=juldat, lat=lat, inflag=0, iceflag=0, liqflag=0,
               cldfrac=np.zeros(nlayers), tauc=np.zeros((NBND, nlayers)), ciwp=np.zeros(nlayers),
               clwp=np.zeros(nlayers), rei=np.zeros(nlayers), rel=np.zeros(nlayers),
               tauaer=np.zeros((nlayers, NBND)))
    if icld >= 1:
        if cld_path is None:
            raise ValueError("icld >= 1 needs an IN_CLD_RRTM file")
        c = read_cld(cld_path)
        out.update(inflag=c["inflag"], iceflag=c["iceflag"], liqflag=c["liqflag"])
        for lay, (frac, d1, d2, d3, d4) in c["layers"].items():
            k = lay - 1
            out["cldfrac"][k] = frac
            if c["inflag"] == 0:
                out["tauc"][:, k] = d1
            else:
                out["ciwp"][k] = d1 * d2
                out["clwp"][k] = d1 * (1.0 - d2)
                out["rei"][k] = d3
                out["rel"][k] = d4
    if iaer == 10:
        if aer_path is None:
            raise ValueError("iaer = 10 needs an IN_AER_RRTM file")
        out["tauaer"] = read_aer(aer_path, nlayers)
    return out


def read_output_rrtm(path):
    """OUTPUT_RRTM -> list of blocks dict(wn1, wn2, level, pz, uflx, dflx, fnet, htr), arrays indexed by level 0..n."""
    blocks = []
    cur = None
    with open(path) as f:
        for ln in f:
            s = ln.strip()
            if s.startswith("Wavenumbers:"):
                t = s.replace("Wavenumbers:", "").replace("cm-1,", "").replace("-", " ", 1).split()
                cur = dict(wn1=float(t[0]), wn2=float(t[1]), rows=[])
                blocks.append(cur)
                continue
            if cur is None:
                continue
            t = s.split()
            if len(t) == 6 and t[0].isdigit():
                cur["rows"].append([float(x) for x in t])
            elif s.startswith("Modules"):
                cur = None
    out = []
    for b in blocks:
        r = np.array(sorted(b["rows"], key=lambda x: x[0]))
        out.append(dict(wn1=b["wn1"], wn2=b["wn2"], level=r[:, 0].astype(int), pz=r[:, 1],
                        uflx=r[:, 2], dflx=r[:, 3], fnet=r[:, 4], htr=r[:, 5]))
    return out


# ---------------------------------------------------------------------------------------------------------------------
# OUTPUT_RRTM writer: the column driver's output records, src/rrtmg_lw.1col.f90:615-639 (write statements) and
# :737-746 (formats 9952-9958, 9899-9901, 9903).
# ---------------------------------------------------------------------------------------------------------------------
WAVENUM1 = (10., 350., 500., 630., 700., 820., 980., 1080., 1180., 1390., 1480., 1800., 2080., 2250., 2380., 2600.)    # src/rrtmg_lw_init.f90:215-220
WAVENUM2 = (350., 500., 630., 700., 820., 980., 1080., 1180., 1390., 1480., 1800., 2080., 2250., 2380., 2600., 3250.)


def _ffmt(x, w, d):
    """Fortran Fw.d edit descriptor as the reference's compiler prints it: right-justified; the optional zero before the decimal
    point is dropped when the field has no room for it (f6.5 of 0.067 -> '.06700'); asterisks on overflow."""
    t = f"{x:.{d}f}"
    if len(t) > w and t.startswith("0."):
        t = t[1:]
    elif len(t) > w and t.startswith("-0."):
        t = "-" + t[2:]
    return t.rjust(w) if len(t) <= w else "*" * w


def format_output_row(i, pz, up, dn, net, htr):
    """One level record: formats 9952-9958, chosen by the pressure (src/rrtmg_lw.1col.f90:623-637)."""
    if pz < 1.e-2:
        pre = " " * 9 + _ffmt(pz, 7, 6) + " " * 3          # 9952
    elif pz < 1.e-1:
        pre = " " * 9 + _ffmt(pz, 6, 5) + " " * 4          # 9953
    elif pz < 1.:
        pre = " " * 8 + _ffmt(pz, 6, 4) + " " * 5          # 9954
    elif pz < 10.:
        pre = " " * 7 + _ffmt(pz, 6, 3) + " " * 6          # 9955
    elif pz < 100.:
        pre = " " * 6 + _ffmt(pz, 6, 2) + " " * 7          # 9956
    else:
        pre = " " * 5 + _ffmt(pz, 6, 1) + " " * 8          # 9957, 9958
    return " " + f"{int(i):3d}" + pre + _ffmt(up, 8, 4) + " " * 6 + _ffmt(dn, 8, 4) + " " * 6 + _ffmt(net, 12, 7) + " " * 10 + _ffmt(htr, 9, 5)


def format_output_block(istart, iend, pz, up, dn, net, htr, iplon=1):
    """Header (9899-9901), the level records from the top level down to 0, and the page-feed record (9903)."""
    lines = [f" Wavenumbers: {_ffmt(WAVENUM1[istart - 1], 6, 1)} - {_ffmt(WAVENUM2[iend - 1], 6, 1)} cm-1, ATM {int(iplon):6d}",
             " LEVEL    PRESSURE   UPWARD FLUX   DOWNWARD FLUX    NET FLUX       HEATING RATE",
             "             mb          W/m2          W/m2           W/m2          degree/day"]
    for i in range(len(pz) - 1, -1, -1):
        lines.append(format_output_row(i, pz[i], up[i], dn[i], net[i], htr[i]))
    lines.append("\f")
    return lines


def write_output_rrtm(path, blocks, footer=None):
    """blocks: list of dict(istart, iend, pz, uflx, dflx, fnet, htr[, iplon]) in the order the driver writes them (the total, then
    the bands when iout = 99).  `footer` replaces the reference's list of module versions (format 9910)."""
    out = []
    for b in blocks:
        out += format_output_block(b["istart"], b["iend"], b["pz"], b["uflx"], b["dflx"], b["fnet"], b["htr"], b.get("iplon", 1))
    out += footer if footer is not None else ["  Modules and versions used in this calculation:", "",
                                               "     rrtmg_lw_amd (MI355X-native RRTMG_LW hot path): librrtmg_lw_hip.so over the prepared-column entry"]
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
