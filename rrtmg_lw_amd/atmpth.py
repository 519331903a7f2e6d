"""IATM = 1: layer quantities from a level sounding, without the reference's LBLATM-derived `rrtatm.f`.

The reference's column driver hands the records after 1.4 of INPUT_RRTM to `RRTATM` (src/rrtmg_lw.1col.f90:998-1002), a 7900-line
atmospheric ray-trace program, and takes from it the layer-mean pressures and temperatures, the level pressures and temperatures and
the molecular column amounts (common blocks /PROFILE/ and /SPECIES/).  The version shipped with RRTMG_LW forces a vertical path
(src/rrtatm.f:ATMPTH `ITYPE = 2`, `ANGLEF = 0.`), so of the ray tracing only this remains:

  * record 3.1 (model, boundaries, molecules; src/rrtatm.f:1569 format 900), 3.2 (H1, H2; format 932), 3.3B (boundary altitudes; 940),
    3.4-3.6 (user profile: RDUNIT :3213-3390 with the unit letters of JOU :3393-3427, defaults from a model atmosphere DEFALT
    :3480-3672, conversion to number densities CONVRT :3868-3975 / WATVAP :3977-4109), or one of the six built-in atmospheres (MDLATM
    :2914-3036);
  * the path levels = profile levels merged with the layer boundaries between H1 and H2 (AMERGE :5075-5251: pressure and densities
    interpolated exponentially, temperature linearly);
  * per path interval the integrals of an exponential atmosphere in steps of at most DELTAS = 5 km (ALAYER :5253-5494 with
    sin(angle) = 0: ds = dz): air column, density-weighted pressure and temperature, molecular amounts;
  * packing into the output layers (FPACK :5805-5981): PBAR, TBAR, amounts, the broadening-gas column WN2L = air - sum of molecules.

`airmwt`: the reference reads the mean molecular weight of air from a common block that nothing sets (the DATA statement,
src/rrtatm.f:1791, is commented out and the driver's /CONSTS/ has nine members, src/rrtmg_lw.1col.f90:792-793, against thirteen in
rrtatm.f:424-425), i.e. 0: every amount given in g/kg (unit letter C - the water vapour of the ICRCCM sonde example) becomes ZERO,
which is what `output_rrtm_ICRCCM_sonde` shows (106.6 W m-2 downward at a 291 K surface).  The default 0.0 reproduces the reference;
pass 28.964 for the value its author commented out.

Not supported (raise): pressure boundaries / pressure-only profiles (negative IBMAX / IMMAX, CMPALT), automatic layering (IBMAX = 0),
layer zeroing (NOZERO = 2), RANGE / BETA path specifications, cross-sections with IATM = 1.
"""
from __future__ import annotations

import math
import os

import numpy as np

PZERO, TZERO = 1013.25, 273.15                      # src/rrtatm.f:1764
AVOGAD, ALOSMT, GASCON = 6.02214199e23, 2.6867775e19, 8.31447200e7      # src/rrtmg_lw_init.f90:255-259 (passed through /CONSTS/)
DELTAS = 5.0                                        # src/rrtatm.f:1763
EPSILN = 1.0e-5                                     # ALAYER
TOL = 5.0e-4                                        # AMERGE

_TABLES = None


def tables():
    global _TABLES
    if _TABLES is None:
        _TABLES = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "mlatmb.npz")))
    return _TABLES


def _f(s):
    s = s.strip()
    return float(s.replace("D", "E").replace("d", "e")) if s else 0.0


def _i(s):
    s = s.strip()
    return int(s) if s else 0


_JOU = {"1": 1, "2": 2, "3": 3, "4": 4, "5": 5, "6": 6, " ": 10, "A": 10, "B": 11, "C": 12, "D": 13, "E": 14, "F": 15, "G": 16,
        "H": 17, "I": 18, "J": 19, "K": 20}


def jou(ch):
    if ch not in _JOU:
        raise ValueError(f"IATM=1: invalid unit letter {ch!r} (JOU)")
    return _JOU[ch]


def expint(x1, x2, a):
    """extra.f EXPINT: exponential interpolation, linear when an end is zero."""
    if x1 == 0.0 or x2 == 0.0:
        return x1 + (x2 - x1) * a
    return x1 * (x2 / x1) ** a


def _lagrange4(z, alt):
    """DEFALT :3480-3560: indices and weights of the 4-point interpolation in the 50 model altitudes."""
    im50 = 50
    i2 = im50
    for im in range(2, im50 + 1):
        i2 = im
        if alt[im - 1] >= z:
            break
    i1, i0, i3 = i2 - 1, i2 - 2, i2 + 1
    if i0 < 1:
        i0, i1, i2, i3 = i1, i2, i3, i3 + 1
    elif i3 > im50:
        if z > alt[im50 - 1]:
            raise ValueError(f"IATM=1: altitude {z} km above the model atmospheres (DEFAULT Z)")
        i3, i2, i1 = i2, i1, i0
        i0 = i1 - 1
    z0, z1, z2, z3 = alt[i0 - 1], alt[i1 - 1], alt[i2 - 1], alt[i3 - 1]
    den1 = (z0 - z1) * (z0 - z2) * (z0 - z3)
    den2 = (z1 - z2) * (z1 - z3) * (z1 - z0)
    den3 = (z2 - z3) * (z2 - z0) * (z2 - z1)
    den4 = (z3 - z0) * (z3 - z1) * (z3 - z2)
    a = (((z - z1) * (z - z2) * (z - z3)) / den1, ((z - z2) * (z - z3) * (z - z0)) / den2,
         ((z - z3) * (z - z0) * (z - z1)) / den3, ((z - z0) * (z - z1) * (z - z2)) / den4)
    return (i0 - 1, i1 - 1, i2 - 1, i3 - 1), a


def _densat(atemp, b):
    return atemp * b * math.exp(18.9766 + (-14.9595) * atemp + (-2.4388) * atemp ** 2) * 1.0e-6


def _watvap(p, t, junit, wmol, amwt, airmwt):
    """WATVAP :3977-4109: water-vapour number density from the profile's unit."""
    rhoair = ALOSMT * (p / PZERO) * (TZERO / t)
    a = TZERO / t
    b = AVOGAD / amwt[0]
    r = airmwt / amwt[0]
    if junit == 10:
        w = wmol * 1.0e-6
        return (w / (1.0 + w)) * rhoair
    if junit == 11:
        return wmol
    if junit == 12:
        w = wmol * r * 1.0e-3
        return (w / (1.0 + w)) * rhoair
    if junit == 13:
        return b * wmol * 1.0e-6
    if junit == 14:
        return ALOSMT * (wmol / PZERO) * (TZERO / t)
    if junit == 15:
        return _densat(TZERO / wmol, b) * wmol / t
    if junit == 16:
        return _densat(TZERO / (TZERO + wmol), b) * (TZERO + wmol) / t
    if junit == 17:
        return _densat(a, b) * (wmol / 100.0)
    raise ValueError(f"IATM=1: water-vapour unit {junit} (WATVAP)")


def _convrt(p, t, junit, wmol, nmol, amwt, airmwt):
    """CONVRT :3868-3975 -> (number densities of the nmol molecules, dry-air density)."""
    rhoair = ALOSMT * (p / PZERO) * (TZERO / t)
    den = np.zeros(nmol)
    den[0] = _watvap(p, t, junit[0], wmol[0], amwt, airmwt)
    dry = rhoair - den[0]
    for k in range(1, nmol):
        b = AVOGAD / amwt[k]
        r = airmwt / amwt[k]
        ju = junit[k]
        if ju <= 10:
            den[k] = wmol[k] * dry * 1.0e-6
        elif ju == 11:
            den[k] = wmol[k]
        elif ju == 12:
            den[k] = r * wmol[k] * 1.0e-3 * dry
        elif ju == 13:
            den[k] = b * wmol[k] * 1.0e-6
        elif ju == 14:
            den[k] = ALOSMT * (wmol[k] / PZERO) * (TZERO / t)
        else:
            raise ValueError(f"IATM=1: unit {ju} of molecule {k + 1} (CONVRT)")
    return den, dry


def read_profile(lines, p, nmol, airmwt):
    """Records 3.4-3.6 (NSMDL / RDUNIT / DEFALT / CONVRT) -> zmdl, pm, tm, denm(level, molecule), next line index."""
    T = tables()
    rec = lines[p].ljust(29); p += 1
    immax_b = _i(rec[0:5])
    if immax_b < 0:
        raise NotImplementedError("IATM=1: profiles on pressure levels (IMMAX < 0, CMPALT) are not supported")
    immax = immax_b
    zmdl, pm, tm = np.zeros(immax), np.zeros(immax), np.zeros(immax)
    denm = np.zeros((immax, nmol))
    for im in range(immax):
        rec = lines[p].ljust(80); p += 1          # format (3E10.3,5X,2A1,1X,A1,1X,39A1)
        z, pr, te = _f(rec[0:10]), _f(rec[10:20]), _f(rec[20:30])
        jcharp, jchart, jlong = rec[35], rec[36], rec[38]
        jchar = rec[40:40 + nmol].ljust(nmol)
        junitp, junitt = jou(jcharp), jou(jchart)
        junit = [jou(c) for c in jchar]
        if jlong == "L":
            w = 15
        elif jlong == " ":
            w = 10
        else:
            raise ValueError(f"IATM=1: JLONG = {jlong!r} on record 3.5")
        wmol = []
        while len(wmol) < nmol:
            ln = lines[p].ljust(8 * w); p += 1
            wmol += [_f(ln[k * w:(k + 1) * w]) for k in range(min(8, nmol - len(wmol)))]
        # CHECK :3429-3478: pressure in atm / torr, temperature in deg C
        if junitp == 11:
            pr = pr * 1013.25
        elif junitp == 12:
            pr = pr * 1013.25 / 760.0
        elif junitp > 12:
            raise ValueError("IATM=1: pressure unit (CHECK)")
        if junitt == 11:
            te = te + 273.15
        elif junitt > 11:
            raise ValueError("IATM=1: temperature unit (CHECK)")
        # DEFALT: values taken from model atmosphere 1-6 at this altitude
        if junitp <= 6 or junitt <= 6 or any(j <= 6 for j in junit):
            idx, a = _lagrange4(z, T["alt"])
            val = lambda x: a[0] * x[idx[0]] + a[1] * x[idx[1]] + a[2] * x[idx[2]] + a[3] * x[idx[3]]
            if junitp <= 6:
                pr = math.exp(val(np.log(T["pm"][junitp - 1])))
            if junitt <= 6:
                te = val(T["tm"][junitt - 1])
            for k in range(nmol):
                if junit[k] <= 6:
                    wmol[k] = val(T["amol"][junit[k] - 1, k]) if k < 7 else val(T["trac"][k - 7])
                    junit[k] = 10
        zmdl[im], pm[im], tm[im] = z, pr, te
        denm[im], _ = _convrt(pr, te, junit, wmol, nmol, T["amwt"], airmwt)
    if (np.diff(zmdl) <= 0).any():
        raise ValueError("IATM=1: input altitudes not in ascending order")
    return zmdl, pm, tm, denm, p


def model_profile(model, nmol):
    """MDLATM :2914-2965: one of the six built-in atmospheres -> zmdl, pm, tm, denm."""
    T = tables()
    m = model - 1
    air = T["amol"][m, 7]
    denm = np.zeros((50, nmol))
    dry = air - T["amol"][m, 0] * air * 1.0e-6
    for k in range(0, min(nmol, 7)):             # (water vapour too: its first assignment, relative to moist air, is overwritten)
        denm[:, k] = T["amol"][m, k] * 1.0e-6 * dry
    for k in range(7, min(nmol, 28)):
        denm[:, k] = T["trac"][k - 7] * 1.0e-6 * dry
    return T["alt"].copy(), T["pm"][m].copy(), T["tm"][m].copy(), denm


def layer_amounts(zmdl, pm, tm, denm, zbnd, h1, h2, re=6371.23):
    """AMERGE + ALAYER (vertical path) + FPACK -> dict(pbar, tbar, pz, tz, altz, amount(molecule, layer), wn2l, rhosum)."""
    zmdl = np.array(zmdl, dtype=float)
    zbnd = np.array(zbnd, dtype=float)
    nmol = denm.shape[1]
    immax = len(zmdl)
    gcair = 1.0e-3 * GASCON / AVOGAD
    hmin, hmax = h1, h2
    # ---- AMERGE: output boundaries zout = {hmin, boundaries inside, hmax}
    zh = [hmin, hmax]
    i1 = len(zbnd)
    for k in range(len(zbnd)):
        if abs(zbnd[k] - zh[0]) < TOL:
            zh[0] = zbnd[k]
        if zbnd[k] > zh[0]:
            i1 = k
            break
    zout = [zh[0]]
    ib = i1
    while True:
        if ib < len(zbnd):
            if abs(zbnd[ib] - zh[1]) < TOL:
                zh[1] = zbnd[ib]
            if zbnd[ib] < zh[1]:
                zout.append(zbnd[ib])
                ib += 1
                continue
        zout.append(zh[1])
        break
    ioutmx = len(zout)
    # ---- path levels: profile levels from hmin merged with zout
    im = None
    for k in range(immax):
        if zmdl[k] >= hmin:
            im = k
            break
    if im is None:
        raise ValueError("IATM=1: AMERGE - HMIN above the profile")
    zpth, pp, tp, denp = [], [], [], []
    iout = 0
    while True:
        take_model = False
        if im < immax:
            if abs(zout[iout] - zmdl[im]) < TOL:
                zmdl[im] = zout[iout]
            if not (zout[iout] < zmdl[im]):
                take_model = True
        if take_model:
            if zout[iout] == zmdl[im]:
                iout += 1
            zpth.append(zmdl[im]); pp.append(pm[im]); tp.append(tm[im]); denp.append(denm[im].copy())
            im += 1
            if abs(zpth[-1] - zout[ioutmx - 1]) < TOL:
                zout[ioutmx - 1] = zpth[-1]
            if zpth[-1] == zout[ioutmx - 1]:
                break
        else:
            z = zout[iout]
            jm = max(im, 1)
            if jm >= immax:
                raise ValueError("IATM=1: layer boundary above the profile")
            a = (z - zmdl[jm - 1]) / (zmdl[jm] - zmdl[jm - 1])
            zpth.append(z)
            pp.append(expint(pm[jm - 1], pm[jm], a))
            tp.append(tm[jm - 1] + (tm[jm] - tm[jm - 1]) * a)
            denp.append(np.array([expint(denm[jm - 1, k], denm[jm, k], a) for k in range(nmol)]))
            iout += 1
            if abs(zpth[-1] - zout[ioutmx - 1]) < TOL:
                zpth[-1] = zout[ioutmx - 1]
            if zpth[-1] == zout[ioutmx - 1]:
                break
    ipmax = len(zpth)
    # ---- ALAYER with sin(angle) = 0: the path element equals the height element (the quadrature weights of the refracted path,
    # w1 + w2 + w3 with ds/dx = 1, add up to dh)
    ppsum, tpsum, rhopsm = np.zeros(ipmax - 1), np.zeros(ipmax - 1), np.zeros(ipmax - 1)
    amtp = np.zeros((ipmax - 1, nmol))
    for j in range(ipmax - 1):
        z1, z2 = zpth[j], zpth[j + 1]
        pa, pb_end = pp[j], pp[j + 1]
        if pb_end == pa:
            raise ValueError("IATM=1: pressures in adjoining levels must differ")
        ta, tb = tp[j], tp[j + 1]
        rhoa, rhob_end = pa / (gcair * ta), pb_end / (gcair * tb)
        dz = z2 - z1
        hp = -dz / math.log(pb_end / pa)
        hrho = -dz / math.log(rhob_end / rhoa) if abs(rhob_end / rhoa - 1.0) >= EPSILN else 1.0e30
        dena = denp[j].copy()
        hden = np.zeros(nmol)
        for k in range(nmol):
            da, db = denp[j][k], denp[j + 1][k]
            if da == 0.0 or db == 0.0 or abs(1.0 - da / db) <= EPSILN:
                hden[k] = 0.0
            else:
                hden[k] = -dz / math.log(db / da)
        h1s = z1
        while True:
            dh = DELTAS
            h3 = h1s + dh
            if h3 > z2:
                h3 = z2
            dh = h3 - h1s
            # the three path points x = r (cos = -1): Simpson-type weights of the refracted-path quadrature, with ds/dx = 1
            r1, r3 = re + h1s, re + h3
            r2 = re + (h1s + dh / 2.0)
            d31, d32, d21 = r3 - r1, r3 - r2, r2 - r1
            if d32 == 0.0 or d21 == 0.0:
                w1, w2, w3 = 0.5 * d31, 0.0, 0.5 * d31
            else:
                w1 = (2.0 - d32 / d21) * d31 / 6.0
                w2 = d31 ** 3 / (d32 * d21 * 6.0)
                w3 = (2.0 - d21 / d32) * d31 / 6.0
            ds = w1 + w2 + w3
            dsdz = ds / dh
            pb = pa * math.exp(-dh / hp)
            rhob = rhoa * math.exp(-dh / hrho)
            if dh / hrho >= EPSILN:
                ppsum[j] += dsdz * (hp / (1.0 + hp / hrho)) * (pa * rhoa - pb * rhob)
                tpsum[j] += dsdz * hp * (pa - pb) / gcair
                rhopsm[j] += dsdz * hrho * (rhoa - rhob)
            else:
                ppsum[j] += 0.5 * ds * (pa * rhoa + pb * rhob)
                tpsum[j] += 0.5 * ds * (pa + pb) / gcair
                rhopsm[j] += 0.5 * ds * (rhoa + rhob)
            denb = np.zeros(nmol)
            for k in range(nmol):
                if hden[k] == 0.0 or abs(dh / hden[k]) < EPSILN:
                    denb[k] = denp[j][k] + (denp[j + 1][k] - denp[j][k]) * (h3 - z1) / dz
                    amtp[j, k] += 0.5 * (dena[k] + denb[k]) * ds * 1.0e5
                else:
                    denb[k] = denp[j][k] * math.exp(-(h3 - z1) / hden[k])
                    amtp[j, k] += dsdz * hden[k] * (dena[k] - denb[k]) * 1.0e5
            pa, rhoa, dena = pb, rhob, denb
            if h3 < z2:
                h1s = h3
            else:
                break
    # ---- FPACK
    nlay = ioutmx - 1
    pbar, tbar, rhosum = np.zeros(nlay), np.zeros(nlay), np.zeros(nlay)
    amount = np.zeros((nmol, nlay))
    pz, tz = np.zeros(nlay + 1), np.zeros(nlay + 1)
    pz[0], tz[0] = pp[0], tp[0]
    io = 0
    for ip in range(ipmax - 1):
        pbar[io] += ppsum[ip]
        tbar[io] += tpsum[ip]
        rhosum[io] += rhopsm[ip]
        amount[:, io] += amtp[ip]
        if zpth[ip + 1] == zout[io + 1]:
            pz[io + 1], tz[io + 1] = pp[ip + 1], tp[ip + 1]
            io += 1
    if io != nlay:
        raise RuntimeError("IATM=1: FPACK - path levels and layer boundaries do not line up")
    pbar = pbar / rhosum
    tbar = tbar / rhosum
    rhosum = rhosum * 1.0e5
    wn2l = rhosum - amount.sum(axis=0)
    return dict(pbar=pbar, tbar=tbar, pz=pz, tz=tz, altz=np.array(zout), amount=amount, wn2l=wn2l, rhosum=rhosum)


def rrtatm(lines, p, ixsect=0, airmwt=0.0):
    """The records from 3.1 on -> dict(nlayers, pavel, tavel, pz, tz, altz, wkl(nmol, nlayers) in molecules cm-2, wbrodl, nmol), next line."""
    if ixsect == 1:
        raise NotImplementedError("IATM=1 with cross-sections (XAMNTS) is not supported")
    rec = lines[p].ljust(90); p += 1          # 3.1: format (7I5,I2,1X,I2,4F10.3,A10)
    model, itype, ibmax_b, n_zero, noprnt, nmol, ipunch = (_i(rec[5 * k:5 * k + 5]) for k in range(7))
    re_, hspace = _f(rec[40:50]), _f(rec[50:60])
    if _f(rec[70:80]) != 0.0:
        raise ValueError("IATM=1: a value has been read for co2mx (record 3.1)")
    if nmol == 0:
        nmol = 7
    if not 0 <= model <= 6 or nmol > 28:
        raise ValueError("IATM=1: record 3.1 (MODEL, NMOL)")
    if ibmax_b < 0:
        raise NotImplementedError("IATM=1: layer boundaries in pressure (IBMAX < 0, CMPALT) are not supported")
    if ibmax_b == 0:
        raise NotImplementedError("IATM=1: automatic layering (IBMAX = 0, AUTLAY) is not supported")
    if n_zero == 2:
        raise NotImplementedError("IATM=1: zeroing of small amounts (NOZERO = 2) is not supported")
    if hspace == 0.0:
        hspace = 100.0
    if re_ == 0.0:
        re_ = 6378.39 if model == 1 else (6356.91 if model in (4, 5) else 6371.23)
    rec = lines[p].ljust(70); p += 1          # 3.2: format (5F10.4,I5,5X,F10.4); the angle is forced to 0 (vertical path)
    h1, h2, rng, beta = _f(rec[0:10]), _f(rec[10:20]), _f(rec[30:40]), _f(rec[40:50])
    if rng > 0.0 or beta > 0.0:
        raise NotImplementedError("IATM=1: RANGE / BETA path specifications are not supported")
    if h1 >= h2:
        raise ValueError("IATM=1: H1 >= H2 with a zenith angle of 0 (FSCGEO)")
    zbnd = []
    while len(zbnd) < ibmax_b:                # 3.3B: format (8F10.3)
        ln = lines[p].ljust(80); p += 1
        zbnd += [_f(ln[10 * k:10 * k + 10]) for k in range(min(8, ibmax_b - len(zbnd)))]
    if (np.diff(zbnd) <= 0).any():
        raise ValueError("IATM=1: ZBND not ascending")
    if model == 0:
        zmdl, pm, tm, denm, p = read_profile(lines, p, nmol, airmwt)
    else:
        zmdl, pm, tm, denm = model_profile(model, nmol)
    n = int(np.sum(zmdl <= hspace + 0.001))   # MDLATM: levels up to HSPACE
    zmdl, pm, tm, denm = zmdl[:n], pm[:n], tm[:n], denm[:n]
    zmax = zmdl[-1]
    if zbnd[0] < zmdl[0]:
        if abs(zbnd[0] - zmdl[0]) <= 1.0e-4:
            zbnd[0] = zmdl[0]
        else:
            raise ValueError("IATM=1: boundaries outside of the atmosphere")
    if h2 > zmax:                             # REDUCE
        h2 = zmax
    if h1 < zmdl[0]:
        raise ValueError("IATM=1: H1 below the profile")
    r = layer_amounts(zmdl, pm, tm, denm, zbnd, h1, h2, re_)
    nlay = len(r["pbar"])
    return dict(nlayers=nlay, pavel=r["pbar"], tavel=r["tbar"], pz=r["pz"], tz=r["tz"], altz=r["altz"], wkl=r["amount"],
                wbrodl=r["wn2l"], nmol=nmol), p
