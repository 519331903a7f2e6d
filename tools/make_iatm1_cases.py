#!/usr/bin/env python3
"""Writes two IATM = 1 test inputs of our own (tests/golden/input_rrtm_iatm1_*; the reference ships only the ICRCCM sonde) that reach the
parts of the layering its example does not: a built-in model atmosphere (MODEL = 6) with boundaries between its levels, and a user
profile with mixed units (pressure in atm, temperature in K and deg C, water vapour as relative humidity / ppmv / g/kg / dew point,
number densities, g/m3, partial pressure, defaults from model atmospheres 2 and 6), a path that starts above the lowest level and a
boundary on a profile level.  Record formats: src/rrtatm.f:1569 (3.1), :1602 (3.2), :1617 (3.3B), NSMDL 905 / RDUNIT 900, 905."""
import os

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
HEAD = ["0        1         2         3         4         5         6         7         8         9",
        "123456789-123456789-123456789-123456789-123456789-123456789-123456789-123456789-123456789-"]
CTL = " HI=0 F4=0 CN=0 AE 0 EM=0 SC=0 FI=0 PL=0 TS=0 AM=1 MG=0 LA=0 OD=0 XS=0   00   00    3    0   00"


def rec31(model, ibmax, nmol):
    return "%5d%5d%5d%5d%5d%5d%5d" % (model, 2, ibmax, 1, 1, nmol, 0)


def bnd(zs):
    out = []
    for k in range(0, len(zs), 8):
        out.append("".join("%10.3f" % z for z in zs[k:k + 8]))
    return out


def main():
    # (A) U.S. standard atmosphere, 0-60 km, 20 boundaries
    zs = [0.0, 0.4, 1.3, 2.5, 3.7, 5.2, 7.1, 9.0, 11.5, 13.9, 16.4, 19.2, 22.8, 26.1, 30.4, 35.0, 41.3, 47.5, 53.0, 60.0]
    a = HEAD + ["$ IATM=1 test: built-in model atmosphere 6, boundaries between the model levels", CTL, "%10.3f" % 288.2,
                rec31(6, len(zs), 7), "%10.4f%10.4f" % (0.0, 60.0)] + bnd(zs) + ["%%%%%"]
    open(os.path.join(G, "input_rrtm_iatm1_model6"), "w").write("\n".join(a) + "\n")
    # (B) user profile, mixed units
    prof = [  # z km, p, T, (JCHARP, JCHART), JCHAR(7), values
        (0.20, 0.985, 291.0, "BA", "HAD6B26", (70.0, 365.0, 9.0e-5, 0.0, 3.0e12, 0.0, 0.0)),
        (0.80, 0.918, 15.2, "BB", "HAD6B26", (64.0, 365.0, 8.0e-5, 0.0, 2.6e12, 0.0, 0.0)),
        (1.50, 850.0, 11.0, "AB", "GAD6B26", (3.5, 365.0, 7.5e-5, 0.0, 2.2e12, 0.0, 0.0)),
        (3.00, 700.0, 273.4, "AA", "CAA6A26", (2.9, 365.0, 0.045, 0.0, 0.12, 0.0, 0.0)),
        (5.50, 505.0, 255.6, "AA", "CAA6A26", (0.9, 365.0, 0.06, 0.0, 0.11, 0.0, 0.0)),
        (9.00, 310.0, 229.0, "AA", "AAA6A26", (210.0, 365.0, 0.12, 0.0, 0.09, 0.0, 0.0)),
        (12.0, 197.0, 216.7, "AA", "AAA6A66", (22.0, 365.0, 0.35, 0.0, 0.06, 0.0, 0.0)),
        (16.0, 105.0, 214.0, "AA", "AAE6A66", (4.5, 365.0, 1.6e-4, 0.0, 0.03, 0.0, 0.0)),
        (22.0, 41.0, 219.5, "AA", "AAA6666", (4.8, 365.0, 5.2, 0.0, 0.0, 0.0, 0.0)),
        (30.0, 12.1, 226.8, "AA", "AAA6666", (5.0, 365.0, 7.1, 0.0, 0.0, 0.0, 0.0)),
        (40.0, 2.9, 251.0, "AA", "A6A6666", (5.3, 0.0, 5.0, 0.0, 0.0, 0.0, 0.0)),
        (50.0, 0.80, 270.5, "6A", "6666666", (0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0)),
        (65.0, 0.0, 0.0, "66", "6666666", (0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0)),
    ]
    zb = [0.5, 0.8, 1.0, 2.0, 3.5, 5.5, 8.0, 11.0, 14.0, 18.0, 23.0, 29.0, 36.0, 44.0, 52.0, 61.0]
    b = HEAD + ["$ IATM=1 test: user profile, mixed units, path from 0.5 km", CTL, "%10.3f" % 289.0,
                rec31(0, len(zb), 7), "%10.4f%10.4f" % (0.5, 61.0)] + bnd(zb) + ["%5d %s" % (len(prof), "mixed-unit test profile")]
    for z, p, t, (cp, ct), jc, w in prof:
        b.append("%10.3f%10.3E%10.3E     %s%s   %s" % (z, p, t, cp, ct, jc))
        b.append("".join("%10.3E" % x for x in w))
    b.append("%%%%%")
    open(os.path.join(G, "input_rrtm_iatm1_units"), "w").write("\n".join(b) + "\n")
    print("wrote", G)


if __name__ == "__main__":
    main()
