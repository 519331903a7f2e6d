#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Writes the two patched copies a 256-g-point build of the REFERENCE needs (kept under oracle/_ref/gen, never
committed): modules/parrrtm.f90 with its own commented-out "Use for 256 g-point model" parameters switched on (and the 140-point ones
off), and src/rrtmg_lw_init.f90 with lwcmbdat's map arrays set as its comment prescribes (:313-314: "the full 256 g-point set can be
restored with ngptlw=256, ngc=16*16, ngn=256*1., etc."), and the sixteen modules/rrlw_kgNN.f90, each of which repeats its band's
reduced g-point count as a local parameter (e.g. rrlw_kg01.f90:60 "ng1 = 10") that sizes the combined arrays: set to 16.  No
algorithm is touched.

usage: patch_g256.py <reference root> <output dir>"""
import os
import re
import sys


def patch_parrrtm(src):
    out, mode = [], None
    for ln in src.splitlines():
        if "Use for 140 g-point model" in ln:
            mode = 140
        elif "Use for 256 g-point model" in ln:
            mode = 256
        elif mode == 140 and re.match(r"\s+integer\(kind=im\), parameter :: (ngptlw|ng\d+|ngs\d+)\b", ln):
            ln = "!" + ln
        elif mode == 256 and re.match(r"!\s+integer\(kind=im\), parameter :: (ngptlw|ng\d+|ngs\d+)\b", ln):
            ln = ln[1:]
        out.append(ln)
    text = "\n".join(out) + "\n"
    assert re.search(r"^\s+integer\(kind=im\), parameter :: ngptlw = 256", text, re.M) and not re.search(r"^\s+integer\(kind=im\), parameter :: ngptlw = 140", text, re.M)
    assert re.search(r"^\s+integer\(kind=im\), parameter :: ng13 = 16", text, re.M) and re.search(r"^\s+integer\(kind=im\), parameter :: ngs15 = 240", text, re.M)
    return text


def patch_init(src):
    a = src.index("      ngc(:) = (/10,12,16,14,16,8,12,8,12,6,8,8,4,2,2,2/)")
    b = src.index("      wt(:) = (/ 0.1527534276_rb")
    new = ("      ngc(:) = 16\n"
           "      ngs(:) = (/16,32,48,64,80,96,112,128,144,160,176,192,208,224,240,256/)\n"
           "      ngm(:) = (/ (mod(igm256-1,16)+1, igm256 = 1, 256) /)\n"
           "      ngn(:) = 1\n"
           "      ngb(:) = (/ ((igm256-1)/16+1, igm256 = 1, 256) /)\n")
    text = src[:a] + new + src[b:]
    # the implied-do index needs a declaration in lwcmbdat
    text = text.replace("      subroutine lwcmbdat\n!***************************************************************************\n\n      save\n",
                        "      subroutine lwcmbdat\n!***************************************************************************\n\n      integer(kind=im) :: igm256\n      save\n")
    assert "igm256" in text and "integer(kind=im) :: igm256" in text
    return text


def patch_kg(src, band):
    text, n = re.subn(r"(integer\(kind=im\), parameter :: ng%d\s*=\s*)\d+" % band, r"\g<1>16", src)
    assert n == 1, band
    return text


def main():
    ref, out = sys.argv[1], sys.argv[2]
    os.makedirs(out, exist_ok=True)
    open(os.path.join(out, "parrrtm.g256.f90"), "w").write(patch_parrrtm(open(os.path.join(ref, "modules", "parrrtm.f90")).read()))
    for b in range(1, 17):
        open(os.path.join(out, "rrlw_kg%02d.g256.f90" % b), "w").write(patch_kg(open(os.path.join(ref, "modules", "rrlw_kg%02d.f90" % b)).read(), b))
    open(os.path.join(out, "rrtmg_lw_init.g256.f90"), "w").write(patch_init(open(os.path.join(ref, "src", "rrtmg_lw_init.f90")).read()))


if __name__ == "__main__":
    main()
