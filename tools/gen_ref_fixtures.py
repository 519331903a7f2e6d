#!/usr/bin/env python3
"""Generate tests/golden/ref_*.npz: outputs of the REFERENCE's own Fortran (oracle/_ref, built from /root/reference
by oracle/Makefile) on small seeded inputs with the stand-in k-tables.  Runs only in the build container (needs
/root/reference for the flang build); the resulting arrays are data and travel with the repo so that
tests/test_oracle_vs_ref.py can pin the C oracle on machines without the reference.

Conventions of the reference build: flang -O2 -fdefault-real-8 -fdefault-double-8 -fdefault-integer-8 (the shipped
makefiles' -r8 -i8), rtrnmr's faccmb* zero-initialised (the reference reads them uninitialised).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.bindings import Reference  # noqa: E402
from rrtmg_lw_amd.io_rrtm import read_input_rrtm  # noqa: E402
from rrtmg_lw_amd.synth import STRESS_KINDS, make_gcm_inputs, make_stress_inputs  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
OUT_KEYS = ("uflx", "dflx", "hr", "uflxc", "dflxc", "hrc", "duflx_dt", "duflxc_dt")
GCM_CASES = [  # name, config, ncol, nlay, icld
    ("clear72", "clear", 4, 72, 0), ("clear51_rtrn", "clear", 3, 51, 1), ("cloudy72_mr", "cloudy", 24, 72, 2),
    ("cloudy72_rnd", "cloudy", 24, 72, 1), ("aer137", "aer_idrv", 8, 137, 2), ("aer40_rnd", "aer_idrv", 8, 40, 1),
    ("cloudy72_icld9", "cloudy", 6, 72, 9),
]
STRESS_CASES = [(k, 12, 72, icld) for k in STRESS_KINDS for icld in (0, 2)]      # name, ncol, nlay, icld (synth.make_stress_inputs)
MCICA_CASES = [  # name, config, ncol, nlay, icld, irng, ims (permuteseed = ims * 140), idcor, juldat
    ("mr_kiss", "cloudy", 6, 40, 2, 0, 1, 0, 0), ("rnd_kiss", "cloudy", 5, 72, 1, 0, 2, 0, 0), ("max_kiss", "cloudy", 5, 51, 3, 0, 1, 0, 0),
    ("exp_kiss", "cloudy", 6, 40, 4, 0, 1, 1, 150), ("exprnd_kiss", "aer_idrv", 6, 40, 5, 0, 3, 1, 300),
    ("mr_mt", "cloudy", 4, 40, 2, 1, 1, 0, 0), ("exprnd_mt", "aer_idrv", 4, 33, 5, 1, 2, 0, 20), ("max_mt", "cloudy", 3, 36, 3, 1, 1, 0, 0),
]
COL_CASES = [  # name, input, cloud file, aerosol file
    ("MLS-clr", "input_rrtm_MLS-clr", None, None), ("MLS-clr-aer12", "input_rrtm_MLS-clr-aer12", None, "in_aer_rrtm-aer12"),
    ("MLS-clr-idrv1", "input_rrtm_MLS-clr-idrv1", None, None), ("MLS-clr-xsec", "input_rrtm_MLS-clr-xsec", None, None),
    ("MLS-cld5-icld2", "input_rrtm_MLS-cld-imca0-icld2", "in_cld_rrtm-cld5", None),
    ("MLS-cld7-icld2", "input_rrtm_MLS-cld-imca0-icld2", "in_cld_rrtm-cld7", None),
    ("MLW-clr", "input_rrtm_MLW-clr", None, None), ("SAW-clr", "input_rrtm_SAW-clr", None, None), ("TROP-clr", "input_rrtm_TROP-clr", None, None),
]


G256_GCM_CASES = [("clear72", "clear", 4, 72, 0), ("cloudy72_mr", "cloudy", 16, 72, 2), ("aer60_rnd", "aer_idrv", 8, 60, 1)]
G256_MCICA_CASES = [("cloudy40_mr_kiss", "cloudy", 10, 40, 2, 0, 2), ("aer33_rnd_mt", "aer_idrv", 6, 33, 1, 1, 1)]     # name, config, ncol, nlay, icld, irng, ims
G256_COL_CASES = [c for c in COL_CASES if c[0] in ("MLS-clr", "MLS-cld5-icld2", "SAW-clr")]
COL_KEYS = ("totuflux", "totdflux", "fnet", "htr", "totuclfl", "totdclfl", "fnetc", "htrc", "dtotuflux_dt", "dtotuclfl_dt")


def g256():
    """ref_g256_*.npz: the reference's own 256-g-point configuration (oracle/patch_g256.py: its commented-out parameters switched on)."""
    ref = Reference("nomcica_g256")
    for name, cfg, ncol, nlay, icld in G256_GCM_CASES:
        d = make_gcm_inputs(ncol, nlay, cfg, col0=4242)
        o = ref.rrtmg_lw(ncol, nlay, icld, d["idrv"], d)
        np.savez_compressed(os.path.join(G, f"ref_g256_gcm_{name}.npz"), config=cfg, ncol=ncol, nlay=nlay, icld=icld, col0=4242,
                            icld_out=o["icld"], **{k: o[k] for k in OUT_KEYS})
    for name, inp, cld, aer in G256_COL_CASES:
        col = read_input_rrtm(os.path.join(G, inp), os.path.join(G, cld) if cld else None, os.path.join(G, aer) if aer else None)
        o = ref.column(col)
        bands = {}
        for b in (1, 6, 13, 16):
            ob = ref.column(col, b, b, 99)
            bands.update({f"b{b}_up": ob["totuflux"], f"b{b}_dn": ob["totdflux"], f"b{b}_htr": ob["htr"]})
        np.savez_compressed(os.path.join(G, f"ref_g256_col_{name}.npz"), inp=inp, cld=cld or "", aer=aer or "",
                            taug=o["taug"].astype(np.float32), fracs=o["fracs"].astype(np.float32), **{k: o[k] for k in COL_KEYS}, **bands)
    # McICA in the same configuration: 256 sub-columns from the reference's one-column generator, then its McICA rrtmg_lw
    refm = Reference("mcica_g256")
    for name, cfg, ncol, nlay, icld, irng, ims in G256_MCICA_CASES:
        d = make_gcm_inputs(ncol, nlay, cfg, col0=777)
        z3 = lambda: np.zeros((256, ncol, nlay), order="F")
        sub = dict(cldfmcl=z3(), ciwpmcl=z3(), clwpmcl=z3(), taucmcl=z3(), reicmcl=np.zeros((ncol, nlay), order="F"),
                   relqmcl=np.zeros((ncol, nlay), order="F"))
        for c in range(ncol):
            r = refm.mcica_subcol_1col(nlay, icld, ims, irng, d["play"][c], d["cldfr"][c], d["cicewp"][c], d["cliqwp"][c],
                                       d["reice"][c], d["reliq"][c], d["taucld"][:, c, :], np.zeros(nlay))
            for k3, k2 in (("cldfmcl", "cldfmc"), ("ciwpmcl", "ciwpmc"), ("clwpmcl", "clwpmc"), ("taucmcl", "taucmc")):
                sub[k3][:, c, :] = r[k2]
            sub["reicmcl"][c] = r["reicmc"]
            sub["relqmcl"][c] = r["relqmc"]
        dd = dict(d)
        dd.update(sub)
        o = refm.rrtmg_lw(ncol, nlay, icld, d["idrv"], dd)
        assert set(np.unique(sub["cldfmcl"])) <= {0.0, 1.0}
        np.savez_compressed(os.path.join(G, f"ref_g256_mcica_{name}.npz"), config=cfg, ncol=ncol, nlay=nlay, icld=icld, irng=irng, ims=ims,
                            col0=777, icld_out=o["icld"], mask=np.packbits(sub["cldfmcl"].astype(np.uint8).ravel(order="F")),
                            ciwpsum=sub["ciwpmcl"].sum(axis=0), clwpsum=sub["clwpmcl"].sum(axis=0), taucsum=sub["taucmcl"].sum(axis=0),
                            **{k: o[k] for k in OUT_KEYS})
    print("wrote 256-g-point fixtures to", G)


RRTATM_CASES = ("input_rrtm_ICRCCM_sonde", "input_rrtm_iatm1_model6", "input_rrtm_iatm1_units")


def rrtatm():
    """ref_rrtatm_*.npz: the reference's own RRTATM (oracle/_ref/libref_rrtatm.so) on the IATM = 1 inputs."""
    from oracle.bindings import reference_rrtatm
    for name in RRTATM_CASES:
        r = reference_rrtatm(os.path.join(G, name))
        np.savez_compressed(os.path.join(G, "ref_rrtatm_" + name.replace("input_rrtm_", "") + ".npz"), inp=name, **r)
    print("wrote RRTATM fixtures to", G)


def main():
    if "--rrtatm" in sys.argv:      # only the IATM = 1 layering fixtures
        return rrtatm()
    if "--g256" in sys.argv:        # only the 256-g-point fixtures (the others are left untouched)
        return g256()
    ref = Reference("nomcica")
    for name, cfg, ncol, nlay, icld in GCM_CASES:
        d = make_gcm_inputs(ncol, nlay, cfg, col0=4242)
        o = ref.rrtmg_lw(ncol, nlay, icld, d["idrv"], d)
        np.savez_compressed(os.path.join(G, f"ref_gcm_{name}.npz"), config=cfg, ncol=ncol, nlay=nlay, icld=icld, col0=4242,
                            icld_out=o["icld"], **{k: o[k] for k in OUT_KEYS})
    for kind, ncol, nlay, icld in STRESS_CASES:
        d = make_stress_inputs(kind, ncol, nlay, col0=31)
        o = ref.rrtmg_lw(ncol, nlay, icld, 0, d)
        assert all(np.isfinite(o[k]).all() for k in OUT_KEYS)
        np.savez_compressed(os.path.join(G, f"ref_stress_{kind}_icld{icld}.npz"), kind=kind, ncol=ncol, nlay=nlay, icld=icld, col0=31,
                            icld_out=o["icld"], **{k: o[k] for k in OUT_KEYS})
    for name, inp, cld, aer in COL_CASES:
        col = read_input_rrtm(os.path.join(G, inp), os.path.join(G, cld) if cld else None, os.path.join(G, aer) if aer else None)
        if col["imca"] == 1:
            col["imca"] = 0
        o = ref.column(col)
        bands = {}
        if col["iout"] == 99:       # per-band blocks as the column driver produces them (istart = iend = band, iout = 99)
            for b in range(1, 17):
                ob = ref.column(col, b, b, 99)
                bands[f"b{b}_up"] = ob["totuflux"]
                bands[f"b{b}_dn"] = ob["totdflux"]
                bands[f"b{b}_htr"] = ob["htr"]
        np.savez_compressed(os.path.join(G, f"ref_col_{name}.npz"), inp=inp, cld=cld or "", aer=aer or "",
                            taug=o["taug"], fracs=o["fracs"], ncbands=o["ncbands"],
                            **{k: o[k] for k in ("totuflux", "totdflux", "fnet", "htr", "totuclfl", "totdclfl", "fnetc", "htrc",
                                                 "dtotuflux_dt", "dtotuclfl_dt")}, **bands)
    # McICA: the reference's one-column generator (src/mcica_subcol_gen_lw.1col.f90) per column - for the Mersenne
    # Twister every column is its own call (its own freshly seeded stream), as in the reference's column driver -
    # followed by the McICA rrtmg_lw (src/rrtmg_lw_rad.f90) on the generated sub-columns.
    refm = Reference("mcica")
    for name, cfg, ncol, nlay, icld, irng, ims, idcor, juldat in MCICA_CASES:
        d = make_gcm_inputs(ncol, nlay, cfg, col0=777)
        rng = np.random.default_rng(12)
        dz, lat = rng.uniform(100, 1500, (ncol, nlay)), rng.uniform(-90, 90, ncol)
        alpha = np.zeros((ncol, nlay), order="F")
        z3 = lambda: np.zeros((140, ncol, nlay), order="F")
        sub = dict(cldfmcl=z3(), ciwpmcl=z3(), clwpmcl=z3(), taucmcl=z3(), reicmcl=np.zeros((ncol, nlay), order="F"),
                   relqmcl=np.zeros((ncol, nlay), order="F"))
        for c in range(ncol):
            alpha[c] = refm.get_alpha_1col(nlay, icld, idcor, 2000.0, dz[c], lat[c], juldat, d["cldfr"][c])
            r = refm.mcica_subcol_1col(nlay, icld, ims, irng, d["play"][c], d["cldfr"][c], d["cicewp"][c], d["cliqwp"][c],
                                       d["reice"][c], d["reliq"][c], d["taucld"][:, c, :], alpha[c])
            for k3, k2 in (("cldfmcl", "cldfmc"), ("ciwpmcl", "ciwpmc"), ("clwpmcl", "clwpmc"), ("taucmcl", "taucmc")):
                sub[k3][:, c, :] = r[k2]
            sub["reicmcl"][c] = r["reicmc"]
            sub["relqmcl"][c] = r["relqmc"]
        dd = dict(d)
        dd.update(sub)
        o = refm.rrtmg_lw(ncol, nlay, icld, d["idrv"], dd)
        assert set(np.unique(sub["cldfmcl"])) <= {0.0, 1.0}
        np.savez_compressed(os.path.join(G, f"ref_mcica_{name}.npz"), config=cfg, ncol=ncol, nlay=nlay, icld=icld, irng=irng, ims=ims,
                            idcor=idcor, juldat=juldat, col0=777, dz=dz, lat=lat, alpha=alpha, icld_out=o["icld"],
                            mask=np.packbits(sub["cldfmcl"].astype(np.uint8).ravel(order="F")),
                            ciwpsum=sub["ciwpmcl"].sum(axis=0), clwpsum=sub["clwpmcl"].sum(axis=0), taucsum=sub["taucmcl"].sum(axis=0),
                            **{k: o[k] for k in OUT_KEYS})
    print("wrote fixtures to", G)


if __name__ == "__main__":
    main()
