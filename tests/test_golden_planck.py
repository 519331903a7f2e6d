"""The k-data-independent part of the reference's golden OUTPUT_RRTM files.

With surface emissivity 1 the upward flux at level 0 of every band block is
    pi * 1e4 * delwave * B_band(tbound) * sum_g fracs(1, g)
(src/rrtmg_lw_rtrn.f90:476-489,549-562,580-583).  The real Planck fractions of a band sum to one, and so do the
stand-in ones, so these 16 + 1 numbers per file pin the Planck tables (totplnk / totplk16), setcoef's temperature
interpolation (src/rrtmg_lw_setcoef.f90:173-269, including the istart = 16 variant), delwave and fluxfac against the
reference's own checked-in results - without the absorption coefficients that are missing from the mount.
"""
import os

import numpy as np
import pytest

from rrtmg_lw_amd.io_rrtm import read_input_rrtm, read_output_rrtm

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("inp,out", [("input_rrtm_MLS-clr", "output_rrtm_MLS-clr"),
                                     ("input_rrtm_MLS-clr-aer12", "output_rrtm_MLS-clr-aer12")])
def test_surface_emission_per_band(oracle, inp, out):
    aer = os.path.join(G, "in_aer_rrtm-aer12") if "aer" in inp else None
    col = read_input_rrtm(os.path.join(G, inp), None, aer)
    blocks = read_output_rrtm(os.path.join(G, out))
    assert len(blocks) == 17
    tot = oracle.column(col)
    # printed with 4 decimals; the real fractions sum to 1 within ~1e-5
    assert abs(tot["totuflux"][0] - blocks[0]["uflx"][0]) < 2e-3
    for b in range(1, 17):
        ob = oracle.column(col, b, b, 99)
        assert abs(ob["totuflux"][0] - blocks[b]["uflx"][0]) < 2e-3 * max(1.0, blocks[b]["uflx"][0] / 50), b


@pytest.mark.parametrize("name,tb", [("MLW-clr", 272.2), ("SAW-clr", 257.2), ("TROP-clr", 300.0)])
def test_surface_emission_other_atmospheres(oracle, name, tb):
    col = read_input_rrtm(os.path.join(G, f"input_rrtm_{name}"))
    assert col["tbound"] == tb
    blocks = read_output_rrtm(os.path.join(G, f"output_rrtm_{name}"))
    tot = oracle.column(col)
    assert abs(tot["totuflux"][0] - blocks[0]["uflx"][0]) < 2e-3
    assert tot["totdflux"][-1] == 0.0 and blocks[0]["dflx"][-1] == 0.0
