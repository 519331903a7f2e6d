"""Full comparison with the reference's 13 golden OUTPUT_RRTM files (run_examples_std_atm) - needs the REAL
absorption coefficients.  The reference mount strips them (.MISSING_LARGE_BLOBS); convert either distribution with
`python -m rrtmg_lw_amd.kdata <rrtmg_lw_k_g.f90 | rrtmg_lw.nc> data/rrtmg_lw.kdata.bin` and these tests switch on.
Bars: |dflux| <= 0.01 W m-2, |dhr| <= 0.001 K/day (BASELINE.json north_star), on the oracle here and on the HIP
path in the -m gpu variant.
"""
import os

import numpy as np
import pytest

from rrtmg_lw_amd import api
from rrtmg_lw_amd.io_rrtm import read_input_rrtm, read_output_rrtm

G = os.path.join(os.path.dirname(__file__), "golden")
needs_kdata = pytest.mark.skipif(not os.path.exists(api.REAL_KDATA),
                                 reason="parity unpinned: real k-data (rrtmg_lw_k_g.f90 / rrtmg_lw.nc) is not available")
CASES = [("MLS-clr", "input_rrtm_MLS-clr", None, None), ("MLS-clr-aer12", "input_rrtm_MLS-clr-aer12", None, "in_aer_rrtm-aer12"),
         ("MLS-clr-idrv1", "input_rrtm_MLS-clr-idrv1", None, None), ("MLS-clr-xsec", "input_rrtm_MLS-clr-xsec", None, None),
         ("MLS-cld5-imca0-icld2", "input_rrtm_MLS-cld-imca0-icld2", "in_cld_rrtm-cld5", None),
         ("MLW-clr", "input_rrtm_MLW-clr", None, None), ("SAW-clr", "input_rrtm_SAW-clr", None, None), ("TROP-clr", "input_rrtm_TROP-clr", None, None)]


def _run(engine_column, name, inp, cld, aer):
    j = lambda n: os.path.join(G, n) if n else None
    col = read_input_rrtm(j(inp), j(cld), j(aer))
    blocks = read_output_rrtm(j(f"output_rrtm_{name}"))
    for i, blk in enumerate(blocks):
        o = engine_column(col, 1, 16, 0) if i == 0 else engine_column(col, i, i, 99)
        up, dn, htr = o["totuflux"].copy(), o["totdflux"], o["htr"]
        if col["idrv"] == 1:        # the driver adjusts by dtbound before printing, src/rrtmg_lw.1col.f90:587-610
            up = up + o["dtotuflux_dt"] * col["dtbound"]
        assert np.abs(up - blk["uflx"]).max() <= 0.01 + 1e-4
        assert np.abs(dn - blk["dflx"]).max() <= 0.01 + 1e-4
        if col["idrv"] != 1:
            assert np.abs(htr - blk["htr"]).max() <= 0.001 + 1e-5


@needs_kdata
@pytest.mark.parametrize("name,inp,cld,aer", CASES, ids=[c[0] for c in CASES])
def test_oracle_matches_golden(name, inp, cld, aer):
    from oracle.bindings import Oracle
    o = Oracle(kdata=api.REAL_KDATA)
    _run(lambda c, a, b, io: o.column(c, a, b, io), name, inp, cld, aer)


@needs_kdata
@pytest.mark.gpu
@pytest.mark.parametrize("name,inp,cld,aer", CASES, ids=[c[0] for c in CASES])
def test_hip_matches_golden(name, inp, cld, aer):
    api.rrtmg_lw_ini(1004.0, kdata=api.REAL_KDATA, device=0)
    _run(lambda c, a, b, io: {k: v[0] for k, v in api.run_columns([c], a, b).items()}, name, inp, cld, aer)
