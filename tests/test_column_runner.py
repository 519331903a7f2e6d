"""OUTPUT_RRTM writer and the standalone column runner (python -m rrtmg_lw_amd.column), SURVEY.md 8(f1).

CPU part: the writer reproduces every record of the reference's 14 checked-in OUTPUT_RRTM files byte for byte from the parsed
numbers (formats src/rrtmg_lw.1col.f90:737-746), and the IOUT -> block sequence follows the driver (:452-466, :689-696).
GPU part: the runner on reference inputs against the oracle run through the same driver logic."""
import glob
import os

import numpy as np
import pytest

from rrtmg_lw_amd.column import band_sequence
from rrtmg_lw_amd.io_rrtm import WAVENUM1, WAVENUM2, format_output_block, format_output_row, read_input_rrtm, read_output_rrtm, write_output_rrtm

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "output_rrtm_*"))), ids=os.path.basename)
def test_writer_reproduces_reference_records(path):
    blocks = read_output_rrtm(path)
    gen = []
    for b in blocks:
        gen += format_output_block(WAVENUM1.index(b["wn1"]) + 1, WAVENUM2.index(b["wn2"]) + 1, b["pz"], b["uflx"], b["dflx"], b["fnet"], b["htr"])
    ref = [ln for ln in open(path).read().split("\n")]
    ref = ref[:next(i for i, ln in enumerate(ref) if ln.startswith("  Modules"))]          # the version footer is the reference's own
    assert [ln.rstrip() for ln in ref] == [ln.rstrip() for ln in gen]
    assert any(ln == "\f" for ln in gen)


def test_row_formats_by_pressure():
    # one record per format 9952 .. 9958; the zero before the decimal point is dropped where the field has no room (PGI / flang alike)
    assert format_output_row(51, 0.067, 281.5358, 0.0, 281.5357819, 0.0) == "  51         .06700    281.5358        0.0000       281.5357819            0.00000"
    assert format_output_row(48, 0.1069, 281.5674, 0.0799, 281.4875530, -5.69915) == "  48        0.1069     281.5674        0.0799       281.4875530           -5.69915"
    assert format_output_row(7, 0.0012, 1.0, 2.0, -1.0, -0.00003).startswith("   7         .001200   ")
    assert format_output_row(0, 1013.0, 0.4758, 0.1813, 0.2944491, -0.00275) == "   0     1013.0          0.4758        0.1813         0.2944491           -0.00275"


def test_block_sequence_follows_iout():
    assert band_sequence(0) == [(1, 16)]
    assert band_sequence(7) == [(7, 7)]
    assert band_sequence(99) == [(1, 16)] + [(b, b) for b in range(1, 17)]
    assert band_sequence(-1) == []


def test_write_then_read_round_trip(tmp_path):
    rng = np.random.default_rng(3)
    pz = np.exp(np.linspace(np.log(1013.0), np.log(0.003), 40))
    blocks = [dict(istart=a, iend=b, pz=pz, uflx=rng.uniform(0, 450, 40), dflx=rng.uniform(0, 400, 40), fnet=rng.uniform(-50, 300, 40),
                   htr=rng.uniform(-30, 5, 40)) for a, b in band_sequence(99)]
    p = str(tmp_path / "OUTPUT_RRTM")
    write_output_rrtm(p, blocks)
    back = read_output_rrtm(p)
    assert len(back) == 17
    for b, r in zip(blocks, back):
        assert (r["wn1"], r["wn2"]) == (WAVENUM1[b["istart"] - 1], WAVENUM2[b["iend"] - 1])
        assert np.abs(r["uflx"] - b["uflx"]).max() <= 5.1e-5 and np.abs(r["dflx"] - b["dflx"]).max() <= 5.1e-5
        assert np.abs(r["fnet"] - b["fnet"]).max() <= 5.1e-8 and np.abs(r["htr"] - b["htr"]).max() <= 5.1e-6


CASES = [("input_rrtm_MLS-clr", None, None), ("input_rrtm_MLS-clr-aer12", None, "in_aer_rrtm-aer12"),
         ("input_rrtm_MLS-clr-idrv1", None, None), ("input_rrtm_MLS-cld-imca0-icld2", "in_cld_rrtm-cld5", None),
         ("input_rrtm_ICRCCM_sonde", None, None)]           # (IATM = 1: layered by rrtmg_lw_amd/atmpth.py)


@pytest.mark.gpu
@pytest.mark.parametrize("inp,cld,aer", CASES, ids=[c[0] for c in CASES])
def test_runner_matches_oracle(tmp_path, hip, oracle, inp, cld, aer):
    """The runner end to end (file in, OUTPUT_RRTM out, stand-in coefficients) against the oracle taken through the same driver steps:
    block sequence of IOUT, DTBOUND adjustment with IDRV = 1; agreement to the printed precision."""
    api = hip          # (the session fixture loads torch's HIP runtime before the library: one runtime per process, tests/conftest.py)
    from rrtmg_lw_amd.column import main
    j = lambda n: os.path.join(G, n) if n else None
    out = str(tmp_path / "OUTPUT_RRTM")
    argv = [j(inp), "-o", out, "--kdata", api.STANDIN_KDATA]
    if cld:
        argv += ["--cld", j(cld)]
    if aer:
        argv += ["--aer", j(aer)]
    try:
        assert main(argv) == 0
    finally:
        api.rrtmg_lw_ini(1004.0, kdata=api.STANDIN_KDATA, device=0)      # main() finalises the library: restore the session fixture's state
    got = read_output_rrtm(out)
    col = read_input_rrtm(j(inp), j(cld), j(aer))
    seq = band_sequence(int(col["iout"]))
    assert len(got) == len(seq) >= 1
    nl = int(col["nlayers"])
    heatfac = 1.0e-2 * 9.8066 * 8.64e4 / 1004.0
    for blk, (a, b) in zip(got, seq):
        o = oracle.column(col, a, b, 99 if a == b else 0)
        up, dn = o["totuflux"].copy(), o["totdflux"]
        htr = o["htr"].copy()
        if int(col["idrv"]) == 1:
            up = up + o["dtotuflux_dt"] * float(col["dtbound"])
            net = up - dn
            htr = np.zeros(nl + 1)
            htr[:nl] = heatfac * (net[:-1] - net[1:]) / (np.asarray(col["pz"])[:-1] - np.asarray(col["pz"])[1:])
        assert np.abs(blk["uflx"] - up).max() <= 1e-4 and np.abs(blk["dflx"] - dn).max() <= 1e-4
        assert np.abs(blk["htr"] - htr).max() <= 6e-5
