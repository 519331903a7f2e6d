"""Host logic of the Mersenne-Twister jump-ahead (rrtmg_lw_amd/csrc/mtjump.hpp): characteristic polynomial and x^n mod phi against
single steps of the recurrence.  The device side (k_mt_jump / k_mt_fill) is covered by tests/test_hip_mcica.py."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_mtjump_polynomials(tmp_path):
    exe = str(tmp_path / "mtjump_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "rrtmg_lw_amd", "csrc"),
                    os.path.join(ROOT, "tests", "cpp", "mtjump_check.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert "degree 19937, 135 terms" in out
    assert out.count(": 0 words differ") == 7, out


def test_kissvec_jump_constants(tmp_path):
    """rrtmg_lw_amd/csrc/kissjump.hpp: composition of the congruential generator, GF(2) matrix of the xorshift, residues of the
    multiply-with-carry pair - against single steps of kissvec."""
    exe = str(tmp_path / "kissjump_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "rrtmg_lw_amd", "csrc"),
                    os.path.join(ROOT, "tests", "cpp", "kissjump_check.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert "kissvec jump: 0 of 360 cases differ" in out, out
