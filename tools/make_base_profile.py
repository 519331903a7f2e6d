#!/usr/bin/env python3
"""Extract the mid-latitude-summer base profile used by the synthetic benchmark columns.

Source: the reference's example input run_examples_std_atm/input_rrtm_MLS-clr (data, BSD-3).  Output:
rrtmg_lw_amd/data/mls_base.bin with level pressure/temperature, layer pressure/temperature and the
seven volume mixing ratios (H2O, CO2, O3, N2O, CO, CH4, O2) of the 51-layer column.
Run:  python tools/make_base_profile.py [/root/reference]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from rrtmg_lw_amd.blob import write_blob  # noqa: E402
from rrtmg_lw_amd.io_rrtm import read_input_rrtm  # noqa: E402


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    col = read_input_rrtm(os.path.join(ref, "run_examples_std_atm", "input_rrtm_MLS-clr"))
    vmr = col["wkl"] / col["coldry"][None, :]          # back to mixing ratio w.r.t. dry air
    dst = os.path.join(os.path.dirname(HERE), "rrtmg_lw_amd", "data", "mls_base.bin")
    write_blob(dst, dict(pz=col["pz"], tz=col["tz"], pavel=col["pavel"], tavel=col["tavel"], vmr=vmr,
                         tbound=np.array([col["tbound"]])))
    print("wrote", dst)


if __name__ == "__main__":
    main()
