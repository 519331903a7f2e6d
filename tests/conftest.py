import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built by __graft_entry__.build() / `make -C oracle liboracle.so`."""
    from oracle.bindings import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def hip():
    """The product library, initialised on cuda:0 with the stand-in k-data.  Fails loudly if it is not built."""
    # tests that hand torch tensors to the device entries need ONE HIP runtime in the process: torch bundles its own
    # libamdhip64, so it has to be loaded before librrtmg_lw_hip.so pulls in the system one (bench.py does the same)
    import torch  # noqa: F401
    from rrtmg_lw_amd import api
    api.rrtmg_lw_ini(1004.0, kdata=api.STANDIN_KDATA, device=0)
    yield api
    api.finalize()


@pytest.fixture(params=["three sweep launches", "one sweep launch", "columns taken by cloud top"])
def sweeps(request, hip):
    """Cloudy batches of up to 4096 columns take one sweep launch per band group (the cloud-zone kernel over all levels) instead of three
    (rrtmg_lw_hip_set_one_sweep_max).  The tests' column counts are below that, so the tests that pin the sweeps - the reference fixtures,
    the cloud-structure cases, the fuzz - run both ways: the three launches are what every production-size batch takes.  Third way: every
    window of 256 columns reordered by cloud top (rrtmg_lw_hip_set_column_sort with threshold 0; by default only windows where that pays
    are, which small test inputs rarely reach), three launches."""
    prev = hip.set_one_sweep_max(1 << 30 if request.param.startswith("one") else 0)
    prev_min = hip.column_sort_min()
    prev_sort = hip.set_column_sort(True, 0 if request.param.startswith("columns") else 1 << 24)
    yield request.param
    hip.set_one_sweep_max(prev)
    hip.set_column_sort(prev_sort, prev_min)
