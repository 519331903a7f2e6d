"""A short, seeded version of tools/soak_host_entry.py: random column ranges of one set of columns, random subsets of the arrays pinned,
several calls back to back - every result must equal, bit for bit, the same columns of one reference call.  The long form found three
defects in round 3 (a level's last bit depending on the block a column shares; pageable user pointers handed to the runtime; arrays
taken for pinned because their ends share pages with pinned neighbours)."""
import numpy as np
import pytest

from rrtmg_lw_amd.synth import make_gcm_inputs

pytestmark = pytest.mark.gpu
NAMES = ("uflx", "dflx", "hr", "uflxc", "dflxc", "hrc", "duflx_dt", "duflxc_dt")


@pytest.mark.parametrize("mode", ["nomcica", "fused_mcica"])
def test_random_ranges_and_pinned_subsets(hip, mode):
    nmax, nlay = 24000, 40
    full = make_gcm_inputs(nmax, nlay, "aer_idrv", col0=7)
    for k in ("co2vmr", "o2vmr"):
        full[k] = np.asfortranarray(np.full((nmax, nlay), float(np.asarray(full[k])[0, 0])))
    if mode == "nomcica":
        solve = lambda d: hip.rrtmg_lw_from_dict(d, icld=2)
    else:
        solve = lambda d: hip.rrtmg_lw_mcica_subcol_from_dict(d, 140, 0, icld=2)
    ref = solve(full)
    rng = np.random.default_rng(5)
    for it in range(14):
        n = int(rng.integers(1, nmax + 1)) if it % 3 else int(rng.integers(1, 300))
        c0 = int(rng.integers(0, nmax - n + 1))
        d = dict(full)
        d["ncol"] = n
        for k, v in full.items():
            if isinstance(v, np.ndarray):
                d[k] = np.asfortranarray(v[:, c0:c0 + n, :] if (v.ndim == 3 and v.shape[0] == 16) else v[c0:c0 + n])
        pinned = [v for v in d.values() if isinstance(v, np.ndarray) and v.nbytes >= 4096 and rng.random() < 0.5]
        for v in pinned:
            hip.host_register(v)
        try:
            for _ in range(int(rng.integers(1, 3))):
                got = solve(d)
                for k in NAMES:
                    assert np.array_equal(got[k], ref[k][c0:c0 + n]), (mode, it, k, n, c0)
        finally:
            for v in pinned:
                hip.host_unregister(v)
