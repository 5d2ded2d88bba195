"""The oracle reproduces the committed golden vectors (tests/golden/*.npz) bit for bit."""
import os

import numpy as np
import pytest

import golden_cases

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", sorted(golden_cases.CASES))
def test_oracle_matches_golden(name):
    net, steps = golden_cases.CASES[name]()
    net.run(steps, voltage_history=True, spike_history=True, st_voltage_history=bool(net.n_cells))
    got = golden_cases.outputs(name, net, steps)
    want = np.load(os.path.join(GOLDEN, name + ".npz"))
    assert sorted(got) == sorted(want.files)
    for k in want.files:
        a, b = np.asarray(got[k]), want[k]
        assert a.dtype == b.dtype and a.shape == b.shape, k
        assert a.tobytes() == b.tobytes(), f"{name}.{k}"


def test_golden_cases_are_not_trivial():
    for name in golden_cases.CASES:
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        assert np.unpackbits(z["raster"]).sum() > 0 or name.startswith("spike_trains"), name
