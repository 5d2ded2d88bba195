"""The committed golden vectors are what the numpy restatement (tests/numpy_net.py: host libm for exp / powf, no code
shared with the C oracle) produces -- tests/golden/make_golden.py writes them with it.  Together with
tests/test_golden_oracle.py (C oracle == fixtures) and tests/test_gpu_golden.py (HIP == fixtures) every vector pins two
independent CPU restatements and the device."""
import os
import sys

import numpy as np
import pytest

import golden_cases
import numpy_net

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLDEN)


@pytest.mark.parametrize("name", sorted(golden_cases.CASES))
def test_numpy_restatement_writes_the_committed_vectors(name):
    import make_golden
    got, _ = make_golden.generate(name)
    want = np.load(os.path.join(GOLDEN, name + ".npz"))
    assert sorted(got) == sorted(want.files)
    for k in want.files:
        a, b = np.asarray(got[k]), want[k]
        assert a.dtype == b.dtype and a.shape == b.shape, k
        assert a.tobytes() == b.tobytes(), f"{name}.{k}"


def test_accumulate_is_the_sequential_sum():
    """np.add.accumulate in float32 adds row after row with one rounding each -- the canonical order inside a chunk"""
    rng = np.random.default_rng(5)
    t = (rng.standard_normal((300, 7)) * 10.0 ** rng.integers(-3, 4, (300, 7))).astype(np.float32)
    want = np.zeros(7, np.float32)
    for row in t:
        want = (want + row).astype(np.float32)
    assert numpy_net.sequential_column_sums(t).tobytes() == want.tobytes()
    total = np.zeros(7, np.float32)
    for c0 in range(0, 300, 256):
        part = np.zeros(7, np.float32)
        for row in t[c0:c0 + 256]:
            part = (part + row).astype(np.float32)
        total = (total + part).astype(np.float32)
    assert numpy_net.chunked_column_sums(t).tobytes() == total.tobytes()


def test_host_libm_is_what_the_oracle_restates():
    """the restatement's exp / powf are libm's own; the oracle's are a restated algorithm -- same bits on samples
    (exhaustively: tests/test_oracle_math.py)"""
    import oracle_binding as ob
    rng = np.random.default_rng(6)
    x = (rng.standard_normal(2000) * 20).astype(np.float32)
    assert numpy_net.expf(x).tobytes() == np.array([ob.expf(v) for v in x], np.float32).tobytes()
    g = rng.random(2000).astype(np.float32)
    for y in (3.0, 4.0):
        assert numpy_net.powf(g, np.float32(y)).tobytes() == np.array([ob.powf(v, y) for v in g], np.float32).tobytes()
