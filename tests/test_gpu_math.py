"""The stepper's device functions (expf, powf(x, 3.), powf(x, 4.), powf -- csrc/snn_math.hpp) against the oracle's, on
the GPU: bit-identical on EVERY binary32 input (all 2^32 bit patterns each for expf / pow3 / pow4, walked in chunks),
so transcendental-bearing models (HH, NMDA, Destexhe, STDP, DeltaDirac) are held to the same bit-exact bar as the
Izhikevich path.  The oracle's functions are in turn pinned to glibc's libm on all 2^32 inputs (test_oracle_math.py)."""
import numpy as np
import pytest

import oracle_binding as ob

pytestmark = pytest.mark.gpu

CHUNK = 1 << 26


def same_bits(a, b):
    return bool(((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all())


@pytest.mark.parametrize("which", [0, 1, 2, 4, 5, 6], ids=["expf", "pow3", "pow4", "expf_main_path", "pow3_main_path", "pow4_main_path"])
def test_every_binary32_input_bit_identical(snn, which):
    for first in range(0, 1 << 32, CHUNK):
        got = snn.probe_math_bits(which, first, CHUNK)
        want = ob.math_bits(which % 4 if which >= 4 else which, first, CHUNK)     # (4, 5, 6 are other device forms of 0, 1, 2)
        if not same_bits(got, want):
            bad = np.flatnonzero(got.view(np.uint32) != want.view(np.uint32))[:5]
            raise AssertionError(f"function {which}: bit patterns {[hex(first + int(i)) for i in bad]} differ: "
                                 f"device {got[bad]}, oracle {want[bad]}")


@pytest.mark.parametrize("y", [5.0, 7.0, -2.0, -3.0, 0.5, 2.5])
def test_general_powf_sampled(snn, y):
    """powf(x, y) of generated models (`x ^ n`, n outside {0, 1, 2, -1}): 2^26 patterns spread over the whole space"""
    n, stride = 1 << 26, 63
    assert same_bits(snn.probe_math_bits(3, 12345, n, stride, y=y), ob.math_bits(3, 12345, n, stride, y=y))


def test_array_probe_matches_bit_probe(snn):
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-110, 95, 300000), rng.uniform(-12, 12, 300000), rng.normal(0, 1, 100000),
                         np.linspace(-104.5, -85, 20000), [0.0, -0.0, 1.0, 88.72, 88.73, 89.0, 89.1, -103.97, -104.0,
                                                           -104.1, np.inf, -np.inf, np.nan]]).astype(np.float32)
    got = snn.probe_math(0, xs)
    want = np.array([ob.expf(x) for x in xs[:50000]], np.float32)
    assert same_bits(got[:50000], want)
    bits = xs.view(np.uint32)
    for which in (1, 2):
        g = snn.probe_math(which, xs)
        w = np.array([ob.math_bits(which, int(b), 1)[0] for b in bits[:20000]], np.float32)
        assert same_bits(g[:20000], w)
