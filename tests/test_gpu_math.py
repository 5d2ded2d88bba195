"""The stepper's device functions (exp, pow3, pow4 -- csrc/snn_math.hpp) against the oracle's, on the GPU:
bit-identical on every input, so transcendental-bearing models (HH, NMDA, Destexhe, STDP, DeltaDirac)
can be held to the same bit-exact bar as the Izhikevich path."""
import numpy as np
import pytest

import oracle_binding as ob

pytestmark = pytest.mark.gpu


def oracle_map(fn, xs):
    return np.array([fn(float(x)) for x in xs], np.float32)


def test_exp_bit_identical(snn):
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-110, 95, 300000), rng.uniform(-12, 12, 300000), rng.normal(0, 1, 100000),
                         np.linspace(-104.5, -85, 20000), [0.0, -0.0, 1.0, 88.72, 88.73, 89.0, 89.1, -103.97, -104.0,
                                                           -104.1, np.inf, -np.inf, np.nan]]).astype(np.float32)
    got = snn.probe_math(0, xs)
    L = ob.lib()
    want = oracle_map(L.snn_o_expf_export, xs)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_pow3_pow4_bit_identical(snn):
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.uniform(0, 1, 200000), rng.uniform(-3, 3, 50000), rng.uniform(0, 1e-12, 1000),
                         [0.0, 1.0, np.inf, np.nan]]).astype(np.float32)
    L = ob.lib()
    assert np.array_equal(snn.probe_math(1, xs).view(np.uint32), oracle_map(L.snn_o_pow3f_export, xs).view(np.uint32))
    assert np.array_equal(snn.probe_math(2, xs).view(np.uint32), oracle_map(L.snn_o_pow4f_export, xs).view(np.uint32))
