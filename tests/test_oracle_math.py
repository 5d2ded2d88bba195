"""The oracle's portable exp / pow against the container's glibc libm -- the libm the reference's Rust
f32::exp / f32::powf resolve to on Linux.  Bar: never more than 1 ULP apart, bit-identical on all but a
small fraction of inputs (glibc documents <= 0.502 ULP for expf; ours is correctly rounded up to ~1e-9)."""
import ctypes

import numpy as np

import oracle_binding as ob

libm = ctypes.CDLL("libm.so.6")
libm.expf.argtypes = [ctypes.c_float]
libm.expf.restype = ctypes.c_float
libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
libm.powf.restype = ctypes.c_float


def ulp_diff(a, b):
    ia = np.float32(a).view(np.int32).astype(np.int64)
    ib = np.float32(b).view(np.int32).astype(np.int64)
    return abs(int(ia) - int(ib))


def test_expf_within_one_ulp_of_libm():
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-104, 89, 20000), rng.uniform(-12, 12, 20000), rng.normal(0, 1, 10000),
                         np.linspace(-103.9, -86, 2000)]).astype(np.float32)     # incl. the subnormal range
    worst, mism = 0, 0
    for x in xs:
        a, b = ob.expf(x), libm.expf(float(x))
        if a != b:
            mism += 1
            worst = max(worst, ulp_diff(a, b))
    assert worst <= 1
    assert mism / len(xs) < 0.005, f"{mism} of {len(xs)} differ from libm"


def test_expf_special_values():
    assert ob.expf(0.0) == 1.0
    assert ob.expf(np.float32(-0.0)) == 1.0
    assert ob.expf(1.0) == np.float32(np.e)
    assert ob.expf(100.0) == np.inf and ob.expf(88.8) == np.inf
    assert ob.expf(-200.0) == 0.0
    assert np.isnan(ob.expf(np.nan))
    assert ob.expf(88.7) == libm.expf(88.7)            # largest finite decade
    assert ob.expf(-103.0) == libm.expf(-103.0)        # subnormal result


def test_pow3_pow4_match_libm_powf():
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.uniform(0, 1, 20000), rng.uniform(-2, 2, 5000), rng.uniform(0, 1e-3, 2000)]).astype(np.float32)
    L = ob.lib()
    bad3 = sum(1 for x in xs if L.snn_o_pow3f_export(float(x)) != libm.powf(float(x), 3.0))
    bad4 = sum(1 for x in xs if L.snn_o_pow4f_export(float(x)) != libm.powf(float(x), 4.0))
    # both sides are "correctly rounded except for double rounding / <0.52 ULP": they may differ on a
    # vanishing fraction of inputs, never by more than 1 ULP
    assert bad3 / len(xs) < 1e-3 and bad4 / len(xs) < 1e-3
    for x in xs[:3000]:
        assert ulp_diff(L.snn_o_pow3f_export(float(x)), libm.powf(float(x), 3.0)) <= 1
        assert ulp_diff(L.snn_o_pow4f_export(float(x)), libm.powf(float(x), 4.0)) <= 1


def test_synthetic_generator_twins_agree():
    """oracle C, the numpy twin in the binding and the product's numpy twin generate the same stream."""
    import snn_amd
    a = ob.uniform_array(7, 1000, -65.0, 30.0, offset=123)
    b = np.array([ob.uniform(7, 123 + i, -65.0, 30.0) for i in range(1000)], np.float32)
    c = snn_amd.synthetic.uniform(7, 1000, -65.0, 30.0, offset=123)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.array_equal(a.view(np.uint32), c.view(np.uint32))
    assert a.min() >= -65.0 and a.max() < 30.0


def test_generated_model_functions_within_one_ulp_of_libm():
    """tanh / sinh / cosh / sin / cos / tan of generated models (the reference forwards them to libm,
    build_test/nb_macro/src/lib.rs:9152-9175) and integer powers against glibc's float functions.  Ours are the correctly rounded values
    (test_modelgen_channels / test_gpu_modelgen compare them with binary64 results); glibc documents up to 2 ULP for
    the hyperbolic functions and tanf, 1 ULP for sinf / cosf -- so that is the distance allowed here."""
    rng = np.random.default_rng(2)
    xs = np.concatenate([rng.uniform(-12, 12, 6000), rng.uniform(-0.1, 0.1, 1500), rng.uniform(-100, 100, 3000),
                         rng.uniform(-1e5, 1e5, 1500), [0.0, -0.0, 0.05, -0.05, 20.5, -45.0]]).astype(np.float32)
    for name, bar in (("tanhf", 2), ("sinhf", 2), ("coshf", 2), ("sinf", 1), ("cosf", 1), ("tanf", 2)):
        ref = getattr(libm, name)
        ref.argtypes, ref.restype = [ctypes.c_float], ctypes.c_float
        ours = getattr(ob, name)
        worst, mism = 0, 0
        for x in xs:
            if name in ("sinhf", "coshf") and abs(x) > 88.0:
                continue
            a, b = np.float32(ours(x)), np.float32(ref(float(x)))
            if a.view(np.uint32) != b.view(np.uint32):
                mism += 1
                worst = max(worst, ulp_diff(a, b))
        assert worst <= bar, (name, worst)
        assert mism / len(xs) < 0.25, f"{name}: {mism} of {len(xs)} differ from libm"
    for n in (2, 3, 4, 7, -1, -2):
        for x in xs[:2000]:
            if x == 0 and n < 0:
                continue
            assert ulp_diff(ob.powif(x, n), libm.powf(float(x), float(n))) <= 1, (x, n)
    assert np.isnan(ob.sinf(np.inf)) and np.isnan(ob.tanf(np.nan)) and ob.coshf(200.0) == np.inf
    assert np.signbit(np.float32(ob.sinf(np.float32(-0.0)))) and ob.tanhf(50.0) == 1.0
