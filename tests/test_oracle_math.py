"""The oracle's expf / powf against the container's glibc libm -- the libm the reference's Rust f32::exp / f32::powf
resolve to on Linux.  Bar: BIT-IDENTICAL on every input.  oracle/check_libm.c walks all 2^32 bit patterns of x for expf,
powf(x, 3.) and powf(x, 4.) and 2^28 sampled (x, y) pairs; this module runs it and adds known values."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import oracle_binding as ob

libm = ctypes.CDLL("libm.so.6")
libm.expf.argtypes = [ctypes.c_float]
libm.expf.restype = ctypes.c_float
libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
libm.powf.restype = ctypes.c_float

ORACLE_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")


def ulp_diff(a, b):
    ia = np.float32(a).view(np.int32).astype(np.int64)
    ib = np.float32(b).view(np.int32).astype(np.int64)
    return abs(int(ia) - int(ib))


def has_fma():
    try:
        return " fma " in open("/proc/cpuinfo").read()
    except OSError:
        return True


def run_check(*args):
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)
    r = subprocess.run([os.path.join(ORACLE_DIR, "_build", "check_libm")] + [str(a) for a in args],
                       capture_output=True, text=True)
    lines = {ln.split()[0]: dict(kv.split("=") for kv in ln.split()[1:]) for ln in r.stdout.splitlines() if ln}
    return r, lines


@pytest.mark.skipif(not has_fma(), reason="glibc selects its non-FMA expf/powf on this CPU; the oracle restates the FMA build")
def test_expf_pow3_pow4_exhaustive_against_libm():
    """All 2^32 inputs of expf, powf(x, 3.), powf(x, 4.) and 2^28 sampled (x, y) pairs of powf: zero mismatches
    (about half a minute on 8 cores)."""
    r, lines = run_check("all", 1)
    assert r.returncode == 0, r.stdout + r.stderr
    for name in ("expf", "pow3", "pow4"):
        assert int(lines[name]["checked"]) == 1 << 32 and int(lines[name]["mismatches"]) == 0, r.stdout
    assert int(lines["powf"]["checked"]) >= 1 << 28 and int(lines["powf"]["mismatches"]) == 0, r.stdout


def test_array_entry_point_matches_scalar_and_libm():
    first, n, stride = 0x3D000000, 4096, 40009
    for which, ref in ((0, lambda x: libm.expf(x)), (1, lambda x: libm.powf(x, 3.0)), (2, lambda x: libm.powf(x, 4.0)),
                       (3, lambda x: libm.powf(x, -2.0))):
        got = ob.math_bits(which, first, n, stride, y=-2.0)
        xs = ((first + np.arange(n, dtype=np.uint64) * stride) & 0xFFFFFFFF).astype(np.uint32).view(np.float32)
        want = np.array([ref(float(x)) for x in xs], np.float32)
        ok = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
        assert ok.all(), which


def test_expf_special_values():
    assert ob.expf(0.0) == 1.0
    assert ob.expf(np.float32(-0.0)) == 1.0
    assert ob.expf(1.0) == np.float32(np.e)
    assert ob.expf(100.0) == np.inf and ob.expf(88.8) == np.inf
    assert ob.expf(-200.0) == 0.0 and ob.expf(-np.inf) == 0.0 and ob.expf(np.inf) == np.inf
    assert np.isnan(ob.expf(np.nan))
    assert ob.expf(88.7) == libm.expf(88.7)            # largest finite decade
    assert ob.expf(-103.0) == libm.expf(-103.0)        # subnormal result
    assert ob.expf(-103.5) == np.float32(2.0 ** -149)  # glibc's may-underflow branch


def test_powf_special_values_follow_libm():
    vals = [0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 2.0, -2.0, 3.0, -3.0, 4.0, 1e-40, -1e-40, 0.75, 1e30, -1e30, np.inf, -np.inf, np.nan]
    for x in vals:
        for y in vals:
            a, b = np.float32(ob.powf(x, y)), np.float32(libm.powf(float(np.float32(x)), float(np.float32(y))))
            assert (np.isnan(a) and np.isnan(b)) or a.view(np.uint32) == b.view(np.uint32), (x, y, a, b)


def test_integer_power_literals_fold_as_llvm_folds_them():
    """powf(x, 2.) -> x * x, powf(x, 1.) -> x, powf(x, 0.) -> 1, powf(x, -1.) -> 1 / x; everything else is libm powf."""
    rng = np.random.default_rng(3)
    for x in rng.uniform(-3, 3, 500).astype(np.float32):
        assert np.float32(ob.powif(x, 2)) == np.float32(x * x)
        assert np.float32(ob.powif(x, 1)) == x and ob.powif(x, 0) == 1.0
        assert np.float32(ob.powif(x, -1)) == np.float32(np.float32(1.0) / x)
        for n in (3, 4, 7, -2, -3):
            assert np.float32(ob.powif(x, n)).view(np.uint32) == np.float32(libm.powf(float(x), float(n))).view(np.uint32)


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
    build_test/nb_macro/src/lib.rs:9152-9175) against glibc's float functions.  Ours are the correctly rounded values
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
    assert np.isnan(ob.sinf(np.inf)) and np.isnan(ob.tanf(np.nan)) and ob.coshf(200.0) == np.inf
    assert np.signbit(np.float32(ob.sinf(np.float32(-0.0)))) and ob.tanhf(50.0) == 1.0
