"""The deterministic exp/exp2/pow and cast helpers of the oracle (the device
versions are compared with these bit-for-bit in test_gpu_prims.py)."""
import numpy as np


def _ulps(a, b):
    a = np.asarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.asarray(b, np.float32).view(np.int32).astype(np.int64)
    return np.abs(a - b)


def test_exp_exp2_pow_within_one_ulp_of_libm(oracle_mod):
    L = oracle_mod.restate_lib()
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-100, 88, 4000), rng.uniform(-1, 1, 2000), [0.0, -0.0, 1.0, -50000.0]]).astype(np.float32)
    e = np.array([L.rmo_exp(float(x)) for x in xs], np.float32)
    assert _ulps(e, np.exp(xs.astype(np.float64)).astype(np.float32)).max() <= 1
    x2 = rng.uniform(-149, 127, 4000).astype(np.float32)
    e2 = np.array([L.rmo_exp2(float(x)) for x in x2], np.float32)
    assert _ulps(e2, np.exp2(x2.astype(np.float64)).astype(np.float32)).max() <= 1
    base = rng.uniform(1e-3, 1.0, 4000).astype(np.float32)
    ex = np.exp2(rng.uniform(4, 10, 4000)).astype(np.float32)
    p = np.array([L.rmo_pow(float(b), float(y)) for b, y in zip(base, ex)], np.float32)
    want = np.power(base.astype(np.float64), ex.astype(np.float64)).astype(np.float32)
    assert _ulps(p, want).max() <= 1
    assert L.rmo_exp(-50000.0) == 0.0 and L.rmo_exp2(200.0) == np.inf and L.rmo_pow(0.5, 0.0) == 1.0


def test_cast_semantics_are_x86(oracle_mod):
    L = oracle_mod.restate_lib()
    assert L.rmo_f2i(3.9) == 3 and L.rmo_f2i(-3.9) == -3
    assert L.rmo_f2i(float("nan")) == -2**31 and L.rmo_f2i(3e9) == -2**31
    assert L.rmo_f2u(-1.0) == 2**32 - 1 and L.rmo_f2u(-2.5) == 2**32 - 2  # wraps, SURVEY F6
    assert L.rmo_f2u(4294967296.0 + 512.0) == 512 and L.rmo_f2u(float("nan")) == 0
    assert L.rmo_convert_int_sat(float("nan")) == 0 and L.rmo_convert_int_sat(1e20) == 2**31 - 1
    assert L.rmo_convert_int_sat(-1e20) == -2**31 and L.rmo_convert_int_sat(-0.9) == 0
