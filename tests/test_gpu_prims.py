"""Device float primitives vs the host: the parity contract (same IEEE op
sequence on x86 and gfx950) rests on these being bit-identical."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _inputs(seed, n=1 << 16):
    rng = np.random.default_rng(seed)
    a = np.concatenate([
        rng.standard_normal(n // 2) * 10.0 ** rng.integers(-30, 30, n // 2),
        rng.uniform(-4, 4, n // 4), rng.uniform(-1e-38, 1e-38, n // 8),  # denormals too
        rng.uniform(-3e9, 3e9, n // 8),
    ]).astype(np.float32)
    b = np.concatenate([
        rng.standard_normal(n // 2) * 10.0 ** rng.integers(-30, 30, n // 2),
        rng.uniform(-4, 4, n // 2),
    ]).astype(np.float32)
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1.0, -1.0, 2147483648.0, -2147483648.0,
                        4294967296.0, -4294967296.0, 9.3e18, -9.3e18, 1e-45, 3.4e38, 0.99999994],
                       np.float32)
    a[: special.size] = special
    b[: special.size] = special[::-1]
    return a, b


def _bits(x):
    return np.asarray(x, np.float32).view(np.uint32)


def _same(got, want):
    want = _bits(want)
    nan = np.isnan(want.view(np.float32))
    assert np.array_equal(got[~nan], want[~nan])
    assert np.isnan(got.view(np.float32)[nan]).all()


def test_div_sqrt_mad_correctly_rounded(gpu_ctx):
    a, b = _inputs(1)
    with np.errstate(all="ignore"):
        _same(gpu_ctx.selftest_prims(0, a, b), a / b)
        _same(gpu_ctx.selftest_prims(1, np.abs(a)), np.sqrt(np.abs(a)))
        _same(gpu_ctx.selftest_prims(8, a, b), (a * b).astype(np.float32) + a)  # NOT fused


def test_shared_divisor_division_is_ieee_division(gpu_ctx):
    """rmd::div_by (one reciprocal + 3 instructions per quotient, rm_detmath.hpp) == x / b:
    the general mix incl. specials (those take its IEEE fallback), and dense sweeps for
    the divisors the voxel walk uses (maxVoxelIter/2, maxVoxelIter/4)."""
    a, b = _inputs(5)
    with np.errstate(all="ignore"):
        _same(gpu_ctx.selftest_prims(9, a, b), a / b)
        rng = np.random.default_rng(6)
        for div in (96.0, 48.0, 24.0, 37.5, 3.0, 16777215.0 / 2.0, 1.9999999, 1e-31, 1e31):
            # every exponent, random significands, both signs + a dense run of consecutive floats
            x = (rng.integers(0, 1 << 32, 1 << 20, dtype=np.uint64).astype(np.uint32)).view(np.float32)
            x = np.concatenate([x, (np.uint32(0x3f000000) + np.arange(1 << 18, dtype=np.uint32)).view(np.float32)])
            d = np.full(x.size, div, np.float32)
            _same(gpu_ctx.selftest_prims(9, x, d), x / d)


def test_exp_exp2_pow_match_oracle_bitwise(gpu_ctx, oracle_mod):
    L = oracle_mod.restate_lib()
    rng = np.random.default_rng(2)
    x = np.concatenate([rng.uniform(-110, 90, 30000), rng.uniform(-1, 1, 10000),
                        [0.0, -0.0, -50000.0, 89.0, np.inf, -np.inf, np.nan]]).astype(np.float32)
    _same(gpu_ctx.selftest_prims(2, x), np.array([L.rmo_exp(float(v)) for v in x], np.float32))
    x2 = np.concatenate([rng.uniform(-152, 129, 30000), [4.0, 10.0, 7.6]]).astype(np.float32)
    _same(gpu_ctx.selftest_prims(3, x2), np.array([L.rmo_exp2(float(v)) for v in x2], np.float32))
    base = np.concatenate([rng.uniform(0, 1.0001, 30000), [0.0, 1.0, 1e-45, np.inf, -1.0, np.nan]]).astype(np.float32)
    ex = np.concatenate([np.exp2(rng.uniform(4, 10, 30000)), [16.0, 1024.0, 2.0, 3.0, 2.0, 1.0]]).astype(np.float32)
    _same(gpu_ctx.selftest_prims(4, base, ex),
          np.array([L.rmo_pow(float(p), float(q)) for p, q in zip(base, ex)], np.float32))


def test_casts_match_x86(gpu_ctx, oracle_mod):
    L = oracle_mod.restate_lib()
    a, _ = _inputs(3)
    a[100:5000] = np.random.default_rng(4).uniform(-70000, 70000, 4900).astype(np.float32)
    want_i = np.array([L.rmo_f2i(float(v)) for v in a], np.int32).view(np.uint32)
    want_u = np.array([L.rmo_f2u(float(v)) for v in a], np.uint32)
    want_s = np.array([L.rmo_convert_int_sat(float(v)) for v in a], np.int32).view(np.uint32)
    assert np.array_equal(gpu_ctx.selftest_prims(5, a), want_i)
    assert np.array_equal(gpu_ctx.selftest_prims(6, a), want_u)
    assert np.array_equal(gpu_ctx.selftest_prims(7, a), want_s)
