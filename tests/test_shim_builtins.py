"""The 22 OpenCL built-ins the reference build links against (oracle/ref_shim.cpp over
oracle/cl_scalar.h) checked against INDEPENDENT definitions: numpy float32 arithmetic for the
functions the OpenCL 1.2 specification defines operation by operation, numpy longdouble (x87
80-bit) and the host libm for the transcendental ones, over dense random and edge inputs.

This does not turn the shim into the reference's runtime (no x86 OpenCL built-in library exists
in this image -- parity stays "unpinned" at that root); it shows that every
stand-in is a correct implementation of its specified function, to the bit where the
specification is exact and within the specification's own ULP bound where it is not.
Build container only (needs oracle/_ref)."""
import numpy as np
import pytest

pytestmark = pytest.mark.reference

F = np.float32
RNG = np.random.default_rng(2024)


def _ulps(a, want):
    """distance in float32 ulps between a (float32) and `want` (any precision), measured at want"""
    want32 = np.asarray(want).astype(F)
    ulp = np.spacing(np.abs(want32)).astype(np.float64)
    return np.abs(a.astype(np.float64) - np.asarray(want, dtype=np.float64)) / np.maximum(ulp, 1e-300)


def _vec(n, lo=-4.0, hi=4.0):
    return RNG.uniform(lo, hi, (n, 3)).astype(F)


def _specials():
    return np.array([0.0, -0.0, 1.0, -1.0, 0.5, 1e-30, -1e-30, 1e30, -1e30, np.inf, -np.inf, 3.5, -3.5,
                     2147483520.0, 2147483648.0, -2147483648.0, -2147483904.0, 4294967296.0, 16777217.0],
                    dtype=F)


@pytest.fixture(scope="module")
def sh(oracle_mod):
    if not oracle_mod.have_ref():
        pytest.skip("oracle/_ref is not built")
    return oracle_mod.shim_eval


def test_conversions(sh):
    ints = np.concatenate([RNG.integers(-2**31, 2**31 - 1, (3000, 3)), [[0, 1, -1], [2**31 - 1, -2**31, 16777217]]]).astype(np.int32)
    got = sh("convert_float3", ints.view(F), out_shape=ints.shape)
    assert np.array_equal(got, ints.astype(F))  # round to nearest even
    x = np.concatenate([_vec(4000, -3e9, 3e9), _vec(2000, -5, 5), np.resize(_specials(), (7, 3)),
                        np.full((1, 3), np.nan, F)])
    got = sh("convert_int3_sat", x, out_shape=x.shape, out_dtype=np.int32)
    x64 = x.astype(np.float64)
    want = np.where(np.isnan(x64), 0.0, np.clip(np.trunc(x64), -2.0**31, 2.0**31 - 1)).astype(np.int64)
    assert np.array_equal(got.astype(np.int64), want)  # OpenCL 6.2.3.3: _sat, round toward zero, NaN -> 0


def test_operation_by_operation_builtins_are_bit_exact(sh):
    """min / max / clamp / step / mix / mad / fabs / sqrt: the specification gives the formula."""
    n = 6000
    a, b, c = (RNG.uniform(-8, 8, n).astype(F) for _ in range(3))
    a[:19], b[:19] = _specials(), _specials()[::-1]
    with np.errstate(all="ignore"):
        assert np.array_equal(sh("min", a, b).view(np.uint32), np.where(b < a, b, a).view(np.uint32))
        assert np.array_equal(sh("max", a, b).view(np.uint32), np.where(a < b, b, a).view(np.uint32))
        lo, hi = np.minimum(b, c), np.maximum(b, c)
        assert np.array_equal(sh("clamp", a, lo, hi), np.minimum(np.maximum(a, lo), hi))
        assert np.array_equal(sh("step", a, b), np.where(b < a, F(0), F(1)))
        assert np.array_equal(sh("fabs", a).view(np.uint32), np.abs(a).view(np.uint32))
        pos = np.abs(a)
        assert np.array_equal(sh("sqrt", pos), np.sqrt(pos))  # IEEE: correctly rounded
        # mad: two roundings (a*b rounded, then + c) -- and NOT a fused multiply-add
        two = (a * b).astype(F) + c
        got = sh("mad", a, b, c)
        assert np.array_equal(got.view(np.uint32), two.astype(F).view(np.uint32))
        fused = (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F)
        assert (fused != two).any(), "inputs never separate fused from unfused: test is vacuous"
    va, vb, vc = _vec(n), _vec(n), _vec(n)
    t = RNG.uniform(-0.5, 1.5, n).astype(F)
    assert np.array_equal(sh("min3", va, vb), np.where(vb < va, vb, va))
    assert np.array_equal(sh("max3", va, vb), np.where(va < vb, vb, va))
    assert np.array_equal(sh("mad3", va, vb, vc), ((va * vb).astype(F) + vc).astype(F))
    assert np.array_equal(sh("mix3", va, vb, vc), (va + ((vb - va).astype(F) * vc).astype(F)).astype(F))
    assert np.array_equal(sh("mix3s", va, vb, t), (va + ((vb - va).astype(F) * t[:, None]).astype(F)).astype(F))
    # cross: the component formula of 6.12.5 in float32
    cr = np.stack([(va[:, 1] * vb[:, 2]).astype(F) - (va[:, 2] * vb[:, 1]).astype(F),
                   (va[:, 2] * vb[:, 0]).astype(F) - (va[:, 0] * vb[:, 2]).astype(F),
                   (va[:, 0] * vb[:, 1]).astype(F) - (va[:, 1] * vb[:, 0]).astype(F)], axis=1).astype(F)
    assert np.array_equal(sh("cross", va, vb), cr)


def test_geometric_builtins_against_extended_precision(sh):
    n = 8000
    va = np.concatenate([_vec(n), _vec(500, -1e-3, 1e-3), _vec(500, -1e4, 1e4)])
    vb = np.concatenate([_vec(n), _vec(500, -1e-3, 1e-3), _vec(500, -1e4, 1e4)])
    la, lb = va.astype(np.longdouble), vb.astype(np.longdouble)
    d = sh("dot", va, vb, out_shape=(va.shape[0],))
    exact = (la * lb).sum(axis=1)
    scale = (np.abs(la * lb)).sum(axis=1)  # a valid float32 dot product errs by a few ulps of this
    assert (np.abs(d.astype(np.longdouble) - exact) <= 3 * np.spacing(scale.astype(F)).astype(np.longdouble)).all()
    ln = sh("length", va, out_shape=(va.shape[0],))
    assert _ulps(ln, np.sqrt((la * la).sum(axis=1))).max() <= 2.0
    nz = sh("normalize", va)
    want = la / np.sqrt((la * la).sum(axis=1))[:, None]
    assert _ulps(nz, want).max() <= 2.5
    zero = np.zeros((4, 3), F)
    zero[1, 0] = -0.0
    assert np.array_equal(sh("normalize", zero).view(np.uint32), zero.view(np.uint32))  # F7: 0 stays 0, no NaN
    # the restatement's own left-to-right float32 dot is what the shim computes (bit for bit)
    d32 = (((va[:, 0] * vb[:, 0]).astype(F) + (va[:, 1] * vb[:, 1]).astype(F)).astype(F) + (va[:, 2] * vb[:, 2]).astype(F)).astype(F)
    assert np.array_equal(d, d32)


def test_transcendentals_against_longdouble_and_libm(sh, oracle_mod):
    """exp / exp2 / pow: deterministic double-precision evaluation rounded once.  OpenCL 1.2 allows
    3 / 3 / 16 ulp (7.4); these stay within 1 ulp of the extended-precision value, and of the
    host libm the second reference build (libref_oracle_libm.so) calls."""
    x = np.concatenate([RNG.uniform(-87, 88, 20000), RNG.uniform(-1, 1, 5000), RNG.uniform(-104, -87, 2000),
                        [0.0, -0.0, 1.0, -1.0, 88.7, -103.9, 1e-20]]).astype(F)
    with np.errstate(all="ignore"):
        e = sh("exp", x)
        assert _ulps(e, np.exp(x.astype(np.longdouble)))[np.isfinite(e) & (e > 1e-37)].max() <= 0.5001 + 1e-3
        x2 = np.concatenate([RNG.uniform(-126, 127, 20000), RNG.uniform(-149, -126, 2000), np.arange(-30, 31)]).astype(F)
        e2 = sh("exp2", x2)
        assert _ulps(e2, np.exp2(x2.astype(np.longdouble)))[e2 > 1e-37].max() <= 0.5001 + 1e-3
        assert np.array_equal(sh("exp2", np.arange(-20, 21).astype(F)), np.exp2(np.arange(-20, 21)).astype(F))
        # pow as the reference uses it (renderer.cl:320-322): nh in (0, 1], exponent 2^(6 s + 4), s in [0, 1]
        nh = np.concatenate([RNG.uniform(1e-4, 1.0, 20000), 1.0 - RNG.uniform(0, 1e-3, 4000), [1.0, 0.5, 0.999999]]).astype(F)
        ex = np.exp2(RNG.uniform(4, 10, nh.size)).astype(F)
        ex[-3:] = [16.0, 1024.0, 1024.0]
        p = sh("pow", nh, ex)
        want = np.power(nh.astype(np.longdouble), ex.astype(np.longdouble))
        ok = p > 1e-37
        assert _ulps(p, want)[ok].max() <= 0.5001 + 2e-3
        # general quadrant x > 0
        bx = RNG.uniform(1e-3, 50, 10000).astype(F)
        by = RNG.uniform(-8, 8, 10000).astype(F)
        pg = sh("pow", bx, by)
        wg = np.power(bx.astype(np.longdouble), by.astype(np.longdouble))
        assert _ulps(pg, wg)[np.isfinite(pg) & (pg > 1e-37)].max() <= 0.5001 + 2e-3
        # special values (C99 F.9.4.4 for the quadrant the kernel can reach)
        assert sh("pow", np.array([0.5, 0.0, 2.0, 1.0], F), np.array([0.0, 2.0, 0.0, 1e30], F)).tolist() == [1.0, 0.0, 1.0, 1.0]
        assert sh("exp", np.array([-1e4, 1e4], F)).tolist() == [0.0, np.inf]
        if oracle_mod.have_ref("libm"):
            for name, args in (("exp", (x,)), ("exp2", (x2,)), ("pow", (nh, ex))):
                a = oracle_mod.shim_eval(name, *args)
                b = oracle_mod.shim_eval(name, *args, fma="libm")
                fin = np.isfinite(a) & (a > 1e-37)
                assert _ulps(a, b.astype(np.float64))[fin].max() <= 1.0, name


def test_get_global_id(sh):
    ids = np.array([0, 1, 12345, 921599], F)
    assert np.array_equal(sh("get_global_id", ids), ids)
