"""The device-contract checker of the GPU tests: the reference kernel built for gfx950 (strict build).

Two sources, in this order:
  1. LIVE   oracle/_ref/renderer_gfx950_strict.hsaco -- the unmodified renderer.cl compiled by
            oracle/Makefile in the build container (git-ignored, travels with the snapshot) and run on
            the GPU through oracle/ref_gfx950_runner.cpp;
  2. FIXED  tests/golden/gfx950_strict/ -- the outputs of exactly that code object on the same inputs,
            recorded ON the GPU by tests/golden/make_golden_gfx950.py and committed (data only): full
            float32 accumulators + ARGB words for the fixture scenes and config 1, sha256 digests (digests.json) + a
            sparse sample of pixels (digest_samples.npz) for the large frames (configs 2-5, pass-packed frames).
A clean clone on a GPU box therefore still checks every device-contract frame bit for bit; when
neither source exists the tests FAIL (they do not skip): a GPU box without a checker is a broken
checkout, not a reason to pass.
"""
import hashlib
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXED = os.path.join(ROOT, "tests", "golden", "gfx950_strict")
SAMPLE_STRIDE = 997  # pixels kept from a digest-pinned frame (prime: walks through rows and columns)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def input_digest(vox, opts, mc, n):
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(vox).tobytes())
    h.update(bytes(opts))
    h.update(np.ascontiguousarray(mc, dtype=np.float32).tobytes())
    h.update(str(int(n)).encode())
    return h.hexdigest()


def _digests():
    p = os.path.join(FIXED, "digests.json")
    return json.load(open(p)) if os.path.exists(p) else {}


class Checker:
    def __init__(self, oracle_mod):
        self.oracle = oracle_mod
        self.live = bool(oracle_mod.have_gfx950_ref("strict")) and os.environ.get("RAYMARCH_PIN_FIXED_ONLY", "0") != "1"

    def source(self):
        return "live reference build" if self.live else "committed fixtures"

    def frame(self, key, vox, opts, mc, n):
        """-> (pixels float32[4n], argb uint32[n]) of the strict reference build for these inputs;
        only for keys whose full output is on file (fixture scenes, c1)."""
        if self.live:
            px, argb, _ = self.oracle.gfx950_render_frame(vox, opts, mc, n, build="strict")
            return px, argb
        path = os.path.join(FIXED, key + ".npz")
        if not os.path.exists(path):
            pytest.fail(f"no device-contract checker for `{key}`: neither oracle/_ref/renderer_gfx950_strict.hsaco "
                        f"nor {os.path.relpath(path, ROOT)} exists")
        z = np.load(path)
        assert str(z["inputs"]) == input_digest(vox, opts, mc, n), f"fixture {key} was recorded for other inputs"
        return z["pixels"].copy(), z["argb"].copy()

    def assert_frame(self, key, vox, opts, mc, n, px, argb):
        """The product's (px, argb) equal the strict reference build's, bit for bit.  Full comparison
        when the reference is live or the fixture holds the frame; digest + sample otherwise."""
        px = np.asarray(px, dtype=np.float32).reshape(-1)
        if self.live or os.path.exists(os.path.join(FIXED, key + ".npz")):
            want, want_argb = self.frame(key, vox, opts, mc, n)
            bad = int((px.view(np.uint32) != want.view(np.uint32)).reshape(-1, 4).any(axis=1).sum())
            assert bad == 0, f"{key}: {bad} pixels differ from the reference build ({self.source()})"
            if argb is not None:
                assert np.array_equal(argb, want_argb), key
            return
        d = _digests().get(key)
        if d is None:
            pytest.fail(f"no device-contract checker for `{key}`: oracle/_ref is absent and tests/golden/gfx950_strict/"
                        f"digests.json has no entry")
        assert d["inputs"] == input_digest(vox, opts, mc, n), f"digest of {key} was recorded for other inputs"
        sample = np.load(os.path.join(FIXED, "digest_samples.npz"))[key]
        got = px.view(np.uint32).reshape(-1, 4)[::SAMPLE_STRIDE].reshape(-1)
        where = np.nonzero(got != sample)[0]
        assert where.size == 0, f"{key}: sampled pixel {int(where[0]) // 4 * SAMPLE_STRIDE} differs from the reference build"
        assert sha(px) == d["pixels_sha"], f"{key}: accumulator digest differs from the reference build's"
        if argb is not None:
            assert sha(np.asarray(argb, dtype=np.uint32)) == d["argb_sha"], f"{key}: ARGB digest differs"


@pytest.fixture(scope="session")
def pin(oracle_mod):
    return Checker(oracle_mod)
