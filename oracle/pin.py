"""oracle.pin -- TEST INFRASTRUCTURE: the checkers of the two device contracts.

A device contract of the product is pinned to a BUILD of the unmodified reference kernel for gfx950:

  build "strict"   RM_CONTRACT_GFX950_STRICT   renderer.cl compiled with -ffp-contract=off
                                                -cl-fp32-correctly-rounded-divide-sqrt
  build "default"  RM_CONTRACT_GFX950_DEFAULT  renderer.cl compiled with NO options (clang's OpenCL defaults:
                                                contraction inside expressions, 2.5-ulp divide); agrees with the
                                                reference's own -cl-fast-relaxed-math build within 1e-4 on ~all pixels

Two sources per build, in this order:
  1. LIVE   oracle/_ref/renderer_gfx950_<build>.hsaco -- compiled by oracle/Makefile in the build container
            (git-ignored, travels with the snapshot), run on the GPU through oracle/ref_gfx950_runner.cpp;
  2. FIXED  tests/golden/gfx950_<build>/ -- the outputs of exactly that code object on the same inputs, recorded
            ON the GPU by tests/golden/make_golden_gfx950.py and committed (data only): full float32 accumulators
            + ARGB words for the fixture scenes and config 1, sha256 digests (digests.json) + a sparse sample of
            pixels (digest_samples.npz) for the large frames (configs 2-5, pass-packed frames).
A third build is recorded the same way but is NOT a bit-exact checker:

  build "fast"     (no contract)               renderer.cl compiled with the reference's OWN options, -cl-fast-relaxed-math
                                                -cl-mad-enable (core.clj:128).  Fast-math lets the compiler re-associate, so no
                                                hand-written kernel can promise its bits; it is the YARDSTICK of BASELINE's
                                                parity metric ("pixels within 1e-4 of the OpenCL reference"): FastReference
                                                below serves its pixels -- live, or from tests/golden/gfx950_fast/ (full
                                                frames for the fixtures and config 1, every 997th pixel of configs 2-5).

A clean clone on a GPU box therefore still checks every device-contract frame bit for bit; when neither source
exists the check raises CheckerMissing (an AssertionError: tests FAIL, they do not skip -- a GPU box without a
checker is a broken checkout, not a reason to pass).

No pytest here: __graft_entry__.smoke() uses this module too.
"""
import hashlib
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAMPLE_STRIDE = 997  # pixels kept from a digest-pinned frame (prime: walks through rows and columns)
BUILDS = ("strict", "default")  # the bit-exact checkers, one per device contract
METRIC_BUILD = "fast"           # the reference's own build options: yardstick of the 1e-4 metric (FastReference)
RECORDED = BUILDS + (METRIC_BUILD,)  # builds whose outputs are committed under tests/golden/gfx950_<build>/
CONTRACT_OF = {"strict": "gfx950-strict", "default": "gfx950-default"}  # raymarchcl_amd._native.CONTRACTS names


class CheckerMissing(AssertionError):
    pass


def fixed_dir(build):
    return os.path.join(ROOT, "tests", "golden", "gfx950_" + build)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def input_digest(vox, opts, mc, n):
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(vox).tobytes())
    h.update(bytes(opts))
    h.update(np.ascontiguousarray(mc, dtype=np.float32).tobytes())
    h.update(str(int(n)).encode())
    return h.hexdigest()


class Checker:
    def __init__(self, oracle_mod, build="strict", fixed=None):
        assert build in BUILDS, build
        self.oracle = oracle_mod
        self.build = build
        self.fixed = fixed or fixed_dir(build)
        self.live = bool(oracle_mod.have_gfx950_ref(build)) and os.environ.get("RAYMARCH_PIN_FIXED_ONLY", "0") != "1"

    def source(self):
        return f"live `{self.build}` reference build" if self.live else f"committed recordings of the `{self.build}` build"

    def _digests(self):
        p = os.path.join(self.fixed, "digests.json")
        return json.load(open(p)) if os.path.exists(p) else {}

    def frame(self, key, vox, opts, mc, n):
        """-> (pixels float32[4n], argb uint32[n]) of the reference build for these inputs;
        only for keys whose full output is on file (fixture scenes, c1)."""
        if self.live:
            px, argb, _ = self.oracle.gfx950_render_frame(vox, opts, mc, n, build=self.build)
            return px, argb
        path = os.path.join(self.fixed, key + ".npz")
        if not os.path.exists(path):
            raise CheckerMissing(f"no device-contract checker for `{key}`: neither oracle/_ref/renderer_gfx950_{self.build}.hsaco "
                                 f"nor {os.path.relpath(path, ROOT)} exists")
        z = np.load(path)
        assert str(z["inputs"]) == input_digest(vox, opts, mc, n), f"fixture {key} was recorded for other inputs"
        return z["pixels"].copy(), z["argb"].copy()

    def assert_frame(self, key, vox, opts, mc, n, px, argb, undefined=None):
        """The product's (px, argb) equal the reference build's, bit for bit.  Full comparison
        when the reference is live or the fixture holds the frame; digest + sample otherwise.
        `undefined`: mask of the work-items whose value the REFERENCE leaves undefined (undefined_work_items below) --
        a live build returns whatever its scratch memory held there, so they are left out of the full comparison."""
        px = np.asarray(px, dtype=np.float32).reshape(-1)
        if self.live or os.path.exists(os.path.join(self.fixed, key + ".npz")):
            want, want_argb = self.frame(key, vox, opts, mc, n)
            keep = np.ones(n, bool) if undefined is None else ~np.asarray(undefined, bool)
            differs = (px.view(np.uint32) != want.view(np.uint32)).reshape(-1, 4).any(axis=1)
            bad = int((differs & keep).sum())
            assert bad == 0, f"{key}: {bad} pixels differ from the reference build ({self.source()})"
            if argb is not None:
                assert np.array_equal(np.asarray(argb)[keep], want_argb[keep]), key
            return
        d = self._digests().get(key)
        if d is None:
            raise CheckerMissing(f"no device-contract checker for `{key}`: oracle/_ref is absent and "
                                 f"{os.path.relpath(self.fixed, ROOT)}/digests.json has no entry")
        assert d["inputs"] == input_digest(vox, opts, mc, n), f"digest of {key} was recorded for other inputs"
        sample = np.load(os.path.join(self.fixed, "digest_samples.npz"))[key]
        got = px.view(np.uint32).reshape(-1, 4)[::SAMPLE_STRIDE].reshape(-1)
        where = np.nonzero(got != sample)[0]
        assert where.size == 0, f"{key}: sampled pixel {int(where[0]) // 4 * SAMPLE_STRIDE} differs from the reference build"
        assert sha(px) == d["pixels_sha"], f"{key}: accumulator digest differs from the reference build's"
        if argb is not None:
            assert sha(np.asarray(argb, dtype=np.uint32)) == d["argb_sha"], f"{key}: ARGB digest differs"


def undefined_work_items(oracle, vox, opts, mc, n):
    """-> bool[n]: the work-items whose material index leaves the record in some pass.  The reference reads its PRIVATE
    copy of the record out of bounds there (renderer.cl:394,418): undefined behaviour -- a build of it returns what its
    scratch memory happened to hold (zeros on a fresh device, which is the value the product defines; anything, NaN
    included, after other kernels have run: seen once in round 6 on pixel 31669 of config 1, its only such work-item).
    Found by the CPU restatement, which marks them; seconds for config 1, too slow for the big configurations (none of
    which has one: profiles/r06_pin_gfx950.txt for config 2; configs 3-5 by their digests holding across boxes)."""
    mask = np.zeros(n, np.uint8)
    scratch = np.zeros(4 * n, np.float32)
    for i in range(len(opts) // 544):
        oracle.render_image(vox, mc[i], opts[i * 544:(i + 1) * 544], scratch, n=n, undefined_mask=mask)
    return mask != 0


def rel_err(a, b):
    """BASELINE's parity metric per pixel: max over r, g, b of |a - b| / max(|a|, |b|, 1e-6); a, b float32 [k, 4] or [4k]."""
    a = np.asarray(a).reshape(-1, 4)[:, :3].astype(np.float64)
    b = np.asarray(b).reshape(-1, 4)[:, :3].astype(np.float64)
    with np.errstate(invalid="ignore"):
        r = np.abs(a - b) / np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-6)
    r[np.isnan(a) | np.isnan(b)] = np.inf
    r[np.isnan(a) & np.isnan(b)] = 0.0
    return r.max(axis=1)


class FastReference:
    """Pixels of the reference built with its own options (`fast`) for BASELINE's 1e-4 metric: the live code object where
    oracle/_ref travelled, else the committed recording.  pixels() -> (float32 [k, 4], stride): stride 1 = every pixel of
    the frame, SAMPLE_STRIDE = every 997th (configs 2-5 from the recordings)."""

    def __init__(self, oracle_mod, fixed=None):
        self.oracle = oracle_mod
        self.fixed = fixed or fixed_dir(METRIC_BUILD)
        self.live = bool(oracle_mod.have_gfx950_ref(METRIC_BUILD)) and os.environ.get("RAYMARCH_PIN_FIXED_ONLY", "0") != "1"

    def source(self):
        return "live `fast` reference build" if self.live else "committed recordings of the `fast` build"

    def pixels(self, key, vox, opts, mc, n):
        if self.live:
            px, _, _ = self.oracle.gfx950_render_frame(vox, opts, mc, n, build=METRIC_BUILD, tonemap=False)
            return px.reshape(-1, 4), 1
        path = os.path.join(self.fixed, key + ".npz")
        if os.path.exists(path):
            z = np.load(path)
            assert str(z["inputs"]) == input_digest(vox, opts, mc, n), f"fixture {key} was recorded for other inputs"
            return z["pixels"].reshape(-1, 4).copy(), 1
        dj = os.path.join(self.fixed, "digests.json")
        d = (json.load(open(dj)) if os.path.exists(dj) else {}).get(key)
        if d is None:
            raise CheckerMissing(f"no `fast` reference for `{key}`: neither oracle/_ref/renderer_gfx950_fast.hsaco nor a "
                                 f"recording under {os.path.relpath(self.fixed, ROOT)} exists")
        assert d["inputs"] == input_digest(vox, opts, mc, n), f"sample of {key} was recorded for other inputs"
        s = np.load(os.path.join(self.fixed, "digest_samples.npz"))[key]
        return s.view(np.float32).reshape(-1, 4).copy(), SAMPLE_STRIDE
