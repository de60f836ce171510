"""The parity chain pinned to the reference kernel itself, compiled for THIS chip with no
stand-in: the unmodified renderer.cl -> gfx950 code objects linked against ROCm's own OpenCL
built-in library (oracle/Makefile `ref_gfx950`; prebuilt in the build container, git-ignored,
travels with the snapshot like every other built file), launched through hipModuleLoad
(oracle/ref_gfx950_runner.cpp) exactly as core.clj:76-97 sequences the kernels.

What can and cannot be asserted: the reference source leaves rounding to the OpenCL compiler
(core.clj:128 builds with :fast-math :enable-mad), and every re-rounding flips hit/miss decisions
of a few pixels -- the contraction-off build (`strict`) disagrees with the other two on ~1-2 % of
the pixels per pass, while `default` and `fast` agree with each other within 1e-4 on ~all of them; the product's
DEFAULT contract is the `default` build bit for bit (last tests of this file: the fraction of pixels within 1e-4 of the
reference's own `fast` build for EVERY BASELINE configuration and every fixture scene, against the live code object or
its committed recording tests/golden/gfx950_fast/).  For the CPU-device contract the tests assert (a) bit-exactness of the HIP path against the CPU oracle in the
GPU cast mode too, (b) BASELINE's metric -- the fraction of pixels within 1e-4 relative -- of
the HIP path against each reference build, which must be as good as the agreement of the
reference builds among themselves, and ~100 % on the pixels where those builds agree.
Numbers: profiles/r06_pin_gfx950.txt (tools/pin_gfx950.py; rounds 3-5: r03_/r05_pin_gfx950.txt)."""
import numpy as np
import pytest

import scenes
from gfx950_pin import fast_ref, pin_default  # noqa: F401  (fixtures: the `fast` yardstick, the checker of the default contract)

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("cpu_contract")]  # this module checks against the CPU oracle


def rel(a, b):
    a = a.reshape(-1, 4)[:, :3].astype(np.float64)
    b = b.reshape(-1, 4)[:, :3].astype(np.float64)
    r = np.abs(a - b) / np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-6)
    return r.max(axis=1)


@pytest.fixture(scope="module")
def refs(oracle_mod):
    if not oracle_mod.have_gfx950_ref("fast"):
        pytest.skip("oracle/_ref/renderer_gfx950_*.hsaco not built (needs /root/reference: build container)")
    return oracle_mod


CASES = {
    "c1": dict(scenes.SCENES["c1_orange"], w=256, h=256),
    "c2_geometry": dict(vol="gyroid", vres=256, w=320, h=180, iter=4, mat="orange-stripes", theta=-45, dist=2.25,
                        dof=0.025),
    "metal_bounces": dict(scenes.SCENES["metal_3spp"], w=128, h=96),
}


@pytest.fixture(scope="module", params=list(CASES))
def rendered(request, refs, native):
    sc = scenes.build(CASES[request.param])
    n = sc["n"]
    ref = {b: refs.gfx950_render_frame(sc["vox"], sc["opts"], sc["mc"], n, build=b)[:2] for b in refs.GFX950_BUILDS}
    hip = {}
    with native.Context(0, contract="cpu") as ctx:  # (module-scoped fixture: the per-test cpu_contract fixture is not active here)
        ctx.set_volume(sc["vox"], sc["vres"])
        for mode in ("x86", "gpu"):
            ctx.set_seed_cast(mode)
            hip[mode] = ctx.render_frame(sc["opts"], sc["mc"], n)
    return request.param, sc, ref, hip


def test_reference_kernels_run_and_tonemap_agrees(rendered, refs):
    """RenderImage / TonemapImage of the code objects execute (OpenCL kernarg layout + hidden
    arguments filled from the metadata) and produce a real image; TonemapImage of the strict build
    equals the oracle's tonemap of the same accumulator."""
    name, sc, ref, hip = rendered
    for b, (px, argb) in ref.items():
        p = px.reshape(-1, 4)
        assert np.isfinite(p).all() and (p[:, 3] == 1.0).all(), b
        assert p[:, :3].std() > 0.05, b
        assert (argb >> 24 == 0xff).all(), b
    px, argb = ref["strict"]
    want = refs.tonemap_image(px.copy(), sc["opts"][:544])
    assert (want != argb).mean() < 1e-3  # (division / multiply order of the device build: a handful of 1-LSB channels)


def test_hip_equals_oracle_in_both_cast_modes(rendered, refs):
    name, sc, ref, hip = rendered
    for mode in ("x86", "gpu"):
        with refs.seed_cast(mode):
            want_px, want_argb = refs.render_frame(sc["vox"], sc["opts"], sc["mc"], sc["n"])
        got_px, got_argb = hip[mode]
        assert np.array_equal(got_px.view(np.uint32), want_px.view(np.uint32)), mode
        assert np.array_equal(got_argb, want_argb), mode
    # the modes do differ (about half of the AO seeds are negative)
    assert not np.array_equal(hip["x86"][0], hip["gpu"][0])


def test_metric_against_every_reference_build(rendered, refs):
    """BASELINE's parity metric of the HIP path (GPU cast mode = the casts of these code objects)
    against each reference build, next to the agreement of the reference builds among themselves."""
    name, sc, ref, hip = rendered
    px = {b: ref[b][0] for b in ref}
    stable = (rel(px["fast"], px["strict"]) <= 1e-4) & (rel(px["fast"], px["default"]) <= 1e-4) & \
             (rel(px["default"], px["strict"]) <= 1e-4)
    among = min((rel(px[a], px[b]) <= 1e-4).mean() for a in px for b in px if a < b)
    got = hip["gpu"][0]
    for b in px:
        r = rel(got, px[b])
        frac, frac_stable = (r <= 1e-4).mean(), (r[stable] <= 1e-4).mean()
        print(f"{name}: HIP gpu-cast vs `{b}`: {100 * frac:.3f} % within 1e-4 ({100 * frac_stable:.3f} % of the stable "
              f"pixels; reference builds among themselves: {100 * among:.3f} %)")
        # measured (profiles/archive_r03.txt FILE r03_pin_gfx950.txt, unchanged since): 96.6-98 % of all pixels, 97.4-99 % of the stable ones
        # -- the x86-strict arithmetic of this path (unfused mad, IEEE normalize, its own exp/pow)
        # against ocml's; the reference builds agree among themselves on 97.3-99.2 %
        assert frac >= 0.95, (b, frac, among)
        assert frac_stable >= 0.96, (b, frac_stable)


# ---- north star: "pixels within 1e-4 of the OpenCL reference" = of the reference built with ITS OWN options ----------------
#
# `fast` = renderer.cl compiled with -cl-fast-relaxed-math -cl-mad-enable (core.clj:128).  The product's default contract
# is the `default` build of the same source bit for bit (tests/test_gpu_device_contract.py); what is asserted here is how
# far that build is from `fast`, per configuration.  A pixel leaves the 1e-4 band as soon as ONE hit/miss, material or
# light decision of ONE of its passes flips under fast-math's re-association: the share grows with the passes
# blended into a pixel and with the bounces per sample (:metal = 3), it is not a rounding that grows.
#
# FLOOR[key] = asserted lower bound on the fraction of pixels within 1e-4; measured values and the residual per
# configuration: profiles/r06_pin_gfx950.txt.  Checked against the live code object where oracle/_ref travelled (every
# pixel), else against the committed recording (tests/golden/gfx950_fast/: every pixel of the fixtures and of config 1,
# every 997th pixel of configs 2-5) -- a clean clone reports the fraction on the sample instead of skipping.
FLOOR = {"c1": 0.999, "c2": 0.999, "c3": 0.999, "c4": 0.999, "c5": 0.999}
FLOOR_SCENES = 0.995  # fixture scenes: 768-3072 pixels each, one pixel = 0.03-0.13 %


def _default_contract_frame(native, vox, vres, opts, mc, n):
    with native.Context(0, contract="gfx950-default") as ctx:  # (named: this module's fixture switches unnamed contexts to "cpu")
        ctx.set_volume(vox, vres)
        px, _ = ctx.render_frame(opts, mc, n, want_argb=False)
    return px


def _report(key, r, stride, src):
    import gfx950_pin as gp

    frac = float((r <= 1e-4).mean())
    print(f"{key}: default contract vs `fast` ({src}; {'every pixel' if stride == 1 else f'every {gp.SAMPLE_STRIDE}th pixel'}, "
          f"{r.size} compared): {100 * frac:.4f} % within 1e-4, {100 * float((r == 0).mean()):.2f} % bit-equal, max rel {r.max():.2e}, "
          f"{int((r > 1e-4).sum())} beyond")
    return frac


@pytest.mark.parametrize("config", ["c1", "c2", "c3", "c4", "c5"])
def test_every_config_full_size_against_the_references_own_build(config, native, fast_ref, pin_default, oracle_mod):
    """ALL FIVE BASELINE configurations at FULL size, the product in its DEFAULT contract against `fast`.  Also, where
    the live builds are present: default contract == `default` build (every float), and at least as close to `fast` as
    the closest other build of the reference is."""
    import bench
    import gfx950_pin as gp

    wl = bench.WORKLOADS[config]
    vox, vres, opts, mc = bench.build_inputs(wl)
    n = wl["w"] * wl["h"]
    got = _default_contract_frame(native, vox, vres, opts, mc, n)
    want, stride = fast_ref.pixels(config, vox, opts, mc, n)
    r = gp.rel_err(got.reshape(-1, 4)[::stride], want)
    frac = _report(config, r, stride, fast_ref.source())
    assert frac >= FLOOR[config], (config, frac)
    undefined = gp.undefined_work_items(oracle_mod, vox, opts, mc, n) if config == "c1" else None  # (one work-item, oracle/pin.py)
    pin_default.assert_frame(config, vox, opts, mc, n, got, None, undefined=undefined)  # ... and it IS the `default` build, bit for bit
    if fast_ref.live and config in ("c1", "c2"):  # (the strict build of the big frames: seconds of GPU time each, shown in the profile)
        strict = oracle_mod.gfx950_render_frame(vox, opts, mc, n, build="strict", tonemap=False)[0]
        frac_strict = float((gp.rel_err(strict, want) <= 1e-4).mean())
        print(f"{config}: the contraction-off build `strict` vs `fast`: {100 * frac_strict:.3f} %")
        assert frac >= frac_strict - 1e-9
    assert len(np.unique(got.reshape(-1, 4)[::97, :3])) > 500  # a real frame


@pytest.mark.parametrize("name", list(scenes.SCENES))
def test_every_fixture_against_the_references_own_build(name, native, fast_ref):
    import gfx950_pin as gp

    sc = scenes.build(name)
    got = _default_contract_frame(native, sc["vox"], sc["vres"], sc["opts"], sc["mc"], sc["n"])
    want, stride = fast_ref.pixels(name, sc["vox"], sc["opts"], sc["mc"], sc["n"])
    assert stride == 1
    frac = _report(name, gp.rel_err(got, want), 1, fast_ref.source())
    assert frac >= FLOOR_SCENES, (name, frac)
