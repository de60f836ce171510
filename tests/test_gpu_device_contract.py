"""RM_CONTRACT_GFX950_DEFAULT (the library default) and RM_CONTRACT_GFX950_STRICT: the kernels reproduce, bit for
bit, the pixels of the reference kernel itself as ROCm's OpenCL compiler builds it for this chip -- with no options
(`default`: clang's OpenCL defaults contract a*b+c inside an expression and lower `/` to 2.5 ulp; this build agrees with
the reference's OWN -cl-fast-relaxed-math build within 1e-4 on ~all pixels, tests/test_gpu_pin_gfx950.py) and with
-ffp-contract=off -cl-fp32-correctly-rounded-divide-sqrt (`strict`).

The checkers are oracle/_ref/renderer_gfx950_{default,strict}.hsaco -- the UNMODIFIED renderer.cl compiled
where it lies (oracle/Makefile ref_gfx950) and linked by the clang driver against ROCm's own OpenCL built-in
library; no stand-in for anything.  They run on the GPU through oracle/ref_gfx950_runner.cpp exactly as the
reference host sequences its kernels (core.clj:76-97).  In these contracts the product's built-ins ARE that
library's functions (csrc/rm_math.hpp), so every float32 of the accumulator and every ARGB word must be
equal -- whole frames, at every BASELINE configuration's full size.  Every test below runs once per contract.

Where a code object is absent (a clean clone: oracle/_ref is git-ignored) the same frames are
checked against its recorded outputs, tests/golden/gfx950_<build>/ (oracle/pin.py): full
accumulators for the fixture scenes and config 1, digests + sampled pixels for the large frames.
With neither, the tests fail -- they never skip on a GPU box."""
import os
import sys

import numpy as np
import pytest

import scenes
from gfx950_pin import pin, pin_default, pin_each  # noqa: F401  (fixtures: live reference build, else the committed recordings)

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def refs(oracle_mod):
    """The LIVE reference builds, for the tests that render inputs no recording exists for (random frames)."""
    if not (oracle_mod.have_gfx950_ref("strict") and oracle_mod.have_gfx950_ref("default")):
        pytest.skip("oracle/_ref/renderer_gfx950_{strict,default}.hsaco not built (needs /root/reference: build container); "
                    "the recorded frames of tests/golden/gfx950_*/ are checked by the other tests of this file")
    return oracle_mod


def _differing(a, b):
    return int((np.asarray(a).view(np.uint32) != np.asarray(b).view(np.uint32)).reshape(-1, 4).any(axis=1).sum())


@pytest.mark.parametrize("name", list(scenes.SCENES))
def test_fixture_scenes_every_kernel(native, pin_each, monkeypatch, name):
    """Frame kernel (accelerated), single-pass kernels, the frame tiled over 3 ranks inside the
    library, and the plain table-free kernels: all equal to the reference build."""
    pin, contract = pin_each
    sc = scenes.build(name)
    n = sc["n"]
    want, want_argb = pin.frame(name, sc["vox"], sc["opts"], sc["mc"], n)
    with native.Context(0) as ctx:
        ctx.set_contract(contract)
        ctx.set_volume(sc["vox"], sc["vres"])
        px, argb = ctx.render_frame(sc["opts"], sc["mc"], n)
        assert _differing(px, want) == 0 and np.array_equal(argb, want_argb)
        acc = np.zeros(4 * n, np.float32)
        for i in range(sc["iter"]):
            ctx.render_image(np.ascontiguousarray(sc["mc"][i]), sc["opts"][i * 544:(i + 1) * 544], acc, n=n)
        assert _differing(acc, want) == 0
        assert np.array_equal(ctx.tonemap_image(acc, sc["opts"][:544], n=n), want_argb)
    with native.Context([0, 0, 0]) as ctx:
        ctx.set_contract(contract)
        ctx.set_volume(sc["vox"], sc["vres"])
        px, argb = ctx.render_frame(sc["opts"], sc["mc"], n)
        assert _differing(px, want) == 0 and np.array_equal(argb, want_argb)
    monkeypatch.setenv("RAYMARCH_NO_ACCEL", "1")
    with native.Context(0) as ctx:
        ctx.set_contract(contract)
        ctx.set_volume(sc["vox"], sc["vres"])
        px, argb = ctx.render_frame(sc["opts"], sc["mc"], n)
        assert _differing(px, want) == 0 and np.array_equal(argb, want_argb)


@pytest.mark.parametrize("passes,pack", [(8, "3"), (16, "4"), (12, "4"), (25, "4")])
def test_pass_packed_wavefronts(native, pin_each, monkeypatch, passes, pack):
    pin, contract = pin_each
    spec = dict(vol="gyroid", vres=64, w=56, h=40, iter=passes, mat="metal", theta=-30, dist=2.2, dof=0.02)
    sc = scenes.build(spec, mc_seed=500)
    monkeypatch.setenv("RAYMARCH_PASS_PACK", pack)
    with native.Context(0) as ctx:
        ctx.set_contract(contract)
        ctx.set_volume(sc["vox"], sc["vres"])
        px, argb = ctx.render_frame(sc["opts"], sc["mc"], sc["n"])
    pin.assert_frame(f"pass_packed_{passes}", sc["vox"], sc["opts"], sc["mc"], sc["n"], px, argb)


@pytest.mark.parametrize("config", ["c1", "c2", "c3", "c4", "c5"])
def test_baseline_configurations_whole_frames(native, pin_each, config, oracle_mod):
    """Every BASELINE configuration at its full size, the WHOLE frame: the reference kernel renders
    it on this GPU (C2: 16 launches, ~0.25 s; C4: 64 launches over 8.3 M pixels; or its recorded
    output is read: all of config 1, digest + every 997th pixel of configs 2-5), the product renders
    it in one launch; all floats and all ARGB words equal."""
    import bench

    pin, contract = pin_each
    wl = bench.WORKLOADS[config]
    vox, vres, opts, mc = bench.build_inputs(wl)
    n = wl["w"] * wl["h"]
    with native.Context(0) as ctx:
        ctx.set_contract(contract)
        ctx.set_volume(vox, vres)
        px, argb = ctx.render_frame(opts, mc, n)
        ms, launches = ctx.last_frame_timing()
    print(f"{config} ({contract}): {n} pixels x {wl['spp']} passes -- this path {ms:.2f} ms, checker: {pin.source()}")
    # (config 1 holds ONE work-item the reference leaves undefined: a live build returns scratch residue there)
    import gfx950_pin as gp

    undefined = gp.undefined_work_items(oracle_mod, vox, opts, mc, n) if config == "c1" else None
    assert undefined is None or int(undefined.sum()) == 1
    pin.assert_frame(config, vox, opts, mc, n, px, argb, undefined=undefined)
    assert len(np.unique(px.reshape(-1, 4)[::97, :3])) > 1000  # a real image


def test_contract_is_per_context_and_switchable(native, pin, pin_default, oracle_mod):
    """The same context renders all three contracts; each equals its own checker; a context that never names a
    contract renders the `default` build's pixels (ABI 4)."""
    sc = scenes.build("orange_dof_2spp")
    n = sc["n"]
    want_strict, _ = pin.frame("orange_dof_2spp", sc["vox"], sc["opts"], sc["mc"], n)
    want_def, _ = pin_default.frame("orange_dof_2spp", sc["vox"], sc["opts"], sc["mc"], n)
    want_cpu, _ = oracle_mod.render_frame(sc["vox"], sc["opts"], sc["mc"], n)
    assert native.DEFAULT_CONTRACT is None and native.lib().rm_abi_version() >= 4
    with native.Context(0) as ctx:
        ctx.set_volume(sc["vox"], sc["vres"])
        assert _differing(ctx.render_frame(sc["opts"], sc["mc"], n)[0], want_def) == 0
        for _ in range(2):
            ctx.set_contract("gfx950-strict")
            assert _differing(ctx.render_frame(sc["opts"], sc["mc"], n)[0], want_strict) == 0
            ctx.set_contract("cpu")
            assert _differing(ctx.render_frame(sc["opts"], sc["mc"], n)[0], want_cpu) == 0
            ctx.set_contract("gfx950-default")
            assert _differing(ctx.render_frame(sc["opts"], sc["mc"], n)[0], want_def) == 0
    assert _differing(want_strict, want_cpu) > 0 and _differing(want_strict, want_def) > 0


@pytest.mark.parametrize("contract", ["gfx950-default", "gfx950-strict"])
def test_randomised_frames_against_the_reference_build(refs, contract):
    """tools/fuzz_parity.py in a device contract: random volumes, cameras (inside and around),
    presets, record overrides, pass counts -- the reference kernel on the GPU is the checker."""
    import subprocess

    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "--contract", contract,
                        "--cases", "60", "--seed", "3"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_work_item_undefined_under_the_device_arithmetic_only(native, refs, oracle_mod):
    """Round-3 fuzz case 1369 of seed 3303, rebuilt from its parameters: ONE work-item of pass 7 differs from
    the reference build.  Its third bounce lands, under THIS chip's arithmetic only, on a material index
    outside the record: the reference kernel then reads its private copy of the record out of bounds
    (renderer.cl:394,418 -- undefined), the product defines such a material as zero.  The CPU restatement
    cannot flag it (under its arithmetic the bounce lands elsewhere); the plain algorithm run in the device
    contract counts the lookup.  Everything else of the frame is bit-identical."""
    import raymarchcl_amd as rm
    from raymarchcl_amd import generators as gen, structs

    base = dict(width=51, height=31, vres=[64, 64, 64], iter=8,
                eyepos=[1.5138196778079402, 0.9110687635350511, 1.4679164504197397],
                targetpos=[-0.20502942760479748, -0.44330606932885597, -0.20290005279972878], mat="metal",
                fov=101.85582954649558, dof=0.001)
    over = dict(fogPow=0.0842790072010613, aoAmp=0.37821923398069385, aoStepDist=0.27685733566557547)
    it, n = 8, 1580
    recs = []
    for i in range(it):
        o = rm.render_options(t=i * 0.333, **base)
        o.update(over)
        recs.append(structs.encode_bytes(o))
    opts = b"".join(recs)
    mc = np.stack([gen.generate_scatter_offsets(0x4000, seed=612096387 + i) for i in range(it)])
    vox = scenes.volume("terrain", 64)
    want, _, _ = refs.gfx950_render_frame(vox, opts, mc, n, build="strict", tonemap=False)
    mask = np.zeros(n, np.uint8)
    acc = np.zeros(4 * n, np.float32)
    for i in range(it):
        oracle_mod.render_image(vox, mc[i], opts[i * 544:(i + 1) * 544], acc, n=n, undefined_mask=mask)
    assert not mask.any()  # defined everywhere under the CPU device's arithmetic
    with native.Context(0) as ctx:
        ctx.set_contract("gfx950-strict")
        ctx.set_volume(vox, [64, 64, 64])
        px, _ = ctx.render_frame(opts, mc, n, want_argb=False)
        items = np.nonzero((px.view(np.uint32) != want.view(np.uint32)).reshape(-1, 4).any(axis=1))[0]
        assert list(items) == [597]
        oob = []
        scratch = np.zeros(4 * n, np.float32)
        for k in range(it):  # lookups outside the record by work-item 597 alone: count(0..597) - count(0..596)
            c1, c0 = native.Counters(), native.Counters()
            ctx.render_image(mc[k], opts[k * 544:(k + 1) * 544], scratch, 598, counters=c1)
            ctx.render_image(mc[k], opts[k * 544:(k + 1) * 544], scratch, 597, counters=c0)
            oob.append(c1.oob_material - c0.oob_material)
        assert oob[7] > 0 and sum(oob[:7]) == 0, oob


def test_recorded_frames_equal_the_live_reference_build(pin_each, oracle_mod):
    """Where both checkers exist they must agree: the recordings of tests/golden/gfx950_<build>/ ARE
    the outputs of oracle/_ref/renderer_gfx950_<build>.hsaco on this chip (every fixture scene in full,
    the pass-packed frames and config 2 by digest)."""
    pin, _contract = pin_each
    if not pin.live:
        pytest.skip("no live reference build on this box: the recordings are the checker")
    import gfx950_pin

    fixed = gfx950_pin.Checker(oracle_mod, pin.build)
    fixed.live = False
    for name in scenes.SCENES:
        sc = scenes.build(name)
        px, argb = pin.frame(name, sc["vox"], sc["opts"], sc["mc"], sc["n"])
        want, want_argb = fixed.frame(name, sc["vox"], sc["opts"], sc["mc"], sc["n"])
        assert _differing(px, want) == 0 and np.array_equal(argb, want_argb), name
    import bench

    wl = bench.WORKLOADS["c2"]
    vox, vres, opts, mc = bench.build_inputs(wl)
    n = wl["w"] * wl["h"]
    px, argb, _ = oracle_mod.gfx950_render_frame(vox, opts, mc, n, build=pin.build)
    fixed.assert_frame("c2", vox, opts, mc, n, px, argb)


def test_a_gpu_box_without_any_checker_fails(oracle_mod, tmp_path):
    """Neither the code object nor a recording: the check must FAIL, not skip."""
    import gfx950_pin

    for build in gfx950_pin.BUILDS:
        c = gfx950_pin.Checker(oracle_mod, build, fixed=str(tmp_path))
        c.live = False
        sc = scenes.build("solid_volume")
        with pytest.raises(gfx950_pin.CheckerMissing):
            c.frame("solid_volume", sc["vox"], sc["opts"], sc["mc"], sc["n"])
        with pytest.raises(gfx950_pin.CheckerMissing):
            c.assert_frame("c2", sc["vox"], sc["opts"], sc["mc"], sc["n"], np.zeros(4 * sc["n"], np.float32), None)
