"""Parity of the HIP path (through the C ABI) with the oracle: the golden
fixtures generated from the reference kernel, the CPU restatement on the same
seeded inputs, and size-independent properties at BASELINE's full sizes.
Bit-exact: the path is float32 with a pinned IEEE op sequence."""
import numpy as np
import pytest

import scenes
from conftest import load_golden

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("cpu_contract")]  # this module checks against the CPU oracle

NAMES = list(scenes.SCENES)


def _eq(a, b):
    return np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32))


@pytest.mark.parametrize("name", NAMES)
def test_frame_matches_reference_fixture(gpu_ctx, name):
    g = load_golden(name)
    sc = scenes.build(name)
    gpu_ctx.set_volume(g["vox"], tuple(int(v) for v in g["vres"]))
    px, argb = gpu_ctx.render_frame(g["opts"].tobytes(), sc["mc"], int(g["n"]))
    assert _eq(px, g["pixels"]), f"{(px.view(np.uint32) != g['pixels'].view(np.uint32)).sum()} floats differ"
    assert np.array_equal(argb, g["argb"])


@pytest.mark.parametrize("name", ["c1_orange", "metal_3spp", "ragged_50x37"])
def test_single_pass_entry_points_match_oracle(gpu_ctx, oracle_mod, native, name):
    """rm_render_image / _range / _counted + rm_tonemap_image == RenderImage / TonemapImage."""
    sc = scenes.build(name)
    n = sc["n"]
    gpu_ctx.set_volume(sc["vox"], sc["vres"])
    rng = np.random.default_rng(1)
    start = rng.uniform(0, 4, 4 * n).astype(np.float32)  # a non-zero accumulator to blend into
    mc = sc["mc"][0].copy()
    o = sc["opts"][:544]
    st = oracle_mod.Stats()
    want = oracle_mod.render_image(sc["vox"], mc, o, start.copy(), n=n, stats=st)
    cnt = native.Counters()
    got = gpu_ctx.render_image(mc, o, start.copy(), n=n, counters=cnt)
    assert _eq(got, want)
    assert cnt.as_dict() == st.as_dict()  # same algorithmic work, event for event
    # sub-ranges (what a tile of the NDRange does)
    a, b = n // 3, n // 3 + 777
    want_r = oracle_mod.render_image(sc["vox"], mc, o, start.copy(), n=n, id0=a, id1=b)
    got_r = gpu_ctx.render_image(mc, o, start.copy(), n=n, id0=a, id1=b)
    assert _eq(got_r, want_r)
    assert np.array_equal(gpu_ctx.tonemap_image(want, o, n=n), oracle_mod.tonemap_image(want, o, n=n))


def test_tonemap_edge_values(gpu_ctx, oracle_mod):
    px = np.zeros((8, 4), np.float32)
    px[:, 0] = [0, 1e-9, 0.5, 1.5, 1e9, np.inf, -0.2, np.nan]
    px[:, 1] = [3.0, 100.0, -1.5, -3.0, 1e-3, 7.0, 0.25, 1.0]
    px[:, 2] = px[::-1, 0]
    o = scenes.build("c1_orange")["opts"][:544]
    assert np.array_equal(gpu_ctx.tonemap_image(px.reshape(-1).copy(), o),
                          oracle_mod.tonemap_image(px.reshape(-1).copy(), o))


def test_config1_full_frame(gpu_ctx, oracle_mod):
    """BASELINE config 1 at full size vs the reference's hashes."""
    g = load_golden("c1_full")
    sc = scenes.build(dict(scenes.SCENES["c1_orange"], w=256, h=256))
    gpu_ctx.set_volume(sc["vox"], sc["vres"])
    px, argb = gpu_ctx.render_frame(sc["opts"], sc["mc"], sc["n"])
    want, want_argb = oracle_mod.render_frame(sc["vox"], sc["opts"], sc["mc"], sc["n"])
    assert _eq(px, want) and np.array_equal(argb, want_argb)  # incl. the undefined work-item
    und = g["undefined_ids"]
    px.reshape(-1, 4)[und] = 0
    argb[und] = 0
    assert scenes.sha(px) == str(g["pixels_sha"]) and scenes.sha(argb) == str(g["argb_sha"])


def test_error_paths(gpu_ctx, native):
    sc = scenes.build("c1_orange")
    ctx = native.Context(0)
    px = np.zeros(4 * sc["n"], np.float32)
    with pytest.raises(native.RmError):  # no volume yet
        ctx.render_image(sc["mc"][0].copy(), sc["opts"][:544], px)
    ctx.set_volume(sc["vox"], sc["vres"])
    bad = bytearray(sc["opts"][:544])
    bad[160:164] = np.int32(32).tobytes()  # voxelRes.x disagrees with the volume
    with pytest.raises(native.RmError):
        ctx.render_image(sc["mc"][0].copy(), bytes(bad), px)
    with pytest.raises(ValueError):
        ctx.set_volume(sc["vox"][:100], sc["vres"])
    ctx.close()


# ---- BASELINE config 2 size: 256^3 gyroid, 1280x720 (properties + sampled oracle) ----

@pytest.fixture(scope="module")
def config2():
    spec = dict(vol="gyroid", vres=256, w=1280, h=720, iter=2, mat="orange-stripes", theta=-45,
                dist=2.25, dof=0.025)
    return scenes.build(spec)


def test_config2_sampled_against_oracle_and_partition_invariance(gpu_ctx, oracle_mod, config2):
    import torch

    sc = config2
    n, it = sc["n"], sc["iter"]
    gpu_ctx.set_volume(sc["vox"], sc["vres"])
    px, argb = gpu_ctx.render_frame(sc["opts"], sc["mc"], n)
    assert not np.isnan(px).any() and (px.reshape(-1, 4)[:, 3] == 1.0).all()
    # determinism: a second run is bit-identical
    px2, _ = gpu_ctx.render_frame(sc["opts"], sc["mc"], n, want_argb=False)
    assert _eq(px, px2)
    # oracle on 3000 random work-items + 2 full rows, all passes in order
    rng = np.random.default_rng(0)
    ids = np.unique(np.concatenate([rng.integers(0, n, 3000), np.arange(1280 * 300, 1280 * 301),
                                    np.arange(1280 * 500, 1280 * 501), [0, n - 1]]))
    want = np.zeros(4 * n, np.float32)
    for i in range(it):
        for lo, hi in _runs(ids):
            oracle_mod.render_image(sc["vox"], sc["mc"][i].copy(), sc["opts"][i * 544:(i + 1) * 544],
                                    want, n=n, id0=lo, id1=hi, threads=1)
    assert _eq(px.reshape(-1, 4)[ids], want.reshape(-1, 4)[ids])
    assert np.array_equal(argb[ids], oracle_mod.tonemap_image(want, sc["opts"][:544], n=n)[ids])
    # tile partition (the multi-GPU split): 3 interleaved partitions, "gathered" by
    # concatenation, resolved on the device == the full frame
    from raymarchcl_amd import multigpu

    dev = torch.device("cuda:0")
    d_opts = torch.frombuffer(bytearray(sc["opts"]), dtype=torch.uint8).to(dev)
    d_mc = torch.from_numpy(sc["mc"]).to(dev)
    parts = 3
    tpp = multigpu.tiles_per_part(sc["w"], n, parts)
    d_all = torch.zeros(parts * tpp * 64 * 4, dtype=torch.float32, device=dev)
    gpu_ctx.check_device_opts(d_opts.data_ptr(), it, n, sc["w"])
    for r in range(parts):
        part = d_all[r * tpp * 256:(r + 1) * tpp * 256]
        gpu_ctx.frame_device(d_opts.data_ptr(), d_mc.data_ptr(), it, n, sc["w"], part.data_ptr(), r, parts)
    d_px = torch.empty(4 * n, dtype=torch.float32, device=dev)
    d_argb = torch.empty(n, dtype=torch.int32, device=dev)
    gpu_ctx.resolve_device(d_all.data_ptr(), parts, d_opts.data_ptr(), n, sc["w"], d_px.data_ptr(),
                           d_argb.data_ptr())
    gpu_ctx.synchronize()
    assert _eq(d_px.cpu().numpy(), px)
    assert np.array_equal(d_argb.cpu().numpy().view(np.uint32), argb)
    # the host mirror of the un-permute agrees with the kernel
    idx = multigpu.gathered_index_map(sc["w"], n, parts)
    assert _eq(d_all.cpu().numpy().reshape(-1, 4)[idx].reshape(-1), px)


def test_frames_in_flight(gpu_ctx, config2):
    """Successive frames on alternating streams: every slot's frame == the serial frame,
    also after a slot has been reused."""
    import torch

    from raymarchcl_amd import multigpu

    sc = config2
    gpu_ctx.set_volume(sc["vox"], sc["vres"])
    px, argb = gpu_ctx.render_frame(sc["opts"], sc["mc"], sc["n"])
    fr = multigpu.FrameRenderer(sc["vox"], sc["vres"], sc["opts"], sc["mc"], sc["n"], sc["w"],
                                frames_in_flight=3)
    assert len(fr.slots) == 3 and len({s.stream.cuda_stream for s in fr.slots}) == 3
    assert torch.cuda.current_stream().cuda_stream not in {s.stream.cuda_stream for s in fr.slots}
    outs = [fr.render() for _ in range(7)]
    torch.cuda.synchronize()
    assert len({o[0].data_ptr() for o in outs}) == 3
    for d_px, d_argb in outs[-3:]:
        assert _eq(d_px.cpu().numpy(), px)
        assert np.array_equal(d_argb.cpu().numpy().view(np.uint32), argb)
    fr.close()


def test_frame_renderer_single_gpu(gpu_ctx, config2):
    """The torch-resident pipeline object bench.py uses == the host-buffer API."""
    import torch

    from raymarchcl_amd import multigpu

    sc = config2
    gpu_ctx.set_volume(sc["vox"], sc["vres"])
    px, argb = gpu_ctx.render_frame(sc["opts"], sc["mc"], sc["n"])
    fr = multigpu.FrameRenderer(sc["vox"], sc["vres"], sc["opts"], sc["mc"], sc["n"], sc["w"])
    d_px, d_argb = fr.render()
    torch.cuda.synchronize()
    assert _eq(d_px.cpu().numpy(), px)
    assert np.array_equal(d_argb.cpu().numpy().view(np.uint32), argb)
    ms, launches = fr.ctx.last_frame_timing()
    assert launches == 1 and ms > 0  # the whole frame, tonemap included, is one kernel launch
    # the device times of the last frames, readable after a loop (what bench.py's roofline.kernel_ms is made of)
    import time
    seen = len(fr.ctx.frame_timing_history(32))
    t0 = time.perf_counter()
    for _ in range(5):
        fr.render()
        torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) * 1e3
    hist = fr.ctx.frame_timing_history(32)
    assert len(hist) == min(32, seen + 5) and hist[-1] == fr.ctx.last_frame_timing()
    last5 = hist[-5:]
    assert all(k == 1 and 0 < m < 1e3 for m, k in last5)
    assert sum(m for m, _ in last5) <= wall_ms  # the kernels ran inside the frames the host clock timed
    assert len(fr.ctx.frame_timing_history(3)) == 3 and fr.ctx.frame_timing_history(3) == hist[-3:]
    fr.close()


def _runs(ids):
    lo = prev = int(ids[0])
    for v in ids[1:]:
        v = int(v)
        if v != prev + 1:
            yield lo, prev + 1
            lo = v
        prev = v
    yield lo, prev + 1


def test_passes_with_different_options_and_kernel_variants(oracle_mod, native, monkeypatch):
    """Records that differ in more than .time cannot share a pass-packed launch of the frame
    kernel: the host splits the frame into launches that continue from each other's
    accumulator.  Every kernel configuration must agree with the oracle."""
    sc = scenes.build("metal_3spp")
    n = sc["n"]
    opts = bytearray(sc["opts"])
    opts[544 + 260:544 + 264] = np.float32(2.0).tobytes()        # pass 1: other exposure
    opts[2 * 544 + 284] = 100                                   # pass 2: other isoVal
    opts = bytes(opts)
    want, want_argb = oracle_mod.render_frame(sc["vox"], opts, sc["mc"], n)
    for env in ({}, {"RAYMARCH_PASS_PACK": "0"}, {"RAYMARCH_PASS_PACK": "1"}, {"RAYMARCH_PASS_PACK": "6"},
                {"RAYMARCH_POW2": "0"}, {"RAYMARCH_POW2": "0", "RAYMARCH_OCTANTS": "0"},
                {"RAYMARCH_NO_ACCEL": "1"}, {"RAYMARCH_OCTANTS": "0"}, {"RAYMARCH_XCD_ROWS": "0"},
                {"RAYMARCH_OCTANTS": "0", "RAYMARCH_PASS_PACK": "0"}, {"RAYMARCH_BRICKS": "1"},
                {"RAYMARCH_BRICKS": "1", "RAYMARCH_PASS_PACK": "0"}, {"RAYMARCH_PACK_WASTE": "0"},
                {"RAYMARCH_ROW_ORDER": "asc"}, {"RAYMARCH_ROW_ORDER": "asc", "RAYMARCH_PASS_PACK": "0"},
                {"RAYMARCH_ROW_ORDER": "desc"}, {"RAYMARCH_ROW_ORDER": "band", "RAYMARCH_PASS_PACK": "1"},
                {"RAYMARCH_XCD_2D": "0"}, {"RAYMARCH_XCD_2D": "1", "RAYMARCH_PASS_PACK": "0"},
                {"RAYMARCH_XCD_2D": "0", "RAYMARCH_ROW_ORDER": "asc"}, {"RAYMARCH_THIN": "0"},
                {"RAYMARCH_THIN": "0", "RAYMARCH_OCTANTS": "0"}, {"RAYMARCH_THIN": "0", "RAYMARCH_BRICKS": "1"}):
        for k in ("RAYMARCH_NO_ACCEL", "RAYMARCH_POW2", "RAYMARCH_PASS_PACK", "RAYMARCH_OCTANTS", "RAYMARCH_THIN",
                  "RAYMARCH_XCD_ROWS", "RAYMARCH_BRICKS", "RAYMARCH_PACK_WASTE", "RAYMARCH_ROW_ORDER", "RAYMARCH_XCD_2D"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with native.Context(0) as ctx:
            ctx.set_volume(sc["vox"], sc["vres"])
            px, argb = ctx.render_frame(opts, sc["mc"], n)
        assert _eq(px, want), (env, int((px.view(np.uint32) != want.view(np.uint32)).sum()))
        assert np.array_equal(argb, want_argb), env


@pytest.mark.parametrize("name,over", [
    ("ao8", dict(aoIter=7)),                      # 8 probes: the most the shared AO phase posts
    ("ao10_fallback", dict(aoIter=9)),            # more: every owner runs its own probes
    ("ao1", dict(aoIter=0)),
    ("one_light", dict(numLights=1)),
    ("four_lights", dict(numLights=4, lightPos=[[-2, 0, -2, 0], [2, 0, 2, 0], [0, 3, 0, 0], [-3, 1, 2, 0]],
                         lightColor=[[28, 18, 8, 0], [8, 18, 28, 0], [10, 10, 10, 0], [5, 20, 5, 0]])),
    ("attenuation_cut", dict(minLightAtt=0.12)),  # some (point, light) pairs skip their shadow ray
    ("three_bounces", dict(reflectIter=3)),
    ("no_lights", dict(numLights=0)),
])
def test_shared_secondary_rays_against_oracle(oracle_mod, native, name, over):
    """The wave-shared AO / shadow phases (rm_shade.hpp shade_wave) with other probe and light
    counts than the presets use: whole frames (2 passes, pass-packed) == oracle, bit for bit."""
    import raymarchcl_amd as rm
    from raymarchcl_amd import generators as gen
    from raymarchcl_amd import structs

    w, h, it, vres = 48, 40, 2, 64
    vox = scenes.volume("gyroid", vres)
    recs = []
    for i in range(it):
        o = rm.render_options(width=w, height=h, vres=[vres] * 3, t=i * 0.333, iter=it,
                              eyepos=rm.compute_eyepos(150, 2.1, 0.4), targetpos=[0, -0.3, 0], mat="metal")
        o.update(over)
        recs.append(structs.encode_bytes(o))
    opts = b"".join(recs)
    mc = np.stack([gen.generate_scatter_offsets(0x4000, seed=77 + i) for i in range(it)])
    n = w * h
    want, want_argb = oracle_mod.render_frame(vox, opts, mc, n)
    with native.Context(0) as ctx:
        ctx.set_volume(vox, (vres,) * 3)
        px, argb = ctx.render_frame(opts, mc, n)
    assert _eq(px, want), (name, int((px.view(np.uint32) != want.view(np.uint32)).sum()))
    assert np.array_equal(argb, want_argb)
    assert len(np.unique(argb)) > 20


@pytest.mark.parametrize("name,over", [
    ("ao_amp_negative", dict(aoAmp=-0.3)),               # farther hits matter: no AO walk limit from d
    ("ao_step_zero", dict(aoStepDist=0.0)),              # d = 0: division 0/0 in the reference's formula
    ("ao_step_negative", dict(aoStepDist=-0.04)),
    ("ao_step_large", dict(aoStepDist=0.6)),             # probes start outside the box
    ("voxel_size_negative", dict(voxelSize=-0.02)),
    ("voxel_size_large", dict(voxelSize=0.2)),
    ("few_samples_odd", dict(maxVoxelIter=33)),
    ("many_samples", dict(maxVoxelIter=400)),
    ("big_eps", dict(eps=0.08)),
    ("shadow_bias_large", dict(shadowBias=0.6)),
    ("ground_high", dict(groundY=0.2)),                  # ground plane cuts through the volume
    ("ground_far", dict(groundY=40.0)),
    ("max_dist_short", dict(maxDist=2.0)),               # marches end on distance inside the scene
    ("few_turns", dict(maxIter=6, shadowIter=3)),        # marches run out of turns
    ("start_dist", dict(startDist=0.7)),
    # degenerate light vectors (round-5 ADVICE: marches along light directions take |dir| = 1 for their walk limits):
    # a light inside the volume without jitter (hits a few voxels from it: tiny, exactly representable offsets), a light
    # so far that 1/d^2 underflows (no march at all), and one whose distance overflows to inf
    ("light_inside_no_jitter", dict(lightScatter=0.0, lightPos=[[0.0, 0.0, 0.0, 0], [0.25, -0.5, 0.25, 0]])),
    ("light_on_the_ground_plane", dict(lightScatter=0.0, lightPos=[[0.5, -1.05, 0.5, 0], [-0.5, -1.05, 0.25, 0]])),
    ("light_at_1e30", dict(lightPos=[[1e30, 0.0, 0.0, 0], [0.0, 3e19, 0.0, 0]])),
    ("anisotropic_scale", dict(invVoxelScale=[0.5, 0.4, 0.625], voxelBounds=[1.0, 1.25, 0.8],
                               voxelBounds2=[2.0, 2.5, 1.6], voxelBoundsMin=[-0.99, -1.2375, -0.792],
                               voxelBoundsMax=[0.99, 1.2375, 0.792])),
])
def test_walk_limits_with_unusual_records(oracle_mod, native, name, over):
    """The walks of AO probes, shadow marches and (lazily) primary / reflection marches stop
    where a hit can no longer change their result (rm_shade.hpp walk_limit_from): records that
    stress the assumptions behind that bound must still give the oracle's bits."""
    import raymarchcl_amd as rm
    from raymarchcl_amd import generators as gen
    from raymarchcl_amd import structs

    w, h, it, vres = 48, 40, 2, 64
    vox = scenes.volume("gyroid", vres)
    recs = []
    for i in range(it):
        o = rm.render_options(width=w, height=h, vres=[vres] * 3, t=i * 0.333, iter=it,
                              eyepos=rm.compute_eyepos(-60, 2.2, 0.5), targetpos=[0, -0.3, 0], mat="metal")
        o.update(over)
        recs.append(structs.encode_bytes(o))
    opts = b"".join(recs)
    mc = np.stack([gen.generate_scatter_offsets(0x4000, seed=300 + i) for i in range(it)])
    n = w * h
    want = np.zeros(4 * n, np.float32)
    mask = np.zeros(n, np.uint8)
    for i in range(it):
        oracle_mod.render_image(vox, mc[i], opts[i * 544:(i + 1) * 544], want, n=n, undefined_mask=mask)
    with native.Context(0) as ctx:
        ctx.set_volume(vox, (vres,) * 3)
        px, _ = ctx.render_frame(opts, mc, n)
    ok = np.repeat(mask == 0, 4)
    a, b = px.view(np.uint32)[ok], want.view(np.uint32)[ok]
    nan = np.isnan(want[ok])
    assert np.array_equal(a[~nan], b[~nan]), (name, int((a[~nan] != b[~nan]).sum()))
    assert np.isnan(px[ok][nan]).all()
    assert ok.mean() > 0.5


@pytest.mark.parametrize("w,h,spp", [(1280, 88, 2), (1024, 40, 16), (768, 24, 3), (1920, 40, 4)])
def test_frame_is_independent_of_the_xcd_unit_width(native, monkeypatch, w, h, spp):
    """The XCD-aware dispatch order (rm_kernels.hip frame_block) is a permutation of the launch's workgroups: whole tile
    rows per XCD, 2-D units of 1/8 .. 1/64 of a row pair, the width the launcher picks by itself and the plain block order
    must render the same frame bit for bit -- with an odd number of tile rows (88 lines = 11 rows), with widths whose rows
    hold 8 / 16 / 32 / 64 tiles per stripe or not (a unit width the row cannot hold falls back to whole rows), and with
    several passes per wavefront.  The plain order is the one every other test checks against the oracle."""
    sc = scenes.build(dict(vol="gyroid", vres=64, w=w, h=h, iter=spp, mat="orange-stripes", theta=-45, dist=2.25, dof=0.025))
    frames = {}
    for key, env in (("plain", {"RAYMARCH_XCD_ROWS": "0"}), ("auto", {}), ("0", {"RAYMARCH_XCD_2D": "0"}), ("1", {"RAYMARCH_XCD_2D": "1"}),
                     ("2", {"RAYMARCH_XCD_2D": "2"}), ("4", {"RAYMARCH_XCD_2D": "4"}), ("8", {"RAYMARCH_XCD_2D": "8"}),
                     ("3", {"RAYMARCH_XCD_2D": "3"}), ("5", {"RAYMARCH_XCD_2D": "5"}), ("6", {"RAYMARCH_XCD_2D": "6"}),
                     ("2asc", {"RAYMARCH_XCD_2D": "2", "RAYMARCH_ROW_ORDER": "asc"})):
        for k in ("RAYMARCH_XCD_ROWS", "RAYMARCH_XCD_2D", "RAYMARCH_ROW_ORDER"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with native.Context(0) as ctx:
            ctx.set_volume(sc["vox"], sc["vres"])
            px, argb = ctx.render_frame(sc["opts"], sc["mc"], sc["n"])
        frames[key] = (px.view(np.uint32).copy(), argb.copy())
    assert len(np.unique(frames["plain"][0])) > 1000
    for key, (px, argb) in frames.items():
        assert np.array_equal(px, frames["plain"][0]), key
        assert np.array_equal(argb, frames["plain"][1]), key


def test_tables_beyond_4_gib(native, oracle_mod):
    """1024^3: the nine distance tables span 9 GiB, so the directional ones are reached
    through 64-bit offsets.  Volume generated on the device, one small pass == oracle."""
    import raymarchcl_amd as rm
    from raymarchcl_amd import generators as gen
    from raymarchcl_amd import structs

    res, w, h = 1024, 40, 30
    with native.Context(0) as ctx:
        vox = ctx.make_gyroid_volume(res)          # resident + host copy (1 GiB)
        opts = structs.encode_bytes(rm.render_options(
            width=w, height=h, vres=[res] * 3, t=0.0, iter=1, eyepos=rm.compute_eyepos(-45, 2.25, 0.35),
            targetpos=[0, -0.4, 0], mat="metal"))
        mc = gen.generate_scatter_offsets(0x4000, seed=4242)
        n = w * h
        px, _ = ctx.render_frame(opts, mc[None, :], n)
        octs = None
    want = np.zeros(4 * n, np.float32)
    mask = np.zeros(n, np.uint8)
    oracle_mod.render_image(vox, mc, opts, want, n=n, undefined_mask=mask)
    ok = np.repeat(mask == 0, 4)
    assert np.array_equal(px.view(np.uint32)[ok], want.view(np.uint32)[ok])
    assert ok.mean() > 0.9 and len(np.unique(px)) > 100


def test_randomised_parity_smoke():
    """A short run of tools/fuzz_parity.py (random cameras, presets, volumes, record overrides);
    the long runs are recorded in profiles/archive_r01.txt (FILE r01_fuzz_parity.txt)."""
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py")
    spec = importlib.util.spec_from_file_location("fuzz_parity", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    import sys

    argv, sys.argv = sys.argv, ["fuzz_parity.py", "--cases", "40", "--seed", "99"]
    try:
        assert mod.main() == 0
    finally:
        sys.argv = argv


@pytest.mark.parametrize("ao_iter", [7, 8, 11])
def test_records_with_more_ao_probes_than_a_wavefront_exchanges(native, oracle_mod, ao_iter):
    """aoIter + 1 probes per hit: up to 8 the wave-shared AO phase of the frame kernel posts them; beyond, the
    library renders the frame through the single-pass kernels (every lane traces its own probes).  Both routes
    equal the oracle, also as partitions inside the library and in the device-resident form."""
    import raymarchcl_amd as rm
    from raymarchcl_amd import generators as gen, structs

    it, w, h = 3, 40, 28
    recs = []
    for i in range(it):
        o = rm.render_options(width=w, height=h, vres=[64, 64, 64], t=i * 0.333, iter=it,
                              eyepos=rm.compute_eyepos(-45, 2.25, 0.35), targetpos=[0, -0.4, 0], mat="orange-stripes")
        o["aoIter"] = ao_iter
        o["aoStepDist"] = 0.03
        recs.append(structs.encode_bytes(o))
    opts = b"".join(recs)
    mc = np.stack([gen.generate_scatter_offsets(0x4000, seed=50 + i) for i in range(it)])
    vox = scenes.volume("gyroid", 64)
    n = w * h
    want, want_argb = oracle_mod.render_frame(vox, opts, mc, n)
    with native.Context(0) as ctx:
        ctx.set_volume(vox, (64, 64, 64))
        px, argb = ctx.render_frame(opts, mc, n)
    assert np.array_equal(px.view(np.uint32), want.view(np.uint32)) and np.array_equal(argb, want_argb)
    with native.Context([0, 0, 0]) as ctx:
        ctx.set_volume(vox, (64, 64, 64))
        px, argb = ctx.render_frame(opts, mc, n)
        assert np.array_equal(px.view(np.uint32), want.view(np.uint32)) and np.array_equal(argb, want_argb)
        _none, argb2 = ctx.render_frame(opts, mc, n, want_pixels=False)
        assert np.array_equal(argb2, want_argb)


@pytest.mark.parametrize("ao_iter", [7, 8, 11, 16])
def test_any_number_of_ao_probes_through_the_frame_kernel_on_baselines_grids(native, oracle_mod, ao_iter):
    """On the table layouts with the grid edge compiled in (256^3 here: layout 5; BASELINE's 512^3 and 1024^3 likewise) the
    frame kernel takes the AO probes of a record in chunks of 8 (rm_shade.hpp occlusion_wave, kChunkedAO): ONE launch per
    16 passes whatever aoIter (no single-pass route, round 5's cliff), bit-identical to the oracle -- also with an AO
    product that stops early (renderer.cl:338: `ao > 0.01`) inside the first chunk, as partitions inside the library, and
    in the device contracts against the reference kernel itself where it travelled."""
    import raymarchcl_amd as rm
    from raymarchcl_amd import generators as gen, structs

    it, w, h = 4, 64, 40
    vox = scenes.volume("gyroid", 256)

    def records(**over):
        recs = []
        for i in range(it):
            o = rm.render_options(width=w, height=h, vres=[256] * 3, t=i * 0.333, iter=it,
                                  eyepos=rm.compute_eyepos(-45, 2.25, 0.35), targetpos=[0, -0.4, 0], mat="orange-stripes", dof=0.025)
            o["aoIter"] = ao_iter
            o["aoStepDist"] = 0.02
            o.update(over)
            recs.append(structs.encode_bytes(o))
        return b"".join(recs)

    mc = np.stack([gen.generate_scatter_offsets(0x4000, seed=90 + i) for i in range(it)])
    n = w * h
    for over in ({}, dict(aoAmp=3.0)):  # (aoAmp 3: most products fall below 0.01 within the first probes)
        opts = records(**over)
        want, want_argb = oracle_mod.render_frame(vox, opts, mc, n)
        with native.Context(0, contract="cpu") as ctx:
            ctx.set_volume(vox, (256,) * 3)
            px, argb = ctx.render_frame(opts, mc, n)
            ms, launches = ctx.last_frame_timing()
        assert launches == 1, (ao_iter, launches)  # the frame kernel, not one single-pass launch per pass
        assert np.array_equal(px.view(np.uint32), want.view(np.uint32)) and np.array_equal(argb, want_argb), (ao_iter, over)
    with native.Context([0, 0, 0], contract="cpu") as ctx:
        ctx.set_volume(vox, (256,) * 3)
        px, argb = ctx.render_frame(opts, mc, n)
        assert np.array_equal(px.view(np.uint32), want.view(np.uint32)) and np.array_equal(argb, want_argb)
    if oracle_mod.have_gfx950_ref("default"):
        for build, contract in (("default", "gfx950-default"), ("strict", "gfx950-strict")):
            ref_px, ref_argb, _ = oracle_mod.gfx950_render_frame(vox, opts, mc, n, build=build)
            with native.Context(0, contract=contract) as ctx:
                ctx.set_volume(vox, (256,) * 3)
                px, argb = ctx.render_frame(opts, mc, n)
            assert np.array_equal(px.view(np.uint32), ref_px.view(np.uint32)) and np.array_equal(argb, ref_argb), (ao_iter, build)
