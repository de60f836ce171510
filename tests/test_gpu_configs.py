"""Parity at the configurations BASELINE.json lists and bench.py measures, with the library's
DEFAULT kernel configuration (what the bench line is produced with): the inputs come from
bench.build_inputs, the frame runs through the same entry points as the bench, and the
oracle renders a sample of the work-items -- every pass, in order -- plus the tonemap.

  C2  256^3 gyroid, 1280x720, 16 spp + DOF  (the headline; also with frames in flight)
  C3  512^3 blob volume (bunny stand-in), 1920x1080, 16 spp, :metal
  C4  256^3 gyroid, 3840x2160, 64 spp       (also as an 8-way tile partition + resolve)
  C5  1024^3 gyroid (dragon stand-in), 1920x1080, 25 spp, :metal

Bit-exact (float32, pinned IEEE op sequence).  Work-items whose material index falls outside
the record (undefined in the reference, renderer.cl:394,418) are excluded."""
import os
import sys

import numpy as np
import pytest

import scenes

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("cpu_contract")]  # this module checks against the CPU oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _eq(a, b):
    return np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32))


def _sample_ids(n, width, count, seed, rows=()):
    rng = np.random.default_rng(seed)
    parts = [rng.integers(0, n, count), [0, n - 1]]
    for r in rows:  # a run of neighbouring work-items (whole 8x8 tiles' rows)
        parts.append(np.arange(width * r, width * r + min(width, 256)))
    return np.unique(np.concatenate(parts)).astype(np.int32)


def _check_against_oracle(oracle_mod, vox, opts, mc, n, ids, px, argb):
    mask = np.zeros(n, np.uint8)
    want = oracle_mod.render_frame_ids(vox, opts, mc, n, ids, undefined_mask=mask)
    ok = ids[mask[ids] == 0]
    assert len(ok) > 0.9 * len(ids)
    a, b = px.reshape(-1, 4)[ok], want.reshape(-1, 4)[ok]
    bad = int((a.view(np.uint32) != b.view(np.uint32)).any(axis=1).sum())
    assert bad == 0, f"{bad} of {len(ok)} sampled work-items differ from the oracle"
    if argb is not None:
        assert np.array_equal(argb[ok], oracle_mod.tonemap_image(want, opts[:544], n=n)[ok])
    assert len(np.unique(a[:, :3])) > len(ok) // 4  # a real image, not a constant


def _bench_inputs(name):
    import bench

    wl = bench.WORKLOADS[name]
    vox, vres, opts, mc = bench.build_inputs(wl)
    return wl, vox, vres, opts, mc


def test_c2_headline_16spp_default_kernel(native, oracle_mod):
    """BASELINE configs[1] exactly as bench.py runs it: rm_frame_device + rm_resolve_device on
    torch streams, default pass packing (16 passes x 4 pixels per wavefront), 3 frames in flight."""
    import torch

    from raymarchcl_amd import multigpu

    wl, vox, vres, opts, mc = _bench_inputs("c2")
    n, w = wl["w"] * wl["h"], wl["w"]
    assert len(opts) == 16 * 544 and "RAYMARCH_PASS_PACK" not in os.environ
    fr = multigpu.FrameRenderer(vox, vres, opts, mc, n, w, frames_in_flight=3)
    outs = [fr.render() for _ in range(4)]  # slot 0 is reused by the 4th frame
    torch.cuda.synchronize()
    px = outs[0][0].cpu().numpy()
    argb = outs[0][1].cpu().numpy().view(np.uint32)
    for d_px, d_argb in outs[1:]:
        assert _eq(d_px.cpu().numpy(), px)
    fr.close()
    ids = _sample_ids(n, w, 2500, 11, rows=(300, 500))
    _check_against_oracle(oracle_mod, vox, opts, mc, n, ids, px, argb)
    # the host-buffer boundary gives the same frame
    with native.Context(0) as ctx:
        ctx.set_volume(vox, vres)
        hpx, hargb = ctx.render_frame(opts, mc, n)
    assert _eq(hpx, px) and np.array_equal(hargb, argb)


@pytest.mark.parametrize("passes,pack", [(8, "3"), (16, "4"), (16, "3"), (8, "4"), (12, "4"), (25, "4"), (32, "4")])
def test_pass_packed_lane_maps_whole_frames(native, oracle_mod, monkeypatch, passes, pack):
    """8 / 16 / odd pass counts with 8 or 16 passes per wavefront: whole small frames == oracle."""
    spec = dict(vol="gyroid", vres=64, w=56, h=40, iter=passes, mat="metal", theta=-30, dist=2.2, dof=0.02)
    sc = scenes.build(spec, mc_seed=500)
    monkeypatch.setenv("RAYMARCH_PASS_PACK", pack)
    want, want_argb = oracle_mod.render_frame(sc["vox"], sc["opts"], sc["mc"], sc["n"])
    with native.Context(0) as ctx:
        ctx.set_volume(sc["vox"], sc["vres"])
        px, argb = ctx.render_frame(sc["opts"], sc["mc"], sc["n"])
    assert _eq(px, want), int((px.view(np.uint32) != want.view(np.uint32)).sum())
    assert np.array_equal(argb, want_argb)


def _single_pass_kernel_equals_frame_kernel(ctx, opts, mc, n, w, row):
    """The single-pass kernel (rm_render_image_range) of the volume's table layout against the frame kernel on a
    one-pass frame, on a band of eight rows: the two kernels instantiate the same walk for the layout."""
    one, _ = ctx.render_frame(opts[:544], mc[:1], n)
    band = np.zeros(4 * n, np.float32)
    id0, id1 = row * w, (row + 8) * w
    ctx.render_image(np.ascontiguousarray(mc[0]), opts[:544], band, n=n, id0=id0, id1=id1)
    assert _eq(band[4 * id0:4 * id1], one[4 * id0:4 * id1])


def test_c3_512_blobs_1080p_16spp(native, oracle_mod):
    wl, vox, vres, opts, mc = _bench_inputs("c3")
    n, w = wl["w"] * wl["h"], wl["w"]
    with native.Context(0) as ctx:
        ctx.set_volume(vox, vres)
        px, argb = ctx.render_frame(opts, mc, n)
        _single_pass_kernel_equals_frame_kernel(ctx, opts, mc, n, w, 536)  # (table layout 3: bricks of the 512^3 grid)
    ids = _sample_ids(n, w, 1500, 12, rows=(540,))
    _check_against_oracle(oracle_mod, vox, opts, mc, n, ids, px, argb)


def test_c4_4k_64spp_and_8_way_partition(native, oracle_mod):
    import torch

    from raymarchcl_amd import multigpu

    wl, vox, vres, opts, mc = _bench_inputs("c4")
    n, w = wl["w"] * wl["h"], wl["w"]
    with native.Context(0) as ctx:
        ctx.set_volume(vox, vres)
        px, argb = ctx.render_frame(opts, mc, n)
        ids = _sample_ids(n, w, 1200, 13, rows=(1080,))
        _check_against_oracle(oracle_mod, vox, opts, mc, n, ids, px, argb)
        # the multi-GPU split of this config: 8 interleaved tile partitions rendered one after the
        # other on this GPU, "gathered" by concatenation, resolved on the device == the full frame
        dev = torch.device("cuda:0")
        d_opts = torch.frombuffer(bytearray(opts), dtype=torch.uint8).to(dev)
        d_mc = torch.from_numpy(np.ascontiguousarray(mc)).to(dev)
        parts = 8
        tpp = multigpu.tiles_per_part(w, n, parts)
        d_all = torch.zeros(parts * tpp * 64 * 4, dtype=torch.float32, device=dev)
        ctx.check_device_opts(d_opts.data_ptr(), 64, n, w)
        for r in range(parts):
            part = d_all[r * tpp * 256:(r + 1) * tpp * 256]
            ctx.frame_device(d_opts.data_ptr(), d_mc.data_ptr(), 64, n, w, part.data_ptr(), r, parts)
        d_px = torch.empty(4 * n, dtype=torch.float32, device=dev)
        d_argb = torch.empty(n, dtype=torch.int32, device=dev)
        ctx.resolve_device(d_all.data_ptr(), parts, d_opts.data_ptr(), n, w, d_px.data_ptr(), d_argb.data_ptr())
        ctx.synchronize()
        assert _eq(d_px.cpu().numpy(), px)
        assert np.array_equal(d_argb.cpu().numpy().view(np.uint32), argb)


def test_c5_1024_volume_1080p_25spp(native, oracle_mod):
    wl, vox, vres, opts, mc = _bench_inputs("c5")  # volume generated on the device, 1 GiB host copy
    n, w = wl["w"] * wl["h"], wl["w"]
    with native.Context(0) as ctx:
        ctx.set_volume(vox, vres)
        px, argb = ctx.render_frame(opts, mc, n)
        _single_pass_kernel_equals_frame_kernel(ctx, opts, mc, n, w, 600)  # (table layout 4: bricks of the 1024^3 grid)
    ids = _sample_ids(n, w, 1000, 14, rows=(600,))
    _check_against_oracle(oracle_mod, vox, opts, mc, n, ids, px, argb)


def test_launches_continue_each_others_accumulators(native, oracle_mod, monkeypatch):
    """A frame whose records change in the middle: 18 equal passes (8 + 8 + 2: a launch holds what one
    wavefront holds), one pass with another exposure, two passes with another isoVal (own tables) --
    five launches of the frame kernel, each continuing from the accumulator the previous one left,
    the last one tonemapping."""
    spec = dict(vol="gyroid", vres=64, w=44, h=36, iter=21, mat="metal2", theta=40, dist=2.3, dof=0.01)
    sc = scenes.build(spec, mc_seed=900)
    opts = bytearray(sc["opts"])
    opts[18 * 544 + 260:18 * 544 + 264] = np.float32(1.7).tobytes()  # exposure of pass 18
    for i in (19, 20):
        opts[i * 544 + 284] = 90                                      # isoVal of the last two
    opts = bytes(opts)
    want, want_argb = oracle_mod.render_frame(sc["vox"], opts, sc["mc"], sc["n"])
    # (a frame this small is a THIN launch: by default its wavefronts hold 16 passes of fewer pixels -- 16 + 2, four
    #  launches; RAYMARCH_THIN=0 keeps the packing of a full-size frame)
    for thin, expect in (("0", 5), (None, 4)):
        monkeypatch.delenv("RAYMARCH_THIN", raising=False)
        if thin is not None:
            monkeypatch.setenv("RAYMARCH_THIN", thin)
        for ranks in (1, 3):
            with native.Context([0] * ranks if ranks > 1 else 0) as ctx:
                ctx.set_volume(sc["vox"], sc["vres"])
                px, argb = ctx.render_frame(opts, sc["mc"], sc["n"])
                ms, launches = ctx.last_frame_timing()
            assert launches == expect, (thin, ranks, launches)
            assert _eq(px, want), (ranks, int((px.view(np.uint32) != want.view(np.uint32)).sum()))
            assert np.array_equal(argb, want_argb)
