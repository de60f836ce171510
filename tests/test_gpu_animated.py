"""Animated volumes (meshvoxel.clj:85-89 make-heatmap-anim -> core.clj:181-213 test-anim): a new volume every frame,
its derived tables built on the library's own stream while the previous frame renders (rm_stage_volume_device /
rm_commit_staged_volume).  Every frame of such a sequence equals the frame of a fresh context that was simply given
that volume -- and therefore the oracle's."""
import numpy as np
import pytest

import scenes

pytestmark = [pytest.mark.gpu]


def _heatmap_volumes(native, res, frames):
    """the reference's animation input: heat-map columns over a moving image (rm_make_heatmap_volume)"""
    vols = []
    with native.Context(0) as g:
        yy, xx = np.mgrid[0:res, 0:res]
        for k in range(frames):
            r = (np.sin(xx * 0.11 + k * 0.9) * 0.5 + 0.5) * 255
            gch = (np.cos(yy * 0.07 - k * 0.6) * 0.5 + 0.5) * 255
            b = ((xx + yy + 13 * k) % 97) * 2.6
            argb = (0xff000000 | (r.astype(np.uint32) << 16) | (gch.astype(np.uint32) << 8) | b.astype(np.uint32)).astype(np.uint32)
            vols.append(g.make_heatmap_volume(argb, 0.8))
    return vols


@pytest.mark.parametrize("res,contract", [(64, "cpu"), (64, None), (256, None)])
def test_staged_volume_sequence_is_bit_identical(native, oracle_mod, res, contract):
    import torch

    frames = 5
    vols = _heatmap_volumes(native, res, frames) if res == 64 else \
        [np.roll(scenes.volume("gyroid", res).reshape(res, res, res), 7 * k, axis=k % 3).reshape(-1).copy() for k in range(frames)]
    sc = scenes.build(dict(vol="gyroid", vres=res, w=96, h=64, iter=4, mat="orange-stripes", theta=-45, dist=2.25, dof=0.025))
    n, w, it = sc["n"], sc["w"], sc["iter"]
    dev = torch.device("cuda", 0)
    d_vol = [torch.from_numpy(v).to(dev) for v in vols]
    d_opts = torch.from_numpy(np.frombuffer(sc["opts"], dtype=np.uint8).copy()).to(dev)
    d_mc = torch.from_numpy(np.ascontiguousarray(sc["mc"], dtype=np.float32)).to(dev)
    d_px = torch.zeros(4 * n, dtype=torch.float32, device=dev)
    d_argb = torch.zeros(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize(dev)
    got = []
    with native.Context(0, contract=contract) as ctx:
        ctx.stage_volume_device(d_vol[0].data_ptr(), (res,) * 3, 32)
        ctx.commit_staged_volume()
        ctx.check_device_opts(d_opts.data_ptr(), it, n, w)
        for k in range(frames):
            ctx.frame_device_full(d_opts.data_ptr(), d_mc.data_ptr(), it, n, w, d_px.data_ptr(), d_argb.data_ptr())
            if k + 1 < frames:
                ctx.stage_volume_device(d_vol[k + 1].data_ptr(), (res,) * 3, 32)  # built beside frame k
            ctx.synchronize()
            got.append((d_px.cpu().numpy().copy(), d_argb.cpu().numpy().view(np.uint32).copy()))
            if k + 1 < frames:
                ctx.commit_staged_volume()  # (same shape: the records stay validated)
        assert ctx.last_table_build_ms() > 0
        # a commit without a stage, and staging on a context with nothing resident yet, are refused / fine
        with pytest.raises(native.RmError):
            ctx.commit_staged_volume()
    assert len({g[1].tobytes() for g in got}) == frames  # five different pictures
    for k in range(frames):
        with native.Context(0, contract=contract) as ref:
            ref.set_volume(vols[k], (res,) * 3)
            px, argb = ref.render_frame(sc["opts"], sc["mc"], n)
        assert np.array_equal(got[k][0].view(np.uint32), px.view(np.uint32)), (k, "float accumulator")
        assert np.array_equal(got[k][1], argb), k
        if contract == "cpu" and k in (0, frames - 1):
            want, want_argb = oracle_mod.render_frame(vols[k], sc["opts"], sc["mc"], n)
            assert np.array_equal(px.view(np.uint32), want.view(np.uint32)) and np.array_equal(argb, want_argb)


def test_staging_from_host_bytes(native):
    """rm_stage_volume: the bytes are copied into a buffer of the library's (what a JNI caller's direct ByteBuffer
    gets); the caller may reuse its array at once; frames equal those of rm_set_volume."""
    vols = [scenes.volume("gyroid", 64), scenes.volume("terrain", 64)]
    sc = scenes.build("orange_dof_2spp")
    with native.Context(0) as ref:
        want = []
        for v in vols:
            ref.set_volume(v, (64,) * 3)
            want.append(ref.render_frame(sc["opts"], sc["mc"], sc["n"]))
    with native.Context(0) as ctx:
        ctx.set_volume(vols[1], (64,) * 3)
        for k in (0, 1, 0):
            scratch = vols[k].copy()
            ctx.stage_volume(scratch, (64,) * 3, 32)
            scratch[:] = 0  # (taken already)
            ctx.commit_staged_volume()
            px, argb = ctx.render_frame(sc["opts"], sc["mc"], sc["n"])
            assert np.array_equal(px.view(np.uint32), want[k][0].view(np.uint32)) and np.array_equal(argb, want[k][1]), k


def test_staging_another_shape_drops_the_validation(native):
    import torch

    dev = torch.device("cuda", 0)
    a = torch.from_numpy(scenes.volume("gyroid", 64)).to(dev)
    b = torch.from_numpy(scenes.volume("gyroid", 32) if False else np.zeros(32 ** 3, np.uint8)).to(dev)
    sc = scenes.build("orange_dof_2spp")
    d_opts = torch.from_numpy(np.frombuffer(sc["opts"], dtype=np.uint8).copy()).to(dev)
    d_mc = torch.from_numpy(np.ascontiguousarray(sc["mc"], dtype=np.float32)).to(dev)
    d_px = torch.zeros(4 * sc["n"], dtype=torch.float32, device=dev)
    with native.Context(0) as ctx:
        ctx.set_volume_device(a.data_ptr(), (64,) * 3)
        ctx.check_device_opts(d_opts.data_ptr(), sc["iter"], sc["n"], sc["w"])
        ctx.frame_device_full(d_opts.data_ptr(), d_mc.data_ptr(), sc["iter"], sc["n"], sc["w"], d_px.data_ptr(), None)
        ctx.stage_volume_device(b.data_ptr(), (32,) * 3, 32)
        ctx.commit_staged_volume()
        with pytest.raises(native.RmError):  # records say voxelRes 64: must be validated against the new volume (and fail there)
            ctx.frame_device_full(d_opts.data_ptr(), d_mc.data_ptr(), sc["iter"], sc["n"], sc["w"], d_px.data_ptr(), None)
        with pytest.raises(native.RmError):
            ctx.check_device_opts(d_opts.data_ptr(), sc["iter"], sc["n"], sc["w"])
    with native.Context([0, 0]) as multi:
        with pytest.raises(native.RmError):
            multi.stage_volume_device(a.data_ptr(), (64,) * 3, 32)
