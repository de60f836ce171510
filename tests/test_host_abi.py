"""The host-side parameter layer of the C ABI (csrc/rm_host.cpp) produces the same
bytes as the Python host layer (which tests/test_host_layer.py pins against the
reference's values).  No GPU needed."""
import ctypes
import math

import numpy as np
import pytest

import raymarchcl_amd as rm
from raymarchcl_amd import generators as gen
from raymarchcl_amd import structs, vio

NAN = float("nan")


def _args(native, **kw):
    a = native.RenderArgs()
    a.width, a.height = kw.get("width", 640), kw.get("height", 360)
    v = kw.get("vres", [256, 256, 256])
    a.vres[0], a.vres[1], a.vres[2] = v
    a.iter, a.t = kw.get("iter", 4), kw.get("t", 0.0)
    for name in ("eyepos", "targetpos"):
        val = kw.get(name)
        for i in range(3):
            getattr(a, name)[i] = NAN if val is None else val[i]
    a.fov_deg = kw.get("fov", NAN) if kw.get("fov") is not None else NAN
    a.dof = kw.get("dof", NAN) if kw.get("dof") is not None else NAN
    a.gamma = kw.get("gamma", NAN) if kw.get("gamma") is not None else NAN
    a.ground_y = kw.get("groundY", NAN) if kw.get("groundY") is not None else NAN
    a.voxel_size = kw.get("voxelSize", NAN) if kw.get("voxelSize") is not None else NAN
    a.mat = kw["mat"].encode() if kw.get("mat") is not None else None
    return a


@pytest.mark.parametrize("kw", [
    dict(mat="orange-stripes", eyepos=rm.compute_eyepos(-45, 2.25, 0.35), targetpos=[0, -0.4, 0], dof=0.025, t=0.333),
    dict(mat="metal", iter=16, t=15 * 0.333), dict(mat="metal2", fov=115, width=1280, height=720),
    dict(mat="ao", vres=[64, 40, 48], voxelSize=0.01, groundY=0.9, gamma=2.2),
    dict(mat="no-such-preset"), dict(mat=None, dof=0.0),
])
def test_render_options_bytes_equal_python(native, kw):
    L = native.lib()
    out = ctypes.create_string_buffer(544)
    a = _args(native, **kw)
    assert L.rm_render_options(ctypes.byref(a), out) == 0
    py = dict(width=640, height=360, vres=[256, 256, 256], iter=4, t=0.0)
    py.update(kw)
    assert out.raw == structs.encode_bytes(rm.render_options(**py))


def test_eyepos_scatter_gyroid_vox(native, tmp_path):
    L = native.lib()
    L.rm_compute_eyepos.argtypes = [ctypes.c_double] * 3 + [ctypes.POINTER(ctypes.c_double * 3)]
    e = (ctypes.c_double * 3)()
    assert L.rm_compute_eyepos(-45.0, 2.25, 0.35, ctypes.byref(e)) == 0
    assert list(e) == rm.compute_eyepos(-45, 2.25, 0.35)
    # scatter table: bit-identical to the numpy generator
    L.rm_make_scatter_table.argtypes = [ctypes.c_uint64, ctypes.c_void_p]
    t = np.zeros(0x4000 * 4, np.float32)
    assert L.rm_make_scatter_table(1003, t.ctypes.data) == 0
    assert np.array_equal(t.view(np.uint32), gen.generate_scatter_offsets(0x4000, seed=1003).view(np.uint32))
    # gyroid volume (same libm cos/sin as numpy)
    L.rm_make_gyroid_host.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p]
    v = np.zeros(64 * 48 * 80, np.uint8)
    assert L.rm_make_gyroid_host(64, 48, 80, v.ctypes.data) == 0
    assert (v != gen.make_gyroid_volume((64, 48, 80))).mean() < 1e-6
    # .vox files are interchangeable with the Python reader / writer
    L.rm_vox_save.argtypes = [ctypes.c_char_p] + [ctypes.c_int] * 3 + [ctypes.c_void_p]
    L.rm_vox_info.argtypes = [ctypes.c_char_p] + [ctypes.POINTER(ctypes.c_int)] * 3
    L.rm_vox_load.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t]
    p = str(tmp_path / "a.vox").encode()
    assert L.rm_vox_save(p, 64, 48, 80, v.ctypes.data) == 0
    back, res = vio.load_volume(p.decode())
    assert res == (64, 48, 80) and np.array_equal(back, v)
    q = str(tmp_path / "b.vox")
    vio.save_volume(q, (64, 48, 80), v)
    rx, ry, rz = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    assert L.rm_vox_info(q.encode(), ctypes.byref(rx), ctypes.byref(ry), ctypes.byref(rz)) == 0
    assert (rx.value, ry.value, rz.value) == (64, 48, 80)
    w = np.zeros(v.size, np.uint8)
    assert L.rm_vox_load(q.encode(), w.ctypes.data, w.size) == 0 and np.array_equal(w, v)
    assert L.rm_vox_load(q.encode(), w.ctypes.data, 10) != 0 and b"too small" in L.rm_last_error()
    assert L.rm_vox_info(b"/nonexistent.vox", ctypes.byref(rx), ctypes.byref(ry), ctypes.byref(rz)) != 0


def test_volume_band_is_the_rows_where_the_clip_box_covers_most(native):
    """rm_debug_volume_band (the tile rows the frame kernel dispatches first) against a brute-force slab test of every
    pixel's central ray (renderer.cl:456-465 camera, :153-161 slab test): the band is the run of tile rows in which the box
    covers at least half as much of the width as in the best row; no band when that is every row (BASELINE's camera: the
    box fills the view) or when the box is not in the image."""
    import numpy as np

    import raymarchcl_amd as rm
    from raymarchcl_amd import structs

    def coverage(o, w, h):
        eye, tgt, up = np.array(o["eyePos"], float), np.array(o["targetPos"], float), np.array(o["up"], float)
        f = tgt - eye
        f /= np.linalg.norm(f)
        r = np.cross(f, up)
        r /= np.linalg.norm(r)
        u = np.cross(r, f)
        px, py = np.meshgrid(np.arange(w) + 0.5, np.arange(h) + 0.5)
        vx = px / w * o["fov"] - o["fov"] * 0.5
        vy = -(py / h * o["fov"] - o["fov"] * 0.5) * o["invAspect"]
        d = r[None, None] * vx[..., None] + u[None, None] * vy[..., None] + f[None, None]
        with np.errstate(divide="ignore", invalid="ignore"):
            lo = (np.array(o["voxelBoundsMin"]) - eye) / d
            hi = (np.array(o["voxelBoundsMax"]) - eye) / d
        near = np.maximum(np.minimum(lo, hi).max(axis=-1), 0.0)
        far = np.maximum(lo, hi).min(axis=-1)
        return (far > near)[4::8].mean(axis=1)  # the centre line of every tile row

    cases = [dict(eyepos=rm.compute_eyepos(-45, 2.25, 0.35), targetpos=[0, -0.4, 0], dof=0.025),   # BASELINE configs: no band
             dict(eyepos=rm.compute_eyepos(20, 5.0, 1.5), targetpos=[0, -0.15, 0]),               # a distant camera
             dict(eyepos=rm.compute_eyepos(135, 6.0, 3.5), targetpos=[0, 0, 0], fov=60),
             dict(eyepos=rm.compute_eyepos(-45, 4.0, 0.35), targetpos=[0, 1.5, 0]),               # the box low in the image
             dict(eyepos=rm.compute_eyepos(10, 0.6, 0.3), targetpos=[0.2, 0.1, 0.0]),             # inside the box
             dict(eyepos=rm.compute_eyepos(0, 4.0, 0.0), targetpos=[0, 6.0, -4.0])]               # the box out of view
    seen = set()
    for k, kw in enumerate(cases):
        w, h = 320, 184
        o = rm.render_options(width=w, height=h, vres=[64] * 3, iter=1, mat="orange-stripes", **kw)
        lo, hi = native.volume_band(structs.encode_bytes(o))
        cov = coverage(o, w, h)
        rows = len(cov)
        assert 0.0 <= lo <= hi <= 1.0
        want = np.nonzero(cov >= 0.5 * cov.max())[0] if cov.max() > 0 else np.array([], int)
        if hi > lo:
            seen.add("band")
            r0, r1 = lo * rows, hi * rows
            assert abs(r0 - want.min()) <= 1.01 and abs(r1 - (want.max() + 1)) <= 1.01, (k, r0, r1, want.min(), want.max())
            assert r1 - r0 < rows
        else:
            seen.add("none")
            assert want.size == 0 or (want.min() <= 1 and want.max() >= rows - 2), (k, want)
    assert seen == {"band", "none"}
    for cfg_w, cfg_h in ((1280, 720), (1920, 1080), (3840, 2160)):  # BASELINE's camera: the box fills the view, plain order
        o = rm.render_options(width=cfg_w, height=cfg_h, vres=[256] * 3, iter=16, mat="orange-stripes", **cases[0])
        assert native.volume_band(structs.encode_bytes(o)) == (0.0, 0.0)


def _check_block_order(native, w, h, passes, first, stride, **kw):
    """every (tile, sub-block) of the partition exactly once; -> (hardware blocks, padding blocks)"""
    n = w * h
    order = native.block_order(w, n, passes, tile_first=first, tile_stride=stride, **kw)
    tiles_x, tiles_y = (w + 7) // 8, (h + 7) // 8
    tiles_total = tiles_x * tiles_y
    pp = 1
    while pp < passes:
        pp *= 2
    mine = np.arange(first, tiles_total, stride, dtype=np.int64)
    want = np.sort(((mine[:, None] << 8) | np.arange(pp, dtype=np.int64)[None, :]).ravel())
    got = np.sort(order[order >= 0])
    assert np.array_equal(got, want), (w, h, passes, first, stride, kw, got.size, want.size)
    return order.size, int((order < 0).sum())


def test_dispatch_order_of_the_frame_kernel_is_a_permutation(native):
    """rm_debug_block_order = the kernel's own block mapping (rm_kernels.hip logical_block) evaluated on the host: whatever
    the order -- plain, whole tile rows per XCD with the last rows in equal shares, 2-D units of every width, bottom to
    top, a band of rows first -- every sub-block of every tile of the launch's partition is rendered by exactly one
    hardware workgroup.  BASELINE's five image sizes, odd and tiny images, widths whose rows hold 8 / 16 / 32 / 64 tiles or
    none of them, 1..32 passes per wavefront, the tile partitions of 2 / 4 / 8 ranks."""
    shapes = [(1280, 720), (256, 256), (1920, 1080), (3840, 2160), (1280, 88), (1024, 40), (768, 24), (512, 8), (64, 64),
              (1283, 77), (17, 9), (8, 8), (2048, 16), (640, 360), (1280, 8)]
    checked = 0
    for w, h in shapes:
        big = w * h > 3_000_000
        for passes in ((16,) if big else (1, 2, 3, 16, 25, 32)):
            for stride in ((1, 8) if big else (1, 2, 4, 8)):
                for first in sorted({0, stride - 1}):
                    for kw in (dict(xcd_rows=False), dict(xcd_2d=0), dict(xcd_2d=0, rows_desc=False), dict(), dict(xcd_2d=1),
                               dict(xcd_2d=2), dict(xcd_2d=4), dict(xcd_2d=8, rows_desc=False), dict(band=(0.33, 0.83)),
                               dict(xcd_2d=3), dict(xcd_2d=5, rows_desc=False), dict(xcd_2d=6),
                               dict(xcd_2d=0, band=(0.0, 0.5))):
                        if big and kw.get("xcd_2d", -1) in (1, 3, 8):
                            continue
                        blocks, padding = _check_block_order(native, w, h, passes, first, stride, **kw)
                        checked += 1
                        # the padding of the XCD-aware grid stays small next to a real frame
                        if w * h >= 640 * 360 and stride == 1:
                            assert padding <= 0.02 * blocks, (w, h, passes, kw, blocks, padding)
    assert checked > 2000
    # the 2-D units need no padding at all where they apply, and the launcher's own choice of the width applies to BASELINE's sizes
    for w, h in ((1280, 720), (1920, 1080), (3840, 2160)):
        order = native.block_order(w, w * h, 16)
        assert (order >= 0).all() and order.size == (w // 8) * ((h + 7) // 8) * 16


def test_block_order_rejects_bad_arguments(native):
    for args in ((0, 64, 1), (64, 0, 1), (64, 64, 0), (64, 64, 65)):
        with pytest.raises(native.RmError):
            native.block_order(*args)
    with pytest.raises(native.RmError):
        native.block_order(64, 64, 1, tile_stride=0)
