"""The derived structures (csrc/rm_accel.hip) against independent host
computations: dist8 == chessboard distance transform (scipy) of the padded
"hit or outside" mask, surf32 == voxel value + the normal terms the reference
algorithm would sum (numpy)."""
import numpy as np
import pytest

import scenes

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("cpu_contract")]  # this module checks against the CPU oracle


@pytest.mark.parametrize("kind,vres,iso", [("gyroid", 64, 32), ("gyroid-crop", (64, 40, 48), 32),
                                           ("terrain", 64, 32), ("blobs", 64, 100),
                                           ("empty", 32, 32), ("solid", 32, 32)])
def test_dist8_and_surf32(gpu_ctx, kind, vres, iso):
    from scipy import ndimage

    vox = scenes.volume(kind, vres)
    rx, ry, rz = (vres,) * 3 if isinstance(vres, int) else vres
    gpu_ctx.set_volume(vox, (rx, ry, rz))
    dist, surf = gpu_ctx.debug_get_accel(iso)
    g = vox.reshape(rz, ry, rx)
    empty = np.zeros((rz + 2, ry + 2, rx + 2), dtype=bool)  # outside the grid counts as hit
    empty[1:-1, 1:-1, 1:-1] = g <= iso
    want = ndimage.distance_transform_cdt(empty, metric="chessboard")[1:-1, 1:-1, 1:-1]
    want = np.minimum(want, 255).astype(np.uint8)
    assert np.array_equal(dist.reshape(rz, ry, rx), want)
    # surf32: value, flat central differences, smooth 3x3x3 sums (occupancy is v >= iso)
    occ = np.zeros((rz + 4, ry + 4, rx + 4), dtype=np.int32)
    occ[2:-2, 2:-2, 2:-2] = g >= iso
    gx = occ[:, :, 2:] - occ[:, :, :-2]
    gy = occ[:, 2:, :] - occ[:, :-2, :]
    gz = occ[2:, :, :] - occ[:-2, :, :]
    gx = gx[1:-1, 1:-1, :]; gy = gy[1:-1, :, 1:-1]; gz = gz[:, 1:-1, 1:-1]  # (rz+2, ry+2, rx+2)
    o1 = occ[1:-1, 1:-1, 1:-1]
    sx = np.zeros((rz, ry, rx), np.int32); sy = sx.copy(); sz = sx.copy()
    for dz in range(3):
        for dy in range(3):
            for dx in range(3):
                sl = (slice(dz, dz + rz), slice(dy, dy + ry), slice(dx, dx + rx))
                sx -= o1[sl] * gx[sl]; sy -= o1[sl] * gy[sl]; sz -= o1[sl] * gz[sl]
    c = (slice(1, -1),) * 3
    hit = g > iso
    s = surf.reshape(rz, ry, rx)
    assert np.array_equal(s & 0xff, g)
    assert np.array_equal(((s >> 8) & 63).astype(np.int32)[hit] - 32, sx[hit])
    assert np.array_equal(((s >> 14) & 63).astype(np.int32)[hit] - 32, sy[hit])
    assert np.array_equal(((s >> 20) & 63).astype(np.int32)[hit] - 32, sz[hit])
    assert np.array_equal(((s >> 26) & 3).astype(np.int32)[hit] - 1, gx[c][hit])
    assert np.array_equal(((s >> 28) & 3).astype(np.int32)[hit] - 1, gy[c][hit])
    assert np.array_equal(((s >> 30) & 3).astype(np.int32)[hit] - 1, gz[c][hit])


def test_device_gyroid_generator_matches_host(native):
    """rm_make_gyroid_volume == generators.make_gyroid_volume up to voxels whose value
    lies within rounding of a threshold (device vs host cos/sin)."""
    from raymarchcl_amd import generators

    with native.Context(0) as ctx:
        for res in (64, (96, 64, 128)):
            got = ctx.make_gyroid_volume(res)
            want = generators.make_gyroid_volume(res)
            assert got.shape == want.shape and set(np.unique(got)) <= {0, 64, 128, 255}
            assert (got != want).mean() < 1e-5
        # it is the resident volume now: rendering works without rm_set_volume
        sc = scenes.build("c1_orange")
        ctx.make_gyroid_volume(64, want_host_copy=False)
        px, _ = ctx.render_frame(sc["opts"], sc["mc"], sc["n"])
        assert np.isfinite(px).all() and px.any()


def _oct_reference(hit):
    """Largest empty cube ahead of every cell for all 8 sign combinations, by brute force:
    n(q) = #{n >= 1 : the n^3 box with corner q is inside the grid and holds no hit cell}
    (box emptiness is monotone in n), boxes summed with a padded cumulative table."""
    rz, ry, rx = hit.shape
    out = np.zeros((8, rz, ry, rx), np.uint8)
    for o in range(8):
        flips = tuple(ax for ax, bit in ((2, 1), (1, 2), (0, 4)) if o & bit)  # x=bit0, y=bit1, z=bit2
        h = np.flip(hit, flips) if flips else hit
        sat = np.zeros((rz + 1, ry + 1, rx + 1), np.int64)
        sat[1:, 1:, 1:] = h.astype(np.int64).cumsum(0).cumsum(1).cumsum(2)
        n_tab = np.zeros((rz, ry, rx), np.int32)
        z, y, x = np.meshgrid(np.arange(rz), np.arange(ry), np.arange(rx), indexing="ij")
        for n in range(1, min(255, max(rx, ry, rz)) + 1):
            ok = (z + n <= rz) & (y + n <= ry) & (x + n <= rx)
            if not ok.any():
                break
            z1, y1, x1 = np.minimum(z + n, rz), np.minimum(y + n, ry), np.minimum(x + n, rx)
            s = (sat[z1, y1, x1] - sat[z, y1, x1] - sat[z1, y, x1] - sat[z1, y1, x] + sat[z, y, x1]
                 + sat[z, y1, x] + sat[z1, y, x] - sat[z, y, x])
            n_tab += (ok & (s == 0))
        t = n_tab.astype(np.uint8)
        out[o] = np.flip(t, flips) if flips else t
    return out


@pytest.mark.parametrize("kind,vres,iso", [("gyroid", 32, 32), ("gyroid-crop", (64, 40, 48), 32),
                                           ("blobs", 32, 100), ("empty", 16, 32), ("solid", 16, 32)])
def test_directional_tables(gpu_ctx, kind, vres, iso):
    vox = scenes.volume(kind, vres)
    rx, ry, rz = (vres,) * 3 if isinstance(vres, int) else vres
    gpu_ctx.set_volume(vox, (rx, ry, rz))
    got = gpu_ctx.debug_get_octants(iso)
    want = _oct_reference(vox.reshape(rz, ry, rx) > iso)
    assert np.array_equal(got, want)
    # every directional value is at least the undirected one (the centred cube of dist8
    # contains a cube of edge d ahead of the cell)
    dist, _ = gpu_ctx.debug_get_accel(iso)
    assert (got >= dist.reshape(1, rz, ry, rx)).all()


@pytest.mark.parametrize("kind,vres", [("gyroid", 64), ("gyroid-crop", (64, 40, 48)), ("terrain", 64)])
def test_bricked_tables_hold_the_same_values(native, monkeypatch, kind, vres):
    """The 128-byte-brick layout of dist8 / oct8 (default from 4 GiB of tables, forced here) is the
    same data: un-bricked through the test hooks it equals the row-major build, cell for cell
    (non-multiples of the brick size included)."""
    vox = scenes.volume(kind, vres)
    res = (vres,) * 3 if isinstance(vres, int) else vres
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("RAYMARCH_BRICKS", mode)
        with native.Context(0) as ctx:
            ctx.set_volume(vox, res)
            out[mode] = (ctx.debug_get_accel(32)[0], ctx.debug_get_octants(32))
    assert np.array_equal(out["0"][0], out["1"][0])
    assert np.array_equal(out["0"][1], out["1"][1])
