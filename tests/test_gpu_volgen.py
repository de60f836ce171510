"""GPU: the device-side volume producers (csrc/rm_volgen.hip, through the C ABI) against
the numpy restatements of the reference loops (oracle/volgen_np.py) -- bit-exact, except
terrain columns whose height is within an ulp of an integer (device sin/cos)."""
import numpy as np
import pytest

from oracle import volgen_np as vg

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("cpu_contract")]  # this module checks against the CPU oracle


def test_terrain_device_matches_restatement(gpu_ctx):
    for res in [(32, 24, 32), (64, 48, 80), (96, 64, 96)]:
        got = gpu_ctx.make_terrain_volume(res)
        want = vg.make_terrain(*res)
        diff = np.count_nonzero(got != want)
        assert diff <= got.size * 1e-4, (res, diff)
        assert gpu_ctx.vres == tuple(res)
    with pytest.raises(Exception):
        gpu_ctx.make_terrain_volume((2, 8, 8))


def _cloud(seed, n):
    rng = np.random.default_rng(seed)
    t = rng.uniform(0, 2 * np.pi, n)
    u = rng.uniform(-1, 1, n)
    r = np.sqrt(1 - u * u)
    return np.stack([3.0 * r * np.cos(t) + 10.0, 1.5 * r * np.sin(t) - 4.0, 0.75 * u], axis=1)


@pytest.mark.parametrize("res,ks", [(16, -1), (32, -1), (32, 0), (32, 2), (48, 3), (8, 20)])
def test_vertex_splat_matches_restatement(gpu_ctx, res, ks):
    verts = _cloud(res + ks, 5000)
    got = gpu_ctx.voxelize_vertices(verts, res, ks)
    want = vg.voxelize(verts, res) if ks < 0 else vg.voxelize_ks(verts, res, ks)
    assert np.array_equal(got, want)
    assert got.any()


def test_vertex_splat_edge_cases(gpu_ctx):
    assert not gpu_ctx.voxelize_vertices(np.zeros((0, 3)), 8, -1).any()      # empty mesh -> empty grid
    with pytest.raises(Exception):
        gpu_ctx.voxelize_vertices(np.ones((4, 3)), 8, -1)                    # zero extent
    with pytest.raises(Exception):
        gpu_ctx.voxelize_vertices(np.array([[0, 0, np.nan], [1, 1, 1.0]]), 8, -1)
    # the max corner lands on cell == res and is dropped by voxelize, clipped in by voxelize-ks
    verts = np.array([[0, 0, 0], [1, 1, 1]], np.float64)
    v = gpu_ctx.voxelize_vertices(verts, 4, -1)
    assert v.sum() == 255 and v[0] == 255
    k = gpu_ctx.voxelize_vertices(verts, 4, 1).reshape(4, 4, 4)
    assert k[0:2, 0:2, 0:2].all() and k[3, 3, 3] == 255 and np.array_equal(k.reshape(-1), vg.voxelize_ks(verts, 4, 1))


def test_generated_volume_is_resident_and_renders(gpu_ctx, oracle_mod):
    """A splatted volume feeds the render path like any other: it is the resident volume
    right after the call, and the GPU pass over it == the oracle's pass over the same bytes."""
    import scenes

    sc = scenes.build(dict(vol="empty", vres=32, w=32, h=24, iter=1, mat="metal", theta=30, dist=2.25))
    vox = gpu_ctx.voxelize_vertices(_cloud(3, 20000), 32, 1)   # resident now
    px = np.zeros(4 * sc["n"], np.float32)
    gpu_ctx.render_image(sc["mc"][0], sc["opts"][:544], px, n=sc["n"])
    ref = np.zeros(4 * sc["n"], np.float32)
    mask = np.zeros(sc["n"], np.uint8)
    oracle_mod.render_image(vox, sc["mc"][0], sc["opts"][:544], ref, n=sc["n"], undefined_mask=mask)
    ok = np.repeat(mask == 0, 4)
    assert np.array_equal(px.view(np.uint32)[ok], ref.view(np.uint32)[ok])
    assert ok.mean() > 0.99 and len(np.unique(px)) > 50        # the object is in view


def test_heatmap_matches_restatement(gpu_ctx):
    rng = np.random.default_rng(11)
    px = (rng.integers(0, 256, (24, 24)).astype(np.uint32) | 0xFF000000)
    px[rng.random((24, 24)) < 0.3] = 0xFF000000
    px[0, 0] = 0xFFABCDE1  # > 224
    for amp in (0.0, 0.03125, float(np.float32(7 / (10 * 1.33333))), 0.75):
        assert np.array_equal(gpu_ctx.make_heatmap_volume(px, amp), vg.make_heatmap(px, amp))


@pytest.mark.parametrize("res,seed", [(32, 0), (48, 12345), (33, 7)])
def test_seeded_scatter_voxeliser_matches_restatement(gpu_ctx, res, seed):
    """voxelize-scatter (meshvoxel.clj:25-43) with seeded draws: device kernel == the loop-by-loop restatement."""
    verts = _cloud(11, 3000)
    got = gpu_ctx.voxelize_scatter(verts, res, seed=seed)
    want = vg.voxelize_scatter(verts, res, seed=seed)
    assert got.dtype == np.uint8 and set(np.unique(got)) <= {0, 64}
    assert np.array_equal(got, want)
    assert 0 < int((got == 64).sum()) < res ** 3
    if seed:
        assert not np.array_equal(got, gpu_ctx.voxelize_scatter(verts, res, seed=seed + 1))  # the draws matter
    assert not gpu_ctx.voxelize_scatter(np.zeros((0, 3)), 8, seed=1).any()
