"""Host mirror of thi.ng.raymarchcl.meshvoxel (meshvoxel.clj): mesh / image -> byte volume.

Same names and argument meaning as the reference; the fill itself runs on the GPU
through the C ABI (rm_voxelize_vertices, rm_make_heatmap_volume) -- there is no CPU
fallback.  Volumes come back as flat uint8 arrays (index z*res^2 + y*res + x), ready
for vio.save_volume / core.init_renderer.
"""
import struct

import numpy as np

from . import _native


def load_mesh(path):
    """meshvoxel.clj:12-14 (mio/read-stl): binary STL -> vertices, float64 [3*n_tri, 3].

    Layout: 80-byte header, uint32 triangle count, per triangle 12 float32 (normal,
    3 vertices) + uint16 attribute, little endian.  (ASCII STL: 'solid' + text.)"""
    with open(path, "rb") as f:
        data = f.read()
    if len(data) >= 84:
        (ntri,) = struct.unpack_from("<I", data, 80)
        if 84 + 50 * ntri == len(data):
            rec = np.frombuffer(data, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]),
                                count=ntri, offset=84)
            return rec["v"].reshape(-1, 3).astype(np.float64)
    if data[:5].lower() == b"solid":
        verts = [ln.split()[1:4] for ln in data.decode("ascii", "replace").splitlines()
                 if ln.strip().lower().startswith("vertex")]
        return np.asarray(verts, dtype=np.float64).reshape(-1, 3)
    raise ValueError(f"{path}: not an STL file")


def mesh_scale(vertices, res):
    """meshvoxel.clj:16-25 -> (p, off, s): v maps to off + (v - p) * s.
    (What the device path computes itself; exposed for inspection like the reference's prn.)"""
    v = np.asarray(vertices, dtype=np.float64).reshape(-1, 3)
    p = v.min(axis=0)
    size = v.max(axis=0) - p
    md = size.max()
    off = (0.5 * res) * (1.0 - size / md)
    return p, off, res / md


def _ctx(ctx):
    return ctx if ctx is not None else _native.Context(0)


def voxelize(vertices, res, ctx=None):
    """meshvoxel.clj:61-71: every vertex sets its own cell to 255 (-1 as a signed byte)."""
    c = _ctx(ctx)
    try:
        return c.voxelize_vertices(vertices, res, ks=-1)
    finally:
        if ctx is None:
            c.close()


def voxelize_ks(vertices, res, ks, ctx=None):
    """meshvoxel.clj:47-59: every vertex sets the clipped (2*ks+1)^3 cube around its cell."""
    if ks < 0:
        raise ValueError("ks must be >= 0")
    c = _ctx(ctx)
    try:
        return c.voxelize_vertices(vertices, res, ks=int(ks))
    finally:
        if ctx is None:
            c.close()


def voxelize_scatter(vertices, res, seed=0, ctx=None):
    """meshvoxel.clj:25-43 with the reference's unseeded (rand) draws replaced by a counter-based
    uniform of (seed, vertex, draw) -- csrc/rm_volgen.hip scatter_kernel; byte 64, index y*res^2 + z*res + x."""
    c = _ctx(ctx)
    try:
        return c.voxelize_scatter(vertices, res, seed=seed)
    finally:
        if ctx is None:
            c.close()


def make_heatmap(pixels, amp, ctx=None):
    """meshvoxel.clj:73-87 from the image's ARGB pixels (uint32 [res, res], what
    pix/get-pixels returns for the square image the reference loads)."""
    c = _ctx(ctx)
    try:
        return c.make_heatmap_volume(pixels, amp)
    finally:
        if ctx is None:
            c.close()


def make_heatmap_anim(pixels, n, ctx=None):
    """meshvoxel.clj:89-93: n volumes with amp = float(i / (n * 1.33333)); yields them
    instead of saving (the caller chooses the path, vio.save_volume)."""
    for i in range(n):
        yield make_heatmap(pixels, float(np.float32(i / (n * 1.33333))), ctx=ctx)
