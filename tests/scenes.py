"""Deterministic scene definitions shared by the golden-fixture generator and
the parity tests.  A scene = (volume, per-pass option records, per-pass
scatter tables, work-item count).  Everything is derived from seeds so the
GPU box can rebuild the inputs bit-for-bit; the fixtures additionally carry
the inputs themselves (volume + option bytes, and a hash of every table)."""
import hashlib

import numpy as np

import raymarchcl_amd as rm
from raymarchcl_amd import generators as gen
from raymarchcl_amd import structs

_VOL_CACHE = {}


def volume(kind, vres):
    key = (kind, tuple(vres) if not isinstance(vres, int) else vres)
    if key not in _VOL_CACHE:
        if kind == "gyroid":
            v = gen.make_gyroid_volume(vres)
        elif kind == "terrain":
            v = gen.make_terrain(vres)
        elif kind == "blobs":
            v = gen.make_blob_volume(vres)
        elif kind == "gyroid-crop":  # non-cubic: crop a 64^3 gyroid to (64, 40, 48), keep z >= 16
            full = gen.make_gyroid_volume(64).reshape(64, 64, 64)
            v = np.ascontiguousarray(full[16:64, 0:40, :]).reshape(-1)
        elif kind == "empty":
            rx, ry, rz = (vres,) * 3 if isinstance(vres, int) else vres
            v = np.zeros(rx * ry * rz, dtype=np.uint8)
        elif kind == "solid":
            rx, ry, rz = (vres,) * 3 if isinstance(vres, int) else vres
            v = np.full(rx * ry * rz, 200, dtype=np.uint8)
        else:
            raise KeyError(kind)
        _VOL_CACHE[key] = v
    return _VOL_CACHE[key]


# name -> dict(vol kind, vres, width, height, iter, n (None = w*h), render-option kwargs)
SCENES = {
    # BASELINE config 1 geometry at reduced image size (the full 256x256 frame is pinned by hash)
    "c1_orange": dict(vol="gyroid", vres=64, w=64, h=48, iter=1, mat="orange-stripes",
                      theta=-45, dist=2.25),
    "orange_dof_2spp": dict(vol="gyroid", vres=64, w=64, h=48, iter=2, mat="orange-stripes",
                            theta=-45, dist=2.25, dof=0.025),
    "metal_3spp": dict(vol="gyroid", vres=64, w=64, h=48, iter=3, mat="metal", theta=135, dist=2.25),
    "metal2_fov115": dict(vol="gyroid", vres=64, w=64, h=48, iter=2, mat="metal2", theta=20,
                          dist=2.25, fov=115, eye_y=0.44, targetpos=[0, -0.15, 0]),
    "ao_terrain": dict(vol="terrain", vres=64, w=64, h=48, iter=1, mat="ao", theta=60, dist=2.5),
    "ragged_50x37": dict(vol="gyroid", vres=64, w=50, h=37, iter=2, mat="orange-stripes",
                         theta=-45, dist=2.25, n=50 * 37 - 13),
    "noncubic": dict(vol="gyroid-crop", vres=(64, 40, 48), w=64, h=48, iter=1, mat="metal2",
                     theta=200, dist=2.0),
    "inside_volume": dict(vol="gyroid", vres=64, w=48, h=48, iter=1, mat="metal", theta=10,
                          dist=0.6, eye_y=0.3, targetpos=[0.2, 0.1, 0.0]),
    "blobs_metal": dict(vol="blobs", vres=64, w=64, h=48, iter=1, mat="metal", theta=300, dist=2.4),
    "empty_volume": dict(vol="empty", vres=32, w=32, h=24, iter=1, mat="orange-stripes",
                         theta=-45, dist=2.25),
    "solid_volume": dict(vol="solid", vres=32, w=32, h=24, iter=1, mat="metal2", theta=-45,
                         dist=2.25),
}


def build(name_or_spec, mc_seed=1000):
    """-> dict(vox, vres, opts (bytes, iter*544), mc (float32 [iter, 65536]), n, w, h, iter)"""
    sp = dict(SCENES[name_or_spec]) if isinstance(name_or_spec, str) else dict(name_or_spec)
    vres = sp.pop("vres")
    vox = volume(sp.pop("vol"), vres)
    w, h, it = sp.pop("w"), sp.pop("h"), sp.pop("iter")
    n = sp.pop("n", None) or w * h
    theta, dist, eye_y = sp.pop("theta", -45), sp.pop("dist", 2.25), sp.pop("eye_y", 0.35)
    sp.setdefault("targetpos", [0, -0.4, 0])
    vres3 = [vres] * 3 if isinstance(vres, int) else list(vres)
    opts = b"".join(
        structs.encode_bytes(rm.render_options(width=w, height=h, vres=vres3, t=i * 0.333, iter=it,
                                               eyepos=rm.compute_eyepos(theta, dist, eye_y), **sp))
        for i in range(it))
    mc = np.stack([gen.generate_scatter_offsets(0x4000, seed=mc_seed + i) for i in range(it)])
    return dict(vox=vox, vres=tuple(vres3), opts=opts, mc=mc, n=n, w=w, h=h, iter=it)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
