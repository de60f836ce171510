"""CPU: the restatement of the QUALITY MODE (oracle/rm_restate.c sdf_*; not the reference's
algorithm) behaves like a sphere tracer over a distance field -- analytic checks on a sphere,
and a pinned digest so that the checker itself cannot drift unnoticed."""
import hashlib

import numpy as np

import raymarchcl_amd as rm
from raymarchcl_amd import generators as gen
from raymarchcl_amd import structs


def _frame(oracle_mod, sdf, vres, w=48, h=36, it=1, mat="ao", eye=(0.0, 0.0, 2.5), target=(0, 0, 0), **over):
    recs = []
    for i in range(it):
        o = rm.render_options(width=w, height=h, vres=list(vres), t=i * 0.333, iter=it, eyepos=list(eye),
                              targetpos=list(target), mat=mat, dof=0.0)
        o.update(over)
        recs.append(structs.encode_bytes(o))
    mc = np.stack([gen.generate_scatter_offsets(0x4000, seed=9 + i) for i in range(it)])
    return oracle_mod.render_sdf_frame(sdf, b"".join(recs), mc, w * h)


def _sphere(res, r, cy=0.0):
    z, y, x = np.meshgrid(*(((np.arange(res) + 0.5) / res * 2 - 1),) * 3, indexing="ij")
    return (np.sqrt(x * x + (y - cy) ** 2 + z * z) - r).astype(np.float32)


def test_sphere_silhouette_and_shading(oracle_mod):
    w, h = 48, 36
    px, _ = _frame(oracle_mod, _sphere(48, 0.5), (48,) * 3, w, h, groundY=50.0)  # ground far below
    bg, _ = _frame(oracle_mod, np.full((48,) * 3, 10.0, np.float32), (48,) * 3, w, h, groundY=50.0)
    img, back = px.reshape(h, w, 4)[..., :3], bg.reshape(h, w, 4)[..., :3]
    assert np.isfinite(img).all()
    hit = np.abs(img - back).sum(-1) > 1e-3                  # where the sphere changed the picture
    # silhouette: radius 0.5 seen from 2.5 -> half-angle asin(0.2) = 11.5 deg; fov 90 deg over the width
    # maps tan(angle) linearly: tan(11.5 deg) / tan(45 deg) * w/2 = 0.204 * 24 = 4.9 px radius (+- jitter)
    row = hit[h // 2]
    assert 8 <= row.sum() <= 13 and row[w // 2 - 3:w // 2 + 3].all()
    col = hit[:, w // 2]
    assert 8 <= col.sum() <= 13
    assert abs(int(row[: w // 2].sum()) - int(row[w // 2:].sum())) <= 1   # centred
    assert not hit[0].any() and not hit[:, 0].any()


def test_empty_field_is_ground_and_sky(oracle_mod):
    far = np.full((16, 16, 16), 10.0, np.float32)            # nothing anywhere near
    px, _ = _frame(oracle_mod, far, (16,) * 3, eye=(0, 0.5, 2.5), target=(0, -0.4, 0))
    assert np.isfinite(px).all() and len(np.unique(px.reshape(-1, 4)[:, 0])) > 10


def test_restatement_digest_is_pinned(oracle_mod):
    sdf = gen.make_sdf_volume(32, "torus")
    px, argb = _frame(oracle_mod, sdf, (32,) * 3, 40, 30, it=2, mat="metal",
                      eye=rm.compute_eyepos(-45, 2.25, 0.6), target=(0, -0.4, 0))
    digest = hashlib.sha256(px.tobytes() + argb.tobytes()).hexdigest()
    assert digest == "b422258a38f874a7801c7a8ce706b40a0368db73cd0fe1890816589181dded90", digest
