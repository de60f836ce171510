"""QUALITY MODE (not the reference's algorithm): rm_render_sdf_frame against
the CPU restatement of the same algorithm (oracle/rm_restate.c sdf_*), bit for bit -- both
follow one arithmetic contract.  There is no reference to pin either to."""
import numpy as np
import pytest

import raymarchcl_amd as rm
from raymarchcl_amd import generators as gen
from raymarchcl_amd import structs

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("cpu_contract")]  # this module checks against the CPU oracle


def _records(w, h, it, vres, mat, theta, over=None):
    recs = []
    for i in range(it):
        o = rm.render_options(width=w, height=h, vres=list(vres), t=i * 0.333, iter=it,
                              eyepos=rm.compute_eyepos(theta, 2.25, 0.6), targetpos=[0, -0.4, 0], mat=mat)
        o.update(over or {})
        recs.append(structs.encode_bytes(o))
    return b"".join(recs)


@pytest.mark.parametrize("kind,vres,mat,theta,over", [
    ("torus", (48, 48, 48), "metal", -45, None),
    ("torus", (64, 40, 56), "orange-stripes", 30, dict(lightScatter=0.05)),
    ("gyroid", (64, 64, 64), "metal2", 120, None),
    ("torus", (32, 32, 32), "ao", 200, dict(aoIter=3, shadowIter=40)),
    ("torus", (32, 32, 32), "ao", 200, dict(aoIter=9, shadowIter=40)),   # more probes than the exchange area holds: per-lane path
])
def test_sdf_frame_matches_its_restatement(native, oracle_mod, kind, vres, mat, theta, over):
    w, h, it = 56, 40, 2
    sdf = gen.make_sdf_volume(vres, kind)
    assert sdf.shape == (vres[2], vres[1], vres[0])
    opts = _records(w, h, it, vres, mat, theta, over)
    mc = np.stack([gen.generate_scatter_offsets(0x4000, seed=31 + i) for i in range(it)])
    n = w * h
    want, want_argb = oracle_mod.render_sdf_frame(sdf, opts, mc, n)
    with native.Context(0) as ctx:
        ctx.set_sdf_volume(sdf, vres)
        px, argb = ctx.render_sdf_frame(opts, mc, n)
    bad = int((px.view(np.uint32) != want.view(np.uint32)).sum())
    assert bad == 0, f"{bad} of {px.size} floats differ"
    assert np.array_equal(argb, want_argb)
    assert len(np.unique(argb)) > 50  # there is an object in view


def test_sdf_api_errors(native):
    with native.Context(0) as ctx:
        opts = _records(16, 16, 1, (16, 16, 16), "metal", 0)
        mc = gen.generate_scatter_offsets(0x4000, seed=1)
        with pytest.raises(Exception):
            ctx.render_sdf_frame(opts, mc, 256)            # no field yet
        ctx.set_sdf_volume(np.ones((8, 8, 8), np.float32), (8, 8, 8))
        with pytest.raises(Exception):
            ctx.render_sdf_frame(opts, mc, 256)            # voxelRes mismatch
        with pytest.raises(Exception):
            ctx.set_sdf_volume(np.ones(8, np.float32), (1, 8, 1))
        with pytest.raises(Exception):
            ctx.set_sdf_volume(np.ones(4097 * 4, np.float32), (4097, 2, 2))   # more than 4096 cells on an axis


def test_field_can_be_replaced_and_breakdown_names_the_devices_that_took_part(native, oracle_mod):
    """rm_set_sdf_volume stages the scalar field, builds the float4 faces and releases the staging copy: a second,
    larger field on the same context renders as exactly as the first.  A quality-mode frame on a multi-device context
    runs on the root alone: rm_last_frame_breakdown reports 0 for the other devices instead of an earlier frame's times."""
    import scenes

    w, h, it = 40, 32, 1
    n = w * h
    mc = np.stack([gen.generate_scatter_offsets(0x4000, seed=77)])
    sc = scenes.build("orange_dof_2spp")
    with native.Context([0, 0]) as ctx:
        ctx.set_volume(sc["vox"], sc["vres"])
        ctx.render_frame(sc["opts"], sc["mc"], sc["n"])          # a frame over both devices
        shares, frame_ms = ctx.last_frame_breakdown()
        assert len(shares) == 2 and all(s > 0 for s in shares) and frame_ms > 0
        for vres in ((24, 24, 24), (40, 32, 48)):
            sdf = gen.make_sdf_volume(vres, "torus")
            opts = _records(w, h, it, vres, "metal", -45)
            want, want_argb = oracle_mod.render_sdf_frame(sdf, opts, mc, n)
            ctx.set_sdf_volume(sdf, vres)
            px, argb = ctx.render_sdf_frame(opts, mc, n)
            assert np.array_equal(px.view(np.uint32), want.view(np.uint32)) and np.array_equal(argb, want_argb)
        shares, frame_ms = ctx.last_frame_breakdown()
        assert shares[0] > 0 and shares[1] == 0.0 and frame_ms > 0
