"""Host-side parameter layer against the values the reference's Clojure code
produces (core.clj:28-74, materials.clj, generators.clj, io.clj), hand-derived
in SURVEY.md Appendix B since no JVM is available to run it."""
import math
import struct

import numpy as np
import pytest

import raymarchcl_amd as rm
from raymarchcl_amd import generators as gen
from raymarchcl_amd import materials, structs, vio


def opts_for(**kw):
    base = dict(width=640, height=360, vres=256, t=0.0, iter=4)
    base.update(kw)
    return rm.render_options(**base)


def test_defaults_match_core_clj():
    o = opts_for(mat="orange-stripes")
    assert o["eps"] == 0.005 and o["aoIter"] == 5 and o["aoStepDist"] == 0.05
    assert o["dof"] == 0.001 and o["exposure"] == 3.5 and o["eyePos"] == [2, 0, 2]
    assert o["flareAmp"] == 0.015 and o["fogPow"] == 0.05
    assert o["fov"] == pytest.approx(math.pi / 2)
    assert o["frameBlend"] == 0.25 and o["gamma"] == 1.5 and o["groundY"] == 1.05
    assert o["invAspect"] == 360 / 640 and o["isoVal"] == 32
    assert o["maxDist"] == 30 and o["maxIter"] == 128 and o["maxVoxelIter"] == 192
    assert o["shadowBias"] == 0.1 and o["shadowIter"] == 128 and o["lightScatter"] == 0.2
    assert o["targetPos"] == [0, -0.15, 0]
    assert o["voxelRes"] == [256, 256, 256, 65536] and o["voxelSize"] == 1 / 256
    assert o["voxelBoundsMin"] == [-0.99] * 3 and o["voxelBoundsMax"] == [0.99] * 3


def test_preset_wins_over_defaults_and_call_site():
    o = opts_for(mat="metal")
    assert o["reflectIter"] == 3 and o["aoAmp"] == 0.25 and o["numLights"] == 2
    assert o["lightPos"][0] == [0, 2, 0, 0]
    # unknown material -> the :ao preset (core.clj:74)
    o = opts_for(mat="no-such-material")
    assert o["numLights"] == 1 and o["reflectIter"] == 0 and o["lightColor"] == [[50, 50, 50, 0]]
    # `or` semantics: 0 is truthy in Clojure, None falls back
    assert opts_for(dof=0)["dof"] == 0 and opts_for(dof=None)["dof"] == 0.001


def test_encode_layout_and_zero_padding():
    o = opts_for(mat="ao", eyepos=[1.5, 0.35, -2.0])
    b = structs.encode_bytes(o)
    assert len(b) == 544
    f = lambda off: struct.unpack_from("<f", b, off)[0]
    i = lambda off: struct.unpack_from("<i", b, off)[0]
    assert (f(0), f(4), f(8), f(12)) == (1.5, np.float32(0.35), -2.0, 0.0)
    assert [i(160 + 4 * k) for k in range(4)] == [256, 256, 256, 65536]
    assert (i(176), i(180)) == (640, 360)
    assert f(184) == np.float32(360 / 640) and f(268) == 0.25
    assert b[284] == 32 and b[285] == 1 and b[286:288] == b"\0\0"
    # :ao has ONE light colour; entries 1..3 are zero padded, lightPos keeps 2 defaults
    assert [f(352 + 4 * k) for k in range(8)] == [50, 50, 50, 0, 0, 0, 0, 0]
    assert [f(288 + 4 * k) for k in range(12)] == [-2, 0, -2, 0, 2, 0, 2, 0, 0, 0, 0, 0]
    # materials[3] of :ao = albedo 1,1,1,1 r0 0 smoothness 1 dummy 0,0
    assert [f(416 + 96 + 4 * k) for k in range(8)] == [1, 1, 1, 1, 0, 1, 0, 0]
    rec = structs.decode(b)
    assert rec["maxVoxelIter"] == 192 and rec["mcTableLength"] == 0


def test_presets_table():
    p = materials.presets
    assert set(p) == {"orange-stripes", "metal", "metal2", "ao"}
    assert p["orange-stripes"]["materials"][1]["albedo"] == [4.9, 0.9, 0.05, 1.0]
    assert p["metal2"]["materials"][3]["r0"] == 0.75 and p["metal"]["materials"][2]["r0"] == 0.7
    assert materials.lookup(":metal") is p["metal"]


def test_compute_eyepos_rotate_y():
    e = rm.compute_eyepos(-45, 2.25, 0.35)
    s = 2.25 * math.sin(math.radians(-45))
    assert e == pytest.approx([s, 0.35, 2.25 * math.cos(math.radians(-45))])
    assert rm.compute_eyepos(0, 2.0, 0.1) == pytest.approx([0.0, 0.1, 2.0])
    assert rm.compute_eyepos(90, 2.0, 0.1) == pytest.approx([2.0, 0.1, 0.0], abs=1e-12)


def test_pass_times():
    from raymarchcl_amd import core

    bufs = core.make_render_option_buffer(3, dict(width=8, height=8, vres=16, iter=3, mat="ao"))
    assert [structs.decode(b)["time"] for b in bufs] == [np.float32(0), np.float32(0.333), np.float32(0.666)]
    core.update_render_option_buffer(bufs, dict(width=8, height=8, vres=16, iter=3, mat="ao"))
    assert structs.decode(bufs[1])["time"] == np.float32(0.3333)
    assert structs.decode(bufs[0])["frameBlend"] == np.float32(1 / 3)


def test_scatter_table_distribution():
    t = gen.generate_scatter_offsets(0x4000, seed=5).reshape(-1, 4)
    assert t.dtype == np.float32 and t.shape == (0x4000, 4)
    assert np.allclose(np.linalg.norm(t.astype(np.float64), axis=1), 1.0, atol=2e-7)
    assert abs(float(t.mean())) < 0.01
    assert np.array_equal(t, gen.generate_scatter_offsets(0x4000, seed=5).reshape(-1, 4))
    assert not np.array_equal(t, gen.generate_scatter_offsets(0x4000, seed=6).reshape(-1, 4))


def test_gyroid_volume_statistics():
    v = gen.make_gyroid_volume(64)
    assert v.size == 64 ** 3 and set(np.unique(v)) <= {0, 64, 128, 255}
    assert int((v > 0).sum()) == 19738
    g = v.reshape(64, 64, 64)
    assert not g[:32].any()  # slabs with (z & 63) < 32 stay empty
    assert not (g[:, :, :32] == 128).any() and not (g[:, :, 32:] == 64).any()
    # spot value straight from the formula (generators.clj:18-42)
    x, y, z = 40, 11, 50
    s = 0.01 * 512 / 64
    X, Y, Z = x * s + 0.3875, y * s, z * s
    val = abs(math.cos(X) * math.sin(Z) + math.cos(Y) * math.sin(X) + math.cos(Z) * math.sin(Y)) - 1.0
    want = (64 if (x & 63) < 32 else 128) if abs(0.2 - val) < 0.05 else (255 if val > 0.35 else 0)
    assert g[z, y, x] == want


def test_vox_round_trip_and_header(tmp_path):
    v = gen.make_gyroid_volume(64)
    p = tmp_path / "g.vox"
    vio.save_volume(str(p), 64, v)
    raw = p.read_bytes()
    assert raw[:5] == b"VOXEL" and raw[5:17] == struct.pack(">iii", 64, 64, 64) and raw[17] == 1
    assert len(raw) == 18 + 64 ** 3
    back, res = vio.load_volume(str(p))
    assert res == (64, 64, 64) and np.array_equal(back, v)
    p.write_bytes(raw[:1000])
    with pytest.raises(ValueError):
        vio.load_volume(str(p))
    p.write_bytes(b"NOPE!" + raw[5:])
    with pytest.raises(ValueError):
        vio.load_volume(str(p))
