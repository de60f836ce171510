"""The committed recordings of the reference kernel built for gfx950 (tests/golden/gfx950_{strict,default,fast}/,
made on the GPU by tests/golden/make_golden_gfx950.py) are well formed and belong to the inputs the
GPU tests rebuild from seeds.  CPU only: the comparison of the product against them is in
tests/test_gpu_device_contract.py."""
import json
import os

import numpy as np
import pytest

import gfx950_pin as pin
import scenes


@pytest.mark.parametrize("build", pin.RECORDED)
@pytest.mark.parametrize("name", list(scenes.SCENES))
def test_scene_recordings_belong_to_the_scenes(name, build):
    sc = scenes.build(name)
    z = np.load(os.path.join(pin.fixed_dir(build), name + ".npz"))
    assert str(z["inputs"]) == pin.input_digest(sc["vox"], sc["opts"], sc["mc"], sc["n"])
    px, argb = z["pixels"], z["argb"]
    assert px.dtype == np.float32 and px.size == 4 * sc["n"] and argb.dtype == np.uint32 and argb.size == sc["n"]
    p = px.reshape(-1, 4)
    assert np.isfinite(p).all() and (p[:, 3] == 1.0).all() and (argb >> 24 == 0xff).all()


@pytest.mark.parametrize("build", pin.BUILDS)
def test_recordings_differ_from_the_cpu_contract(oracle_mod, build):
    """The two contracts are different functions (seed casts, fma): a recording that equalled the CPU
    oracle would have been made with the wrong checker."""
    sc = scenes.build("orange_dof_2spp")
    z = np.load(os.path.join(pin.fixed_dir(build), "orange_dof_2spp.npz"))
    want_cpu, _ = oracle_mod.render_frame(sc["vox"], sc["opts"], sc["mc"], sc["n"])
    diff = (z["pixels"].view(np.uint32) != want_cpu.view(np.uint32)).reshape(-1, 4).any(axis=1).mean()
    assert 0.01 < diff < 0.999  # (default build: the contracted camera ray moves the last bit of most pixels)
    # ... but the same picture: most pixels within 1e-4
    a, b = z["pixels"].reshape(-1, 4)[:, :3].astype(np.float64), want_cpu.reshape(-1, 4)[:, :3].astype(np.float64)
    rel = (np.abs(a - b) / np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-6)).max(axis=1)
    assert (rel <= 1e-4).mean() > 0.6


@pytest.mark.parametrize("build", pin.RECORDED)
def test_digest_entries_and_samples(build):
    d = json.load(open(os.path.join(pin.fixed_dir(build), "digests.json")))
    s = np.load(os.path.join(pin.fixed_dir(build), "digest_samples.npz"))
    assert set(d) == set(s.files) == {"pass_packed_8", "pass_packed_16", "pass_packed_12", "pass_packed_25", "c2", "c3", "c4", "c5"}
    for k, e in d.items():
        assert len(e["pixels_sha"]) == 64 and len(e["argb_sha"]) == 64 and len(e["inputs"]) == 64
        assert s[k].dtype == np.uint32 and s[k].size == 4 * len(range(0, e["n"], pin.SAMPLE_STRIDE))
    import bench

    wl = bench.WORKLOADS["c2"]
    vox, vres, opts, mc = bench.build_inputs(wl)
    assert d["c2"]["inputs"] == pin.input_digest(vox, opts, mc, wl["w"] * wl["h"])


def test_the_two_builds_were_recorded_from_different_code_objects():
    """strict and default are different functions (contraction, division): identical recordings would mean one
    directory was filled from the wrong build.  Same inputs, most pixels close."""
    a = np.load(os.path.join(pin.fixed_dir("strict"), "orange_dof_2spp.npz"))
    b = np.load(os.path.join(pin.fixed_dir("default"), "orange_dof_2spp.npz"))
    assert str(a["inputs"]) == str(b["inputs"])
    diff = (a["pixels"].view(np.uint32) != b["pixels"].view(np.uint32)).reshape(-1, 4).any(axis=1).mean()
    assert 0.01 < diff < 0.999


def test_the_fast_recording_is_a_third_build_within_1e4_of_default():
    """`fast` (the reference's own options, core.clj:128) is recorded as the yardstick of BASELINE's 1e-4 metric: a
    different code object (its bits differ from `default` on some pixels) that agrees with `default` -- the build the
    library's default contract reproduces bit for bit -- within 1e-4 on the fixture and on the sampled pixels of every
    BASELINE configuration (the GPU tests compare the PRODUCT against it: tests/test_gpu_pin_gfx950.py)."""
    a = np.load(os.path.join(pin.fixed_dir("fast"), "orange_dof_2spp.npz"))
    b = np.load(os.path.join(pin.fixed_dir("default"), "orange_dof_2spp.npz"))
    assert str(a["inputs"]) == str(b["inputs"])
    diff = (a["pixels"].view(np.uint32) != b["pixels"].view(np.uint32)).reshape(-1, 4).any(axis=1).mean()
    assert 0.005 < diff < 0.999
    assert (pin.rel_err(a["pixels"], b["pixels"]) <= 1e-4).mean() >= 0.995
    fa = np.load(os.path.join(pin.fixed_dir("fast"), "digest_samples.npz"))
    de = np.load(os.path.join(pin.fixed_dir("default"), "digest_samples.npz"))
    for cfg in ("c2", "c3", "c4", "c5"):
        r = pin.rel_err(fa[cfg].view(np.float32), de[cfg].view(np.float32))
        assert r.size >= 900 and (r <= 1e-4).mean() >= 0.999, (cfg, float((r <= 1e-4).mean()))
