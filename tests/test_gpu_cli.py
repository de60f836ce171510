"""SURVEY 8(f) n1 end to end on the GPU: a volume written in the reference's .vox format
(io.clj:9-17) -> `python -m raymarchcl_amd render --vox ...` (the reference's test-render,
core.clj:154-179: load-volume, render-options, the pipeline, PNG export) -> the decoded PNG equals
the oracle's ARGB frame for the same inputs, pixel for pixel."""
import os
import subprocess
import sys

import numpy as np
import pytest

import raymarchcl_amd as rm
import scenes
from raymarchcl_amd import generators as gen
from raymarchcl_amd import structs, vio

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("contract", ["cpu", "gfx950-strict", "gfx950-default"])
def test_vox_file_to_png_equals_the_oracle(tmp_path, oracle_mod, contract):
    from PIL import Image

    build = contract.split("-")[-1]
    if contract != "cpu" and not oracle_mod.have_gfx950_ref(build):
        pytest.skip(f"oracle/_ref/renderer_gfx950_{build}.hsaco not built")
    res, w, h, it = 64, 96, 64, 2
    vox = scenes.volume("gyroid", res)
    vio.save_volume(str(tmp_path / "g.vox"), res, vox)
    assert os.path.getsize(tmp_path / "g.vox") == 18 + res ** 3  # io.clj: 5 + 3*4 + 1 header bytes
    out = tmp_path / "frame.png"
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-m", "raymarchcl_amd", "render", "--vox", str(tmp_path / "g.vox"), "--mat",
                        "orange-stripes", "--width", str(w), "--height", str(h), "--iter", str(it), "--theta", "-45",
                        "--dist", "2.25", "--dof", "0.025", "--seed", "4321", "--contract", contract, "--out", str(out)],
                       cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.asarray(Image.open(out).convert("RGB"))
    assert got.shape == (h, w, 3)
    # the same frame from the same inputs through the checker
    opts = b"".join(structs.encode_bytes(rm.render_options(
        width=w, height=h, vres=[res] * 3, t=i * 0.333, iter=it, eyepos=rm.compute_eyepos(-45, 2.25, 0.35),
        targetpos=[0, -0.4, 0], mat="orange-stripes", dof=0.025)) for i in range(it))
    mc = np.stack([gen.generate_scatter_offsets(0x4000, seed=4321 + i) for i in range(it)])
    if contract == "cpu":
        _px, argb = oracle_mod.render_frame(vox, opts, mc, w * h)
    else:
        _px, argb, _ms = oracle_mod.gfx950_render_frame(vox, opts, mc, w * h, build=build)
    want = np.stack([(argb >> 16) & 255, (argb >> 8) & 255, argb & 255], axis=-1).astype(np.uint8).reshape(h, w, 3)
    assert np.array_equal(got, want)
    assert len(np.unique(got.reshape(-1, 3), axis=0)) > 200  # a real image
