"""bindings/c/test_render.c: the reference's `test-render` (core.clj:154-179) as a plain C program over the C ABI
(include/raymarch_hip.h + libraymarch_hip.so, nothing else).

not gpu: it compiles warning-free against the product header, links against the product library, and on a box
         without a device ends with the library's message and a non-zero status (no CPU fallback).
gpu:     its frame -- built with the C ABI's own parameter layer (rm_render_options, rm_compute_eyepos,
         rm_make_scatter_table, rm_make_gyroid_host) -- equals the frame of the Python host layer
         (core.test_render) word for word, from the host gyroid and from a .vox file."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "bindings", "c", "test_render.c")


def _build(tmp_path):
    from raymarchcl_amd import _native

    _native.build()
    libdir = os.path.dirname(_native.LIB_PATH)
    libname = re.sub(r"^lib|\.so$", "", os.path.basename(_native.LIB_PATH))
    exe = tmp_path / "test_render"
    subprocess.check_call(["gcc", "-std=c11", "-O2", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           SRC, "-o", str(exe), "-L" + libdir, "-l" + libname, "-Wl,-rpath," + libdir, "-lm"])
    return str(exe)


def _fnv1a64(words):
    h = 1469598103934665603
    for b in np.ascontiguousarray(words, dtype="<u4").tobytes():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def test_c_client_builds_and_fails_loudly_without_a_device(tmp_path, native):
    exe = _build(tmp_path)
    # only C-ABI functions the header declares and the library exports
    und = subprocess.check_output(["nm", "--undefined-only", exe], text=True)
    used = re.findall(r" U (rm_\w+)", und)
    assert len(used) >= 9 and all(u in native.EXPORTS for u in used), used
    bad = subprocess.run([exe], capture_output=True, text=True)
    assert bad.returncode == 1 and "usage" in bad.stderr
    if native.device_count() > 0:
        pytest.skip("a GPU is present: the rest is tests/test_c_client.py::test_c_client_frame_equals_the_python_host_layer")
    r = subprocess.run([exe, "32", "24", "1", "32", "ao", str(tmp_path / "o.ppm")], capture_output=True, text=True)
    assert r.returncode == 2 and "rm_create failed" in r.stderr and "no CPU fallback" in r.stderr
    assert not (tmp_path / "o.ppm").exists()


@pytest.mark.gpu
@pytest.mark.parametrize("width,height,iters,vres,mat,from_file", [(64, 48, 2, 64, "orange-stripes", False),
                                                                     (96, 40, 3, 32, "metal", True)])
def test_c_client_frame_equals_the_python_host_layer(tmp_path, native, width, height, iters, vres, mat, from_file):
    from raymarchcl_amd import core, generators, vio

    exe = _build(tmp_path)
    vox = generators.make_gyroid_volume(vres)
    out = tmp_path / "frame.ppm"
    cmd = [exe, str(width), str(height), str(iters), str(vres), mat, str(out)]
    if from_file:
        path = tmp_path / "g.vox"
        vio.save_volume(str(path), vres, vox)
        cmd += [str(path), "1234"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    argb = core.test_render(width=width, height=height, iter=iters, vres=vres, mat=mat, voxels=vox, out_path=None,
                            mc_seed=1234 if from_file else 1000)
    m = re.search(r"argb fnv1a64 ([0-9a-f]{16})", r.stdout)
    assert m and int(m.group(1), 16) == _fnv1a64(argb), r.stdout
    # the exported image: the RGB bytes of the same words
    data = open(out, "rb").read()
    head = b"P6\n%d %d\n255\n" % (width, height)
    assert data.startswith(head)
    rgb = np.frombuffer(data[len(head):], dtype=np.uint8).reshape(height, width, 3)
    assert np.array_equal(rgb, core.argb_to_rgb8(argb, width, height))
