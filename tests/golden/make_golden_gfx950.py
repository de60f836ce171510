#!/usr/bin/env python3
"""Record the outputs of the reference kernel BUILT FOR gfx950 (strict, default and fast builds) -- run on a GPU box.

The checkers of the two device arithmetic contracts are oracle/_ref/renderer_gfx950_{strict,default}.hsaco: the
unmodified /root/reference/resources/renderer.cl compiled by oracle/Makefile (`make -C oracle ref_gfx950`:
clang -x cl -target amdgcn-amd-amdhsa -mcpu=gfx950, `strict` with -ffp-contract=off
-cl-fp32-correctly-rounded-divide-sqrt, `default` with no options; linked by the clang driver against ROCm's own
OpenCL library).  `fast` (-cl-fast-relaxed-math -cl-mad-enable, the reference's own options, core.clj:128) is recorded the
same way: not a bit-exact checker but the yardstick of BASELINE's 1e-4 metric (oracle/pin.py FastReference).  Those files are git-ignored; this script runs them on the GPU exactly as core.clj:76-97 sequences
the kernels (oracle/ref_gfx950_runner.cpp) and stores DATA ONLY, per build, under tests/golden/gfx950_<build>/:

  <scene>.npz    every scene of tests/scenes.py and config 1 at full size: float32 accumulator after all
                 passes, ARGB words, sha256 of the inputs (volume, records, scatter tables, n)
  digests.json   the large frames (BASELINE configs 2-5, the pass-packed frames of
                 tests/test_gpu_device_contract.py): sha256 of accumulator and ARGB buffer;
  digest_samples.npz  every 997th pixel's bit patterns of those frames

    gpurun -- 'python tests/golden/make_golden_gfx950.py gpurun_out/golden_gfx950 [build ...]'   # then copy
    gpurun_out/golden_gfx950/<build>/ into tests/golden/gfx950_<build>/ (gpurun only brings gpurun_out/ back)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bench  # noqa: E402
import gfx950_pin as pin  # noqa: E402
import oracle  # noqa: E402
import scenes  # noqa: E402

PASS_PACKED = [(8, "3"), (16, "4"), (12, "4"), (25, "4")]  # tests/test_gpu_device_contract.py


def pass_packed_scene(passes):
    return scenes.build(dict(vol="gyroid", vres=64, w=56, h=40, iter=passes, mat="metal", theta=-30, dist=2.2, dof=0.02),
                        mc_seed=500)


def main():
    oracle.build(ref=False)
    for build in (sys.argv[2:] or pin.RECORDED):
        record(build, os.path.join(sys.argv[1], build) if len(sys.argv) > 1 else pin.fixed_dir(build))


def record(build, out):
    os.makedirs(out, exist_ok=True)
    assert oracle.have_gfx950_ref(build), f"oracle/_ref/renderer_gfx950_{build}.hsaco is not here"

    def full(key, vox, opts, mc, n):
        px, argb, ms = oracle.gfx950_render_frame(vox, opts, mc, n, build=build)
        np.savez_compressed(os.path.join(out, key + ".npz"), pixels=px, argb=argb,
                            inputs=np.array(pin.input_digest(vox, opts, mc, n)), n=np.int32(n))
        print(f"{key:18s} n={n:8d} ref {ms:9.2f} ms  sha {pin.sha(px)[:12]}", flush=True)

    for name in scenes.SCENES:
        sc = scenes.build(name)
        full(name, sc["vox"], sc["opts"], sc["mc"], sc["n"])
    wl = bench.WORKLOADS["c1"]
    vox, vres, opts, mc = bench.build_inputs(wl)
    full("c1", vox, opts, mc, wl["w"] * wl["h"])

    digests, samples = {}, {}

    def digest(key, vox, opts, mc, n):
        px, argb, ms = oracle.gfx950_render_frame(vox, opts, mc, n, build=build)
        digests[key] = dict(inputs=pin.input_digest(vox, opts, mc, n), n=int(n), pixels_sha=pin.sha(px),
                            argb_sha=pin.sha(argb))
        samples[key] = px.view(np.uint32).reshape(-1, 4)[::pin.SAMPLE_STRIDE].reshape(-1).copy()
        print(f"{key:18s} n={n:8d} ref {ms:9.2f} ms  sha {digests[key]['pixels_sha'][:12]}", flush=True)

    for passes, _pack in PASS_PACKED:
        sc = pass_packed_scene(passes)
        digest(f"pass_packed_{passes}", sc["vox"], sc["opts"], sc["mc"], sc["n"])
    for cfg in ("c2", "c3", "c4", "c5"):
        wl = bench.WORKLOADS[cfg]
        vox, vres, opts, mc = bench.build_inputs(wl)
        digest(cfg, vox, opts, mc, wl["w"] * wl["h"])
    json.dump(digests, open(os.path.join(out, "digests.json"), "w"), indent=1)
    np.savez_compressed(os.path.join(out, "digest_samples.npz"), **samples)


if __name__ == "__main__":
    main()
