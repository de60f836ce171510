#!/usr/bin/env python3
"""Debug helper: render one frame of a bench workload with the RM_WORK_STATS build
(hipcc <flags of _native.HIPCC_FLAGS> -DRM_WORK_STATS -o raymarchcl_amd/libraymarch_hip_stats.so) and print what
render_samples_kernel executed per sample (marches, turns, filtered turns, walks,
dist8 fetches, samples advanced, AO loops)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raymarchcl_amd import _native
_native.LIB_PATH = os.path.join(_native.HERE, "libraymarch_hip_stats.so")
import bench
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
vox, vres, opts, mc = bench.build_inputs(wl)
n = wl["w"] * wl["h"]
with _native.Context(0) as ctx:
    ctx.set_volume(vox, vres)
    ctx.render_frame(opts, mc, n, want_pixels=False, want_argb=True)
