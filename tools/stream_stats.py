#!/usr/bin/env python3
"""Debug helper: render one frame of a bench workload with the RM_WORK_STATS build
(libraymarch_hip_stats.so: hipcc ... -DRM_WORK_STATS) and print what each stream
kernel executed (rays, outer iterations, filtered slab tests, voxel walks, lookups)."""
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
