#!/usr/bin/env python3
"""Debug helper: render one frame of a bench workload with the RM_WORK_STATS build
(hipcc <flags of _native.HIPCC_FLAGS> -DRM_WORK_STATS -o raymarchcl_amd/libraymarch_hip_stats.so) and print what
render_frame_kernel executed per sample (marches, turns, filtered turns, walks,
dist8 fetches, samples advanced, AO loops).

    python tools/work_stats.py [c2] [--clock]

--clock uses the -DRM_PHASE_CLOCK build (libraymarch_hip_clock.so: nothing but a shader-clock
read around the phases of shade_wave) and prints the wave time per phase."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raymarchcl_amd import _native
clock = "--clock" in sys.argv
args = [a for a in sys.argv[1:] if a != "--clock"]
_native.LIB_PATH = os.path.join(_native.HERE, "libraymarch_hip_clock.so" if clock else "libraymarch_hip_stats.so")
import bench
wl = bench.WORKLOADS[args[0] if args else "c2"]
vox, vres, opts, mc = bench.build_inputs(wl)
n = wl["w"] * wl["h"]
with _native.Context(0) as ctx:
    ctx.set_volume(vox, vres)
    ctx.render_frame(opts, mc, n, want_pixels=False, want_argb=True)
