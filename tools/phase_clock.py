#!/usr/bin/env python3
"""Shader-clock time of the frame kernel's wavefronts by phase on the bench workload (GPU).

    patch -p1 < tools/wave_stats.patch; python tools/ab_build.py pclock="-DRM_PHASE_CLOCK=1"; git checkout raymarchcl_amd/csrc
    RAYMARCH_LIB=libraymarch_hip_ab_pclock.so python tools/phase_clock.py [--workload c2]

The -DRM_PHASE_CLOCK=1 build reads s_memtime at the (wave-uniform) phase boundaries of sample_colour_wave /
lighting_wave and adds the differences to one counter per phase: the shares are of the wavefronts' RESIDENT
time (issue + stalls + waits), summed over all wavefronts of one frame.
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

NAMES = ["sample + camera", "primary march", "reflection marches", "AO phases", "shadow phases (incl. pair selection)",
         "lighting arithmetic", "rest (materials, atmosphere, exchange)"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--contract", default="gfx950")
    a = ap.parse_args()
    import torch

    from raymarchcl_amd import _native, multigpu

    wl = bench.WORKLOADS[a.workload]
    vox, vres, opts, mc = bench.build_inputs(wl)
    n, width = wl["w"] * wl["h"], wl["w"]
    dev = torch.device("cuda", 0)
    fr = multigpu.FrameRenderer(vox, vres, opts, mc, n, width, device=dev, frames_in_flight=1, contract=a.contract)
    L = ctypes.CDLL(_native.LIB_PATH)
    buf = (ctypes.c_ulonglong * 8)()
    fr.render()
    torch.cuda.synchronize(dev)
    assert L.rm_debug_phase_clock(buf, 1) == 0
    fr.render()
    torch.cuda.synchronize(dev)
    ms, launches = fr.ctx.last_frame_timing()
    assert L.rm_debug_phase_clock(buf, 0) == 0
    v = [int(x) for x in buf][:7]
    tot = sum(v)
    print(f"# {wl['desc']}: one frame, {ms:.2f} ms with the clock reads in; shares of the wavefronts' resident time")
    for name, x in zip(NAMES, v):
        print(f"{name:42s} {100.0 * x / tot:5.1f} %")
    fr.close()


if __name__ == "__main__":
    main()
