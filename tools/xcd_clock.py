#!/usr/bin/env python3
"""When does each XCD finish its share of a frame?  (diagnostic build, GPU box)

    python tools/ab_build.py clock=-DRM_XCD_CLOCK          # CPU box: raymarchcl_amd/libraymarch_hip_ab_clock.so
    python tools/xcd_clock.py [--workload c2] [--frames 6]  # GPU box; environment knobs (RAYMARCH_XCD_2D ...) apply

The frame kernel of that build notes, per XCD (HW_REG_XCC_ID), the first start and the last end of its wavefronts on the
100 MHz constant clock and the sum of their lifetimes (rm_kernels.hip xcd_clock_note).  The dispatcher deals workgroup i to XCD
i mod 8 (the tool counts the wavefronts for which that does not hold), so an XCD's share is fixed by the kernel's block
mapping: the spread of the end times is what a better balanced mapping could gain, the spread of the lifetime sums says
whether the shares differ in WORK or in how fast the XCD got through it.
"""
import argparse
import ctypes
import os
import sys

os.environ.setdefault("RAYMARCH_LIB", "libraymarch_hip_ab_clock.so")
os.environ.setdefault("RAYMARCH_SKIP_LINT", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--frames", type=int, default=6)
    args = ap.parse_args()
    import numpy as np
    import torch
    import bench
    from raymarchcl_amd import _native, multigpu

    L = _native.lib()
    fn = L.rm_debug_xcd_clock
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    wl = bench.WORKLOADS[args.workload]
    vox, vres, opts, mc = bench.build_inputs(wl)
    n = wl["w"] * wl["h"]
    dev = torch.device("cuda", 0)
    fr = multigpu.FrameRenderer(vox, vres, opts, mc, n, wl["w"], device=dev, frames_in_flight=1)
    for _ in range(4):
        fr.render()
        torch.cuda.synchronize(dev)
    buf = np.zeros(64, np.uint64)
    print(f"{args.workload}: {wl['desc']}   (times in us from the first wavefront's start; one blocking frame each)")
    for f in range(args.frames):
        assert fn(None, 1) == 0
        fr.render()
        torch.cuda.synchronize(dev)
        assert fn(buf.ctypes.data, 0) == 0
        ms = fr.ctx.last_frame_timing()[0]
        x = [i for i in range(16) if buf[48 + i] & np.uint64(0xffffffff)]
        t0 = min(int(buf[i]) for i in x)
        start = np.array([(int(buf[i]) - t0) / 100.0 for i in x])
        end = np.array([(int(buf[16 + i]) - t0) / 100.0 for i in x])
        life = np.array([int(buf[32 + i]) / 100.0 for i in x])
        waves = np.array([int(buf[48 + i]) & 0xffffffff for i in x])
        stray = sum(int(buf[48 + i]) >> 32 for i in x)
        print(f"frame {f}: kernel {ms:.4f} ms (all launches); XCDs {x}; wavefronts per XCD {waves.min()}..{waves.max()}, {stray} not on XCD block mod 8")
        print("   first start us : " + " ".join(f"{v:8.1f}" for v in start))
        print("   last end us    : " + " ".join(f"{v:8.1f}" for v in end) + f"   spread {end.max() - end.min():.1f} us = {100 * (end.max() - end.min()) / end.max():.2f} % of the frame")
        print("   lifetimes, ms  : " + " ".join(f"{v / 1000:8.1f}" for v in life) + f"   max/mean {life.max() / life.mean():.4f}")
        print("   mean in flight : " + " ".join(f"{l / (e - s):8.1f}" for l, e, s in zip(life, end, start)))


if __name__ == "__main__":
    main()
