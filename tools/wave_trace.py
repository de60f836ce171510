#!/usr/bin/env python3
"""Per-task event traces of the shared phases (AO probes, shadow marches) of sampled wavefronts (GPU).

    patch -p1 < tools/wave_stats.patch; python tools/ab_build.py stats="-DRM_STATS=1"; git checkout raymarchcl_amd/csrc
    RAYMARCH_LIB=libraymarch_hip_ab_stats.so python tools/wave_trace.py --out gpurun_out/trace_c2.npz

One frame of the bench workload with the -DRM_STATS=1 build; every 768th wavefront writes, for each round of
its shared phases, the event string of every lane's task (rm_shade.hpp tr_emit): tools/wave_sim.py replays
them under other schedules.
"""
import argparse
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--contract", default="gfx950")
    ap.add_argument("--out", default="gpurun_out/trace.npz")
    a = ap.parse_args()
    import torch

    from raymarchcl_amd import _native, multigpu

    wl = bench.WORKLOADS[a.workload]
    vox, vres, opts, mc = bench.build_inputs(wl)
    n, width = wl["w"] * wl["h"], wl["w"]
    dev = torch.device("cuda", 0)
    fr = multigpu.FrameRenderer(vox, vres, opts, mc, n, width, device=dev, frames_in_flight=1, contract=a.contract)
    L = ctypes.CDLL(_native.LIB_PATH)
    L.rm_debug_trace.restype = ctypes.c_longlong
    L.rm_debug_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    slots = L.rm_debug_trace(None, None, 1)
    assert slots > 0, slots
    fr.render()
    torch.cuda.synchronize(dev)
    hdr = np.zeros(slots, dtype=np.uint32)
    ev = np.zeros((slots, 64, 128), dtype=np.uint8)
    assert L.rm_debug_trace(hdr.ctypes.data, ev.ctypes.data, 0) == slots
    used = hdr != 0
    print(f"{int(used.sum())} rounds traced in {slots} slots")
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    np.savez_compressed(a.out, hdr=hdr, ev=ev)
    fr.close()


if __name__ == "__main__":
    main()
