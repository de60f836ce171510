#!/usr/bin/env python3
"""Wave-level / lane-level event counts of the frame kernel on the bench workload (GPU).

    patch -p1 < tools/wave_stats.patch                  # the instrumentation is not in the product sources
    python tools/ab_build.py stats="-DRM_STATS=1"; git checkout raymarchcl_amd/csrc
    RAYMARCH_LIB=libraymarch_hip_ab_stats.so python tools/wave_stats.py [--workload c2]

Renders ONE frame with the -DRM_STATS=1 build (rm_shade.hpp: global atomics, very slow) and prints,
per context (primary march / reflection march / AO probe / shadow march / repeated cut turn), how
many times each loop body ran per WAVEFRONT and per LANE -- the ratio is the lane utilisation of
that body, the wave-level count times the body's instruction count its share of the VALU stream.
"""
import argparse
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

EV = ["EST_W", "EST_L", "BOX_L", "WALK_W", "WALK_L", "TRIP_W", "TRIP_L", "GO_L", "J_L", "JMAX_W", "ADDIT_W",
      "HIT_W", "HIT_L", "ROUND_W", "ROUND_L", "FILT_W", "FILT_L", "MARCH_W", "MARCH_L", "PHASE_W", "TASK_L",
      "PROUND_W", "J1_L", "J3_L", "JMAX8_W", "JMAX16_W"]
CTX = ["primary", "reflect", "ao", "shadow", "repeat"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--contract", default="gfx950")
    a = ap.parse_args()
    import torch

    from raymarchcl_amd import _native, multigpu

    wl = bench.WORKLOADS[a.workload]
    vox, vres, opts, mc = bench.build_inputs(wl)
    n, width, spp = wl["w"] * wl["h"], wl["w"], wl["spp"]
    dev = torch.device("cuda", 0)
    fr = multigpu.FrameRenderer(vox, vres, opts, mc, n, width, device=dev, frames_in_flight=1, contract=a.contract)
    L = ctypes.CDLL(_native.LIB_PATH)
    buf = (ctypes.c_ulonglong * 256)()
    L.rm_debug_stats(buf, 256, 1)
    fr.render()
    torch.cuda.synchronize(dev)
    assert L.rm_debug_stats(buf, 256, 0) == 0
    st = np.array(list(buf), dtype=np.float64).reshape(-1)[:len(EV) * 5].reshape(len(EV), 5)
    samples = n * spp
    waves = samples / 64
    print(f"# {wl['desc']}: {samples} samples = {waves:.0f} wavefronts; counts per SAMPLE (lane-level, _L) and per WAVEFRONT (_W)")
    print(f"{'event':10s} " + " ".join(f"{c:>12s}" for c in CTX) + f" {'total':>12s}")
    for i, e in enumerate(EV):
        d = samples if e.endswith("_L") else waves
        row = st[i] / d
        print(f"{e:10s} " + " ".join(f"{v:12.3f}" for v in row) + f" {row.sum():12.3f}")
    def g(e):
        return st[EV.index(e)]
    with np.errstate(divide="ignore", invalid="ignore"):
        print("# lane utilisation of each body = lanes / (64 * wave-level executions)")
        for name, l, w in (("estimate set-up", "EST_L", "EST_W"), ("walk (per call)", "WALK_L", "WALK_W"),
                           ("walk trip", "TRIP_L", "TRIP_W"), ("march round", "ROUND_L", "ROUND_W"),
                           ("filtered turn", "FILT_L", "FILT_W"), ("hit evaluation", "HIT_L", "HIT_W")):
            u = g(l) / (64 * g(w))
            print(f"{name:18s} " + " ".join(f"{v:12.3f}" for v in u) + f" {g(l).sum() / (64 * g(w).sum()):12.3f}")
        print("# per walk trip: samples advanced per going lane, wave maximum, add-loop iterations (4 samples each)")
        print("j mean/lane       " + " ".join(f"{v:12.3f}" for v in g("J_L") / g("GO_L")))
        print("j max/wave trip   " + " ".join(f"{v:12.3f}" for v in g("JMAX_W") / g("TRIP_W")))
        print("add iters/trip    " + " ".join(f"{v:12.3f}" for v in g("ADDIT_W") / g("TRIP_W")))
        print("P(jmax>8)/trip    " + " ".join(f"{v:12.3f}" for v in g("JMAX8_W") / g("TRIP_W")))
        print("P(jmax>16)/trip   " + " ".join(f"{v:12.3f}" for v in g("JMAX16_W") / g("TRIP_W")))
        print("trips/walk (wave) " + " ".join(f"{v:12.3f}" for v in g("TRIP_W") / g("WALK_W")))
        print("trips/walk (lane) " + " ".join(f"{v:12.3f}" for v in g("TRIP_L") / g("WALK_L")))
    fr.close()


if __name__ == "__main__":
    main()
