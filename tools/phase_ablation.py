#!/usr/bin/env python3
"""Where a frame's time goes, by switching phases off through the option record (no
instrumented build): the headline frame (bench.py workload c2) rendered whole, without
reflection bounces, without AO probes, without lights (no shadow marches), and with the
primary march alone.  Frame time = HIP events around the one launch, median of 7."""
import os, struct, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from raymarchcl_amd import multigpu

def patch(opts, **kw):
    off = dict(reflectIter=(240, "<i"), numLights=(285, "<B"), aoIter=(216, "<i"))
    b = bytearray(opts)
    for k, v in kw.items():
        o, f = off[k]
        for rec in range(len(b) // 544):
            struct.pack_into(f, b, rec * 544 + o, v)
    return bytes(b)

import torch
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
vox, vres, opts, mc = bench.build_inputs(wl)
n, w = wl["w"] * wl["h"], wl["w"]
cases = [("whole frame", {}), ("no reflection bounces", dict(reflectIter=0)), ("no AO probes", dict(aoIter=-1)),
         ("no lights (no shadow marches)", dict(numLights=0)), ("no AO, no lights", dict(aoIter=-1, numLights=0)),
         ("primary march only", dict(reflectIter=0, aoIter=-1, numLights=0))]
for name, kw in cases:
    fr = multigpu.FrameRenderer(vox, vres, patch(opts, **kw), mc, n, w, frames_in_flight=1,
                                contract=os.environ.get("RM_CONTRACT", "cpu"))
    ms = []
    for _ in range(9):
        fr.render(); torch.cuda.synchronize()
        ms.append(fr.ctx.last_frame_timing()[0])
    fr.close()
    print(f"{name:<32} {np.median(ms[2:]):7.3f} ms")
