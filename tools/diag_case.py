#!/usr/bin/env python3
"""One dumped fuzz case (tools/fuzz_parity.py --only N --dump f.npz) taken apart on the GPU: which pass
and which part of the shading makes a work-item of the device contract differ from the reference kernel
built for gfx950.  Usage: python tools/diag_case.py f.npz <work-item>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from raymarchcl_amd import _native, structs  # noqa: E402

z = np.load(sys.argv[1])
item = int(sys.argv[2])
vox, vres, mc, n = z["vox"], [int(v) for v in z["vres"]], z["mc"], int(z["n"])
opts = z["opts"].tobytes()
it = len(opts) // 544


def both(rec, table, ctx):
    ref, _, _ = oracle.gfx950_render_frame(vox, rec, table[None], n, build="strict", tonemap=False)
    frame, _ = ctx.render_frame(rec, table[None], n, want_argb=False)
    px = np.zeros(4 * n, np.float32)
    ctx.render_image(table, rec, px, n)
    return ref.reshape(-1, 4), frame.reshape(-1, 4), px.reshape(-1, 4)


def patched(rec, **kw):
    a = np.frombuffer(rec, dtype=structs.TRenderOpts).copy()
    for k, v in kw.items():
        a[k] = v
    return a.tobytes()


with _native.Context(0) as ctx:
    ctx.set_contract("gfx950")
    ctx.set_volume(vox, vres)
    bad_pass = []
    for i in range(it):
        rec = opts[i * 544:(i + 1) * 544]
        ref, frame, px = both(rec, mc[i], ctx)
        nd_f = int((ref.view(np.uint32) != frame.view(np.uint32)).any(axis=1).sum())
        nd_p = int((ref.view(np.uint32) != px.view(np.uint32)).any(axis=1).sum())
        same = np.array_equal(ref[item].view(np.uint32), frame[item].view(np.uint32))
        print(f"pass {i}: frame kernel differs in {nd_f} work-items, pass kernel in {nd_p}; item {item}: ref {ref[item][:3]} "
              f"frame {frame[item][:3]} pass {px[item][:3]}")
        if not same:
            bad_pass.append(i)
    for i in bad_pass:
        rec = opts[i * 544:(i + 1) * 544]
        base = np.frombuffer(rec, dtype=structs.TRenderOpts)[0]
        print(f"pass {i}: aoIter {base['aoIter']} reflectIter {base['reflectIter']} numLights {base['numLights']} "
              f"shadowIter {base['shadowIter']} maxVoxelIter {base['maxVoxelIter']} isoVal {base['isoVal']}")
        for name, kw in (("no lights", dict(numLights=0)), ("no AO (aoIter -1)", dict(aoIter=-1)), ("aoAmp 0", dict(aoAmp=0.0)),
                         ("no bounces", dict(reflectIter=0)), ("no fog", dict(fogPow=0.0)), ("no flares", dict(flareAmp=0.0)),
                         ("shadowIter 0", dict(shadowIter=0)), ("dof 0", dict(dof=0.0)), ("lightScatter 0", dict(lightScatter=0.0))):
            ref, frame, px = both(patched(rec, **kw), mc[i], ctx)
            print(f"   {name:<18} item {item}: ref {ref[item][:3]}  frame {frame[item][:3]}  pass {px[item][:3]}  "
                  f"{'SAME' if np.array_equal(ref[item].view(np.uint32), frame[item].view(np.uint32)) else 'DIFFERENT'}")

    # event counts of the plain algorithm under the SAME contract: does a work-item of this pass index the
    # materials outside the record (undefined in the reference: it reads its private copy out of bounds)?
    for i in bad_pass:
        rec = opts[i * 544:(i + 1) * 544]
        cnt = _native.Counters()
        px = np.zeros(4 * n, np.float32)
        ctx.render_image(mc[i], rec, px, n, counters=cnt)
        print(f"pass {i}, plain algorithm in the device contract: {cnt.as_dict()}")
        print(f"   its value for item {item}: {px.reshape(-1, 4)[item][:3]}")
        ctx.set_contract("cpu")
        cnt2 = _native.Counters()
        ctx.render_image(mc[i], rec, px, n, counters=cnt2)
        print(f"pass {i}, plain algorithm in the CPU-device contract: oob_material {cnt2.oob_material}")
        ctx.set_contract("gfx950")
