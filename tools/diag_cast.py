"""Localise the wrong pixels of a frame-kernel build by switching phases off through the option
record (no recompilation): which phase must be present for the GPU-cast frame kernel to differ
from the oracle?  (debugging aid; RAYMARCH_LIB selects the build)"""
import os, struct, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import oracle, scenes
from raymarchcl_amd import _native

def patch(opts, **kw):
    off = dict(reflectIter=(240, "<i"), numLights=(285, "<B"), aoIter=(216, "<i"), dof=(264, "<f"),
               maxIter=(196, "<i"), shadowIter=(236, "<i"), time=(188, "<f"), lightScatter=(248, "<f"),
               flareAmp=(276, "<f"), minLightAtt=(252, "<f"), fogPow=(272, "<f"))
    b = bytearray(opts)
    for k, v in kw.items():
        o, f = off[k]
        for rec in range(len(b) // 544):
            struct.pack_into(f, b, rec * 544 + o, v)
    return bytes(b)

def diff(a, b):
    return int((a.view(np.uint32) != b.view(np.uint32)).reshape(-1, 4).any(axis=1).sum())

base = scenes.build(dict(scenes.SCENES["c1_orange"], w=64, h=48))
n = base["n"]
cases = [("as is", {}), ("no bounces", dict(reflectIter=0)), ("no lights", dict(numLights=0)),
         ("no AO probes", dict(aoIter=-1)), ("no bounces, no lights", dict(reflectIter=0, numLights=0)),
         ("no bounces, no AO", dict(reflectIter=0, aoIter=-1)), ("no lights, no AO", dict(numLights=0, aoIter=-1)),
         ("nothing but the primary march", dict(reflectIter=0, numLights=0, aoIter=-1)),
         ("time 1.0", dict(time=1.0)), ("maxIter 0", dict(maxIter=0)),
         ("1 light", dict(numLights=1)), ("lightScatter 0", dict(lightScatter=0.0)), ("flareAmp 0", dict(flareAmp=0.0)),
         ("shadowIter 0", dict(shadowIter=0)), ("minLightAtt 1e9 (no light passes)", dict(minLightAtt=1e9)),
         ("minLightAtt 1e9, flareAmp 0", dict(minLightAtt=1e9, flareAmp=0.0)),
         ("no bounces, flareAmp 0", dict(reflectIter=0, flareAmp=0.0)),
         ("no bounces, minLightAtt 1e9", dict(reflectIter=0, minLightAtt=1e9)),
         ("maxIter 0, flareAmp 0", dict(maxIter=0, flareAmp=0.0))]
with _native.Context(0) as ctx:
    ctx.set_volume(base["vox"], base["vres"])
    for name, kw in cases:
        opts = patch(base["opts"], **kw)
        res = []
        for mode in ("gpu", "x86"):
            with oracle.seed_cast(mode):
                want, _ = oracle.render_frame(base["vox"], opts, base["mc"], n)
            ctx.set_seed_cast(mode)
            got, _ = ctx.render_frame(opts, base["mc"], n)
            res.append(diff(got, want))
        print(f"{name:<34} frame kernel vs oracle: gpu-cast {res[0]:5d}  x86-cast {res[1]:5d}  of {n}")
