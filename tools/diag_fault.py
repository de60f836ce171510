"""Two frames that expose the allocation-dependent fault of the accelerated frame kernel
(DESIGN.md 4c), for A/B builds selected with RAYMARCH_LIB:
  A  device contract, 25 passes as 16 + 9 per wavefront (MULTI instantiation) vs the reference build
  B  GPU-cast mode (asm form when built with -DRM_F2U_GPU_ASM=1), 1 pass, vs the oracle"""
import os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import oracle, scenes
from raymarchcl_amd import _native

def diff(a, b):
    return int((a.view(np.uint32) != b.view(np.uint32)).reshape(-1, 4).any(axis=1).sum())

os.environ["RAYMARCH_PASS_PACK"] = "4"
sa = scenes.build(dict(vol="gyroid", vres=64, w=56, h=40, iter=25, mat="metal", theta=-30, dist=2.2, dof=0.02), mc_seed=500)
sb = scenes.build(dict(scenes.SCENES["c1_orange"], w=64, h=48))
want_a, _, _ = oracle.gfx950_render_frame(sa["vox"], sa["opts"], sa["mc"], sa["n"], build="strict")
with oracle.seed_cast("gpu"):
    want_b, _ = oracle.render_frame(sb["vox"], sb["opts"], sb["mc"], sb["n"])
with oracle.seed_cast("x86"):
    want_c, _ = oracle.render_frame(sa["vox"], sa["opts"], sa["mc"], sa["n"])
with _native.Context(0) as ctx:
    ctx.set_volume(sa["vox"], sa["vres"])
    ctx.set_contract("gfx950")
    a = diff(ctx.render_frame(sa["opts"], sa["mc"], sa["n"])[0], want_a)
    ctx.set_contract("cpu")
    c = diff(ctx.render_frame(sa["opts"], sa["mc"], sa["n"])[0], want_c)
    ctx.set_volume(sb["vox"], sb["vres"])
    ctx.set_seed_cast("gpu")
    b = diff(ctx.render_frame(sb["opts"], sb["mc"], sb["n"])[0], want_b)
print(f"{os.environ.get('RAYMARCH_LIB', 'product'):<40} A device/MULTI: {a:5d} of {sa['n']}   B gpu-cast: {b:5d} of {sb['n']}   "
      f"(x86 contract, same 25-pass frame: {c})")
