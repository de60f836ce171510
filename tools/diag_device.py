"""RM_CONTRACT_GFX950 frames against the reference kernel built for this chip (strict build):
per scene, the number of pixels whose floats differ, through the frame kernel (accelerated and
plain) and the single-pass kernels.  (debugging aid for tests/test_gpu_device_contract.py)"""
import os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import oracle, scenes
from raymarchcl_amd import _native

def diff(a, b):
    return int((a.view(np.uint32) != b.view(np.uint32)).reshape(-1, 4).any(axis=1).sum())

names = sys.argv[1:] or list(scenes.SCENES)
for name in names:
    sc = scenes.build(name)
    n = sc["n"]
    ref, ref_argb, _ = oracle.gfx950_render_frame(sc["vox"], sc["opts"], sc["mc"], n, build="strict")
    with _native.Context(0) as ctx:
        ctx.set_volume(sc["vox"], sc["vres"])
        ctx.set_contract("gfx950")
        got, argb = ctx.render_frame(sc["opts"], sc["mc"], n)
        px = np.zeros(4 * n, np.float32)
        for i in range(sc["iter"]):
            ctx.render_image(np.ascontiguousarray(sc["mc"][i]), sc["opts"][i * 544:(i + 1) * 544], px, n=n)
    r = np.abs(got.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1e-6)
    print(f"{name:<18} frame kernel: {diff(got, ref):5d} of {n} pixels differ (argb {int((argb != ref_argb).sum())}), "
          f"single-pass kernels: {diff(px, ref):5d};  max rel {r.max():.2e}", flush=True)
