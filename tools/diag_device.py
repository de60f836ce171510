"""RM_CONTRACT_GFX950 frames against the reference kernel built for this chip (strict build):
per scene, the pixels whose floats differ and whether the restatement marks them as undefined in the
reference (material index outside the record).  (debugging aid for tests/test_gpu_device_contract.py)"""
import os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import oracle, scenes
from raymarchcl_amd import _native

spec = dict(scenes.SCENES["metal2_fov115"], vol="blobs", w=256, h=192)
sc = scenes.build(spec)
n = sc["n"]
ref, _, _ = oracle.gfx950_render_frame(sc["vox"], sc["opts"], sc["mc"], n, build="strict", tonemap=False)
with _native.Context(0) as ctx:
    ctx.set_volume(sc["vox"], sc["vres"])
    ctx.set_contract("gfx950")
    got, _ = ctx.render_frame(sc["opts"], sc["mc"], n, want_argb=False)
mask = np.zeros(n, np.uint8)
acc = np.zeros(4 * n, np.float32)
for i in range(sc["iter"]):
    oracle.render_image(sc["vox"], sc["mc"][i], sc["opts"][i * 544:(i + 1) * 544], acc, n=n, undefined_mask=mask)
bad = (got.view(np.uint32) != ref.view(np.uint32)).reshape(-1, 4).any(axis=1)
print("differing pixels:", int(bad.sum()), "of", n, "; flagged undefined by the restatement:", int(mask.sum()),
      "; differing AND flagged:", int((bad & (mask != 0)).sum()))
for p in np.nonzero(bad)[0][:8]:
    print(p, got.reshape(-1, 4)[p], ref.reshape(-1, 4)[p], "flag", mask[p])
