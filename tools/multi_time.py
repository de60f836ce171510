import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import bench
from raymarchcl_amd import _native
wl = bench.WORKLOADS["c2"]
vox, vres, opts, mc = bench.build_inputs(wl)
n = wl["w"] * wl["h"]
for ranks in (1, 2, 8):
    with _native.Context([0] * ranks if ranks > 1 else 0) as ctx:
        ctx.set_volume(vox, vres)
        ctx.render_frame(opts, mc, n)
        t = time.perf_counter()
        for _ in range(5):
            ctx.render_frame(opts, mc, n)
        print(f"rm_render_frame, {ranks} rank(s) on one GPU, host buffers: {(time.perf_counter() - t) / 5 * 1e3:.2f} ms per frame")
