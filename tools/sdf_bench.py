#!/usr/bin/env python3
"""QUALITY MODE (not reference-equivalent) on the bench geometry: 256^3 float distance field,
1280x720, 16 spp + DOF.  Prints the kernel time of a frame; optionally writes a PNG.

    python tools/sdf_bench.py [--res 256] [--kind gyroid|torus] [--png gpurun_out/sdf.png]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--kind", default="gyroid")
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--mat", default="orange-stripes")
    ap.add_argument("--png", default=None)
    args = ap.parse_args()

    import raymarchcl_amd as rm
    from raymarchcl_amd import _native, generators as gen, structs

    w, h, it = args.width, args.height, args.spp
    sdf = gen.make_sdf_volume(args.res, args.kind)
    opts = b"".join(structs.encode_bytes(rm.render_options(
        width=w, height=h, vres=[args.res] * 3, t=i * 0.333, iter=it, eyepos=rm.compute_eyepos(-45, 2.25, 0.35),
        targetpos=[0, -0.4, 0], mat=args.mat, dof=0.025)) for i in range(it))
    mc = np.stack([gen.generate_scatter_offsets(0x4000, seed=1000 + i) for i in range(it)])
    with _native.Context(0, contract="cpu") as ctx:
        ctx.set_sdf_volume(sdf, (args.res,) * 3)
        ctx.render_sdf_frame(opts, mc, w * h, want_pixels=False)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            _px, argb = ctx.render_sdf_frame(opts, mc, w * h, want_pixels=False)
            best = min(best, time.perf_counter() - t0)
            ms, launches = ctx.last_frame_timing()
        print(f"quality mode {args.kind} {args.res}^3 {w}x{h}x{it}: render kernel {ms:.2f} ms "
              f"({w * h * it / ms / 1e3:.0f} Mrays/s), host call {best * 1e3:.1f} ms")
    if args.png:
        from raymarchcl_amd import core

        core.save_png(argb, w, h, args.png)
        print("wrote", args.png)


if __name__ == "__main__":
    main()
