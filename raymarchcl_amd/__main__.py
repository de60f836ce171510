"""Command line front end for the render path.

    python -m raymarchcl_amd gen-gyroid --res 256 --out gyroid-256.vox
    python -m raymarchcl_amd gen-terrain --res 256 --out terrain-256.vox
    python -m raymarchcl_amd voxelize --stl bunny.stl --res 512 --ks 1 --out bunny-512.vox
    python -m raymarchcl_amd render --vox gyroid-256.vox --mat metal --width 1280 --height 720 \\
        --iter 16 --theta 135 --dist 2.25 --out frame.png

`render` mirrors the reference's test-render (core.clj:154-179): same defaults,
same parameter names.  Needs the HIP library and an MI355X (no CPU fallback).
"""
import argparse
import sys
import time


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m raymarchcl_amd")
    sub = ap.add_subparsers(dest="cmd", required=True)
    g = sub.add_parser("gen-gyroid", help="write the gyroid benchmark volume as a .vox file")
    g.add_argument("--res", type=int, default=256)
    g.add_argument("--out", required=True)
    g.add_argument("--cpu", action="store_true", help="generate with numpy instead of the device kernel")
    t = sub.add_parser("gen-terrain", help="write gen/make-terrain (generators.clj:44-60) as a .vox file")
    t.add_argument("--res", type=int, default=256)
    t.add_argument("--out", required=True)
    v = sub.add_parser("voxelize", help="STL mesh -> .vox by vertex splatting (meshvoxel.clj voxelize / voxelize-ks)")
    v.add_argument("--stl", required=True)
    v.add_argument("--res", type=int, default=256)
    v.add_argument("--ks", type=int, default=-1, help="cube half-size around every vertex; -1 = the vertex's own cell")
    v.add_argument("--out", required=True)
    r = sub.add_parser("render", help="render one frame of a .vox volume to a PNG (test-render)")
    r.add_argument("--vox", required=True)
    r.add_argument("--out", default="foo.png")
    r.add_argument("--width", type=int, default=640)
    r.add_argument("--height", type=int, default=360)
    r.add_argument("--iter", type=int, default=1)
    r.add_argument("--mat", default="metal")
    r.add_argument("--theta", type=float, default=135)
    r.add_argument("--dist", type=float, default=2.25)
    r.add_argument("--dof", type=float, default=None)
    r.add_argument("--fov", type=float, default=None)
    r.add_argument("--gamma", type=float, default=None)
    r.add_argument("--seed", type=int, default=1000, help="seed of the scatter tables")
    r.add_argument("--device", type=int, default=0)
    r.add_argument("--contract", default="gfx950-default", choices=["cpu", "gfx950-default", "gfx950-strict", "gfx950"],
                   help="whose results the kernels reproduce, bit for bit: gfx950-default (default) = the reference kernel as "
                        "ROCm's OpenCL compiler builds it for this GPU with no options (within 1e-4 of the reference's own "
                        "-cl-fast-relaxed-math build on ~all pixels); gfx950-strict (= gfx950) = the same built with "
                        "-ffp-contract=off and correctly rounded divide/sqrt; cpu = an OpenCL CPU device on x86-64 "
                        "(include/raymarch_hip.h rm_set_contract)")
    args = ap.parse_args(argv)

    from . import core, generators, vio

    if args.cmd == "gen-gyroid":
        t0 = time.time()
        if args.cpu:
            vox = generators.make_gyroid_volume(args.res)
        else:
            from . import _native

            with _native.Context(0) as ctx:
                vox = ctx.make_gyroid_volume(args.res)
        vio.save_volume(args.out, args.res, vox)
        print(f"{args.out}: {args.res}^3, {int((vox > 0).sum())} filled voxels, {time.time() - t0:.2f} s")
        return 0
    if args.cmd in ("gen-terrain", "voxelize"):
        from . import _native, meshvoxel

        t0 = time.time()
        with _native.Context(0) as ctx:
            if args.cmd == "gen-terrain":
                vox = ctx.make_terrain_volume(args.res)
            else:
                vox = ctx.voxelize_vertices(meshvoxel.load_mesh(args.stl), args.res, args.ks)
        vio.save_volume(args.out, args.res, vox)
        print(f"{args.out}: {args.res}^3, {int((vox > 0).sum())} filled voxels, {time.time() - t0:.2f} s")
        return 0
    vox, vres = vio.load_volume(args.vox)
    extra = {k: getattr(args, k) for k in ("dof", "fov", "gamma") if getattr(args, k) is not None}
    t0 = time.time()
    core.test_render(width=args.width, height=args.height, iter=args.iter, vres=list(vres), mat=args.mat,
                     theta=args.theta, dist=args.dist, voxels=vox, out_path=args.out, mc_seed=args.seed,
                     device=args.device, contract=args.contract, **extra)
    print(f"{args.out}: {args.width}x{args.height}, {args.iter} spp, {time.time() - t0:.2f} s")
    return 0


if __name__ == "__main__":
    sys.exit(main())
