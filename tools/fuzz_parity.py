#!/usr/bin/env python3
"""Randomised parity: small frames with random cameras, presets, volumes and record overrides,
GPU (rm_render_frame through the C ABI) against the CPU restatement, bit for bit.

    python tools/fuzz_parity.py [--cases 200] [--seed 1]

Prints every mismatching case with the parameters that reproduce it; exit code 1 if any.
(The oracle is used here exactly as the tests use it: as the checker.)"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--contract", default="cpu", choices=["cpu", "gfx950", "gfx950-strict", "gfx950-default"],
                    help="cpu: kernels in the OpenCL-CPU-device contract against the CPU restatement; gfx950: kernels "
                         "in a device contract against the reference kernel built for gfx950 (strict / default build), on the GPU")
    ap.add_argument("--seconds", type=float, default=0.0, help="stop after this much wall time (0: run all cases)")
    ap.add_argument("--passes", default="1,2,4,3,8,16", help="pass counts the cases draw from (default: the list the "
                    "earlier rounds' seeds were logged with; add 25,32 for runs that fill 32 pass slots per wavefront)")
    ap.add_argument("--ao-max", type=int, default=8, help="largest aoIter the record overrides draw (default 8 = the "
                    "logged seeds' stream; 20 exercises the chunked AO exchange of the 256^3 / 512^3 frame kernels)")
    ap.add_argument("--grids", default="", help="comma-separated cubic gyroid edges to draw the volume from instead "
                    "of the default mix (e.g. 256 or 256,512: the table layouts with the edge compiled in)")
    ap.add_argument("--only", type=int, default=-1, help="replay the random stream but render only this case")
    ap.add_argument("--dump", default="", help="with --only: save the case's inputs and both results to this .npz")
    args = ap.parse_args()

    import time

    import oracle
    import raymarchcl_amd as rm
    import scenes
    from raymarchcl_amd import _native, generators as gen, materials, structs

    rng = np.random.default_rng(args.seed)
    pass_counts = [int(v) for v in args.passes.split(",")]
    vols = [("gyroid", 64), ("terrain", 64), ("blobs", 64), ("gyroid-crop", (64, 40, 48)), ("gyroid", 32),
            ("gyroid", 128), ("gyroid", 256), ("terrain", 128)]
    if args.grids:
        vols = [("gyroid", int(v)) for v in args.grids.split(",")]
    sparse = gen.make_blob_volume(64, radius=(0.01, 0.03))
    mats = sorted(materials.presets)
    bad = skipped = undefined_dev = 0
    ctx = _native.Context(0)
    ctx.set_contract(args.contract)
    ref_build = "default" if args.contract == "gfx950-default" else "strict"  # the reference build a device contract is pinned to
    t_start = time.time()
    done = 0
    for case in range(args.cases):
        if args.seconds and time.time() - t_start > args.seconds:
            break
        done = case + 1
        kind, vres = vols[int(rng.integers(len(vols)))]
        vox = sparse if (kind == "blobs" and rng.random() < 0.5) else scenes.volume(kind, vres)
        vres3 = [vres] * 3 if isinstance(vres, int) else list(vres)
        w, h, it = int(rng.integers(17, 64)), int(rng.integers(9, 48)), int(rng.choice(pass_counts))
        inside = rng.random() < 0.25
        eye = (rng.uniform(-0.9, 0.9, 3) if inside else
               rm.compute_eyepos(rng.uniform(0, 360), rng.uniform(1.2, 3.5), rng.uniform(-0.9, 1.6)))
        base = dict(width=w, height=h, vres=vres3, iter=it, eyepos=[float(v) for v in eye],
                    targetpos=[float(v) for v in rng.uniform(-0.5, 0.5, 3)], mat=str(rng.choice(mats)),
                    fov=float(rng.uniform(40, 120)), dof=float(rng.choice([0.0, 0.001, 0.025, 0.1])))
        over = {}
        if rng.random() < 0.5:
            pool = dict(aoIter=int(rng.integers(0, args.ao_max + 1)), aoStepDist=float(rng.uniform(0.01, 0.3)),
                        aoAmp=float(rng.uniform(0.0, 0.6)), voxelSize=float(rng.uniform(0.001, 0.05)),
                        groundY=float(rng.uniform(0.3, 1.5)), maxVoxelIter=int(rng.integers(8, 300)),
                        lightScatter=float(rng.uniform(0.0, 0.5)), shadowBias=float(rng.uniform(0.01, 0.3)),
                        eps=float(rng.uniform(0.001, 0.03)), maxDist=float(rng.uniform(3, 40)),
                        maxIter=int(rng.integers(4, 160)), shadowIter=int(rng.integers(2, 160)),
                        reflectIter=int(rng.integers(0, 4)), isoVal=int(rng.integers(0, 200)),
                        fogPow=float(rng.uniform(0, 0.2)), numLights=int(rng.integers(1, 3)))
            for k in rng.choice(sorted(pool), size=int(rng.integers(1, 5)), replace=False):
                over[str(k)] = pool[str(k)]
        if rng.random() < 0.3:
            # the box the byte grid occupies: anisotropic scale, shifted / asymmetric clip planes,
            # a march that starts away from the eye (the exact-skip arguments read all of these)
            sc3 = rng.uniform(0.7, 1.4, 3)
            kind_box = int(rng.integers(0, 4))
            if kind_box in (0, 3):
                over.update(voxelBounds=[float(v) for v in sc3], voxelBounds2=[float(2 * v) for v in sc3],
                            invVoxelScale=[float(0.5 / v) for v in sc3],
                            voxelBoundsMin=[float(-0.99 * v) for v in sc3], voxelBoundsMax=[float(0.99 * v) for v in sc3])
            if kind_box in (1, 3):
                lo = rng.uniform(-0.99, -0.3, 3)
                hi = rng.uniform(0.3, 0.99, 3)
                b = over.get("voxelBounds", [1.0, 1.0, 1.0])
                over.update(voxelBoundsMin=[float(l * bb) for l, bb in zip(lo, b)],
                            voxelBoundsMax=[float(hh * bb) for hh, bb in zip(hi, b)])
            if kind_box in (2, 3):
                over["startDist"] = float(rng.uniform(0.0, 1.5))
            if rng.random() < 0.3:
                over["up"] = [float(v) for v in rng.normal(size=3)]
        if rng.random() < 0.15:
            # signs and zeros in the inputs the unlit-pair shortcut of lighting_wave reasons about
            nl = int(rng.integers(1, 5))
            over.update(numLights=nl, minLightAtt=float(rng.choice([0.0, 0.0, 0.05, 0.2])),
                        lightPos=[[float(v) for v in rng.uniform(-2.5, 2.5, 3)] + [0.0] for _ in range(nl)],
                        lightColor=[[float(v) for v in rng.choice([-30.0, -0.0, 0.0, 20.0, 60.0], 3)] + [0.0]
                                    for _ in range(nl)])
            if rng.random() < 0.5:
                over["materials"] = [dict(albedo=[float(v) for v in rng.choice([-0.5, -0.0, 0.0, 0.3, 1.5], 3)] + [1.0],
                                          r0=float(rng.choice([0.0, 0.1, 0.7])), smoothness=float(rng.uniform(0, 1)))
                                     for _ in range(4)]
            if rng.random() < 0.5:
                over["skyColor1"] = [float(v) for v in rng.choice([-1.0, -0.0, 0.0, 1.8], 3)]
        recs = []
        for i in range(it):
            o = rm.render_options(t=i * 0.333, **base)
            o.update(over)
            recs.append(structs.encode_bytes(o))
        opts = b"".join(recs)
        seed = int(rng.integers(1 << 30))
        mc = np.stack([gen.generate_scatter_offsets(0x4000, seed=seed + i) for i in range(it)])
        n = w * h - int(rng.integers(0, 5))
        if args.only >= 0 and case != args.only:
            continue
        want = np.zeros(4 * n, np.float32)
        mask = np.zeros(n, np.uint8)
        for i in range(it):
            oracle.render_image(vox, mc[i], opts[i * 544:(i + 1) * 544], want, n=n, undefined_mask=mask)
        if args.contract != "cpu":
            # The checker is the reference kernel itself.  A work-item whose material index leaves
            # the record (undefined in the reference, renderer.cl:394,418) makes that kernel read its
            # private copy of the record out of bounds -- on this chip a memory access fault that
            # takes the process down -- so frames the restatement marks are not given to it.
            if mask.any():
                skipped += 1
                continue
            want, _, _ = oracle.gfx950_render_frame(vox, opts, mc, n, build=ref_build, tonemap=False)
        ctx.set_volume(vox, vres3)
        px, _ = ctx.render_frame(opts, mc, n)
        ok = np.repeat(mask == 0, 4)
        a, b = px.view(np.uint32)[ok], want.view(np.uint32)[ok]
        nan = np.isnan(want[ok])
        diff = int((a[~nan] != b[~nan]).sum()) + int((~np.isnan(px[ok][nan])).sum())
        if args.only >= 0:
            d4 = (px.view(np.uint32) != want.view(np.uint32)).reshape(-1, 4).any(axis=1) & (mask == 0)
            for i in np.nonzero(d4)[0][:16]:
                print(f"  work-item {i} (x {i % w}, y {i // w}): ours {px.reshape(-1, 4)[i]} reference {want.reshape(-1, 4)[i]}")
            if args.dump:
                np.savez(args.dump, vox=vox, vres=np.array(vres3), opts=np.frombuffer(opts, np.uint8), mc=mc, n=n, w=w,
                         ours=px, want=want, mask=mask)
        if diff and args.contract != "cpu":
            # A work-item may index the materials outside the record under THIS contract's arithmetic only (a
            # bounce that lands elsewhere): the restatement's mask above cannot know.  The plain algorithm in
            # the device contract counts such lookups; per item = count over items 0..i minus count over 0..i-1.
            d4 = (px.view(np.uint32) != want.view(np.uint32)).reshape(-1, 4).any(axis=1) & (mask == 0)
            items = np.nonzero(d4)[0]
            explained = len(items) <= 8
            for i in (items if explained else []):
                oob = 0
                for k in range(it):
                    rec = opts[k * 544:(k + 1) * 544]
                    c1, c0 = _native.Counters(), _native.Counters()
                    scratch = np.zeros(4 * n, np.float32)
                    ctx.render_image(mc[k], rec, scratch, int(i) + 1, counters=c1)
                    if i > 0:
                        ctx.render_image(mc[k], rec, scratch, int(i), counters=c0)
                    oob += c1.oob_material - c0.oob_material
                explained &= oob > 0
            if explained:
                undefined_dev += 1
                diff = 0
        if diff:
            bad += 1
            print(f"MISMATCH case {case}: {diff} floats; volume {kind} {vres}, base {base}, over {over}, mc seed {seed}, n {n}",
                  flush=True)
    ctx.close()
    print(f"{done} cases (seed {args.seed}, {args.contract} contract), {bad} mismatching" +
          (f", {skipped} not run (a work-item with an undefined material index)" if skipped else "") +
          (f", {undefined_dev} with work-items that differ AND index the materials outside the record under the device "
           f"contract's arithmetic only (undefined in the reference)" if undefined_dev else ""))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
