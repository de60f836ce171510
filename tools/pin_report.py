#!/usr/bin/env python3
"""How much of the parity claim rests on the stand-in for the OpenCL built-in library?

The parity oracle is the UNMODIFIED reference kernel (renderer.cl) compiled for x86-64 and
linked against oracle/ref_shim.cpp, whose exp / exp2 / pow are deterministic double-precision
evaluations (oracle/cl_scalar.h).  A CPU OpenCL runtime would most likely call the host libm.
This tool renders the same scenes with three builds of the reference kernel

    A  oracle build            (contract off, deterministic exp/exp2/pow)   <- the oracle
    B  libm build              (contract off, libm expf/exp2f/powf)
    C  FMA-contracted build    (what -cl-mad-enable legally allows)

and reports BASELINE.json's metric for B against A: fraction of pixel channels within 1e-4
relative, maximum and percentiles of the relative difference -- over all pixels and over the
pixels that are STABLE under legal re-rounding (|A - C| <= 1e-4 relative: the others flip a
hit/miss decision somewhere and no two conforming OpenCL builds agree on them).

Build container only (needs /root/reference through oracle/_ref).  Output: profiles/archive_r02.txt (FILE r02_pin_report.txt)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def rel(a, b):
    a = a.reshape(-1, 4)[:, :3].astype(np.float64)
    b = b.reshape(-1, 4)[:, :3].astype(np.float64)
    return np.abs(a - b) / np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-6)


def render(oracle, sc, which):
    n = sc["n"]
    px = np.zeros(4 * n, np.float32)
    for i in range(sc["iter"]):
        oracle.ref_render_image_mt(sc["vox"], sc["mc"][i].copy(), sc["opts"][i * 544:(i + 1) * 544], px,
                                   os.cpu_count() or 1, n=n, fma=which)
    return px


def line(name, r):
    q = np.percentile(r, [50, 90, 99, 99.9])
    return (f"  {name:<34} within 1e-4: {100.0 * (r <= 1e-4).mean():8.4f} %   max {r.max():.3e}   "
            f"p50 {q[0]:.1e}  p90 {q[1]:.1e}  p99 {q[2]:.1e}  p99.9 {q[3]:.1e}")


def main():
    import oracle
    import scenes

    oracle.build(ref=True)
    cases = [
        ("BASELINE config 1: 64^3 gyroid 256x256 1 spp :orange-stripes", dict(scenes.SCENES["c1_orange"], w=256, h=256)),
        ("64^3 gyroid 128x96 3 spp :metal (3 bounces)", dict(scenes.SCENES["metal_3spp"], w=128, h=96)),
        ("64^3 gyroid 128x96 2 spp :metal2 fov 115", dict(scenes.SCENES["metal2_fov115"], w=128, h=96)),
        ("config-2 geometry: 256^3 gyroid 320x180 2 spp + DOF :orange-stripes",
         dict(vol="gyroid", vres=256, w=320, h=180, iter=2, mat="orange-stripes", theta=-45, dist=2.25, dof=0.025)),
        ("64^3 blobs 128x96 1 spp :metal", dict(scenes.SCENES["blobs_metal"], w=128, h=96)),
    ]
    out = [__doc__.split("Build container only")[0].strip(), ""]
    for title, spec in cases:
        sc = scenes.build(spec)
        a, b, c = render(oracle, sc, False), render(oracle, sc, "libm"), render(oracle, sc, True)
        rab, rac = rel(a, b), rel(a, c)
        stable = (rac <= 1e-4).all(axis=1)
        bits = float((a.view(np.uint32) == b.view(np.uint32)).reshape(-1, 4)[:, :3].all(axis=1).mean())
        out.append(f"{title}  ({sc['n']} pixels)")
        out.append(line("libm build vs oracle, all pixels", rab.max(axis=1)))
        out.append(line("libm build vs oracle, stable pixels", rab.max(axis=1)[stable]))
        out.append(f"  {'':<34} bit-identical pixels: {100.0 * bits:.4f} %   stable under FMA contraction: "
                   f"{100.0 * stable.mean():.4f} % of pixels")
        out.append(line("FMA build vs oracle, all pixels", rac.max(axis=1)))
        out.append("")
    text = "\n".join(out)
    print(text)
    with open(os.path.join(ROOT, "profiles", "r02_pin_report.txt"), "w") as f:
        f.write(text + "\n")


if __name__ == "__main__":
    main()
