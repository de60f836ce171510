#!/usr/bin/env python3
"""Randomised check (GPU box) that a frame does not depend on the dispatch order of the frame kernel: random image sizes
(widths that hold 8 .. 64 tiles per stripe or none, odd tile-row counts, ragged right / bottom edges), pass counts, tile
partitions and order switches (RAYMARCH_XCD_2D / _XCD_ROWS / _ROW_ORDER / _ROW_BAND / _THIN), every frame against the same frame
in the plain block order (which tests/ check against the oracle).

    python tools/fuzz_order.py [--seconds 240] [--seed 1]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("RAYMARCH_XCD_ROWS", "RAYMARCH_XCD_2D", "RAYMARCH_ROW_ORDER", "RAYMARCH_ROW_BAND", "RAYMARCH_THIN", "RAYMARCH_PASS_PACK")


def render(spec, env):
    """one context per (frame, environment): the library reads its switches when a context is created"""
    import numpy as np
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import scenes
    from raymarchcl_amd import _native
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)
    sc = scenes.build(dict(spec))
    with _native.Context(0) as ctx:
        ctx.set_volume(sc["vox"], sc["vres"])
        px, argb = ctx.render_frame(sc["opts"], sc["mc"], sc["n"])
    return px.view(np.uint32).copy(), argb.copy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import numpy as np
    rng = np.random.default_rng(args.seed)
    t0, cases, frames, bad = time.time(), 0, 0, 0
    while time.time() - t0 < args.seconds:
        w = int(rng.choice([64 * int(rng.integers(1, 33)), 8 * int(rng.integers(1, 260)), int(rng.integers(9, 2100))]))
        h = int(rng.choice([8 * int(rng.integers(1, 14)), int(rng.integers(1, 110))]))
        spp = int(rng.choice([1, 2, 3, 4, 8, 16, 25]))
        if w * h * spp > 3_000_000:
            continue
        spec = dict(vol=str(rng.choice(["gyroid", "blobs"])), vres=int(rng.choice([32, 64])), w=w, h=h, iter=spp,
                    mat=str(rng.choice(["orange-stripes", "metal"])), theta=float(rng.uniform(-180, 180)), dist=float(rng.uniform(1.2, 3.0)),
                    dof=float(rng.choice([0.0, 0.025])))
        want = render(spec, {"RAYMARCH_XCD_ROWS": "0", "RAYMARCH_THIN": "0"})
        cases += 1
        for _ in range(5):
            env = {}
            if rng.random() < 0.8:
                env["RAYMARCH_XCD_2D"] = str(int(rng.integers(0, 9)))
            if rng.random() < 0.4:
                env["RAYMARCH_ROW_ORDER"] = str(rng.choice(["asc", "desc", "band"]))
            if rng.random() < 0.2:
                lo = float(rng.uniform(0, 0.8))
                env["RAYMARCH_ROW_BAND"] = f"{lo:.2f},{min(1.0, lo + float(rng.uniform(0.05, 0.6))):.2f}"
            if rng.random() < 0.3:
                env["RAYMARCH_THIN"] = "0"
            got = render(spec, env)
            frames += 1
            if not (np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])):
                bad += 1
                print("MISMATCH", spec, env, int((got[0] != want[0]).sum()), flush=True)
    print(f"{cases} random frames x 5 random orders = {frames} renders against the plain block order (seed {args.seed}), {bad} differing")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
