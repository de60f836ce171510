#!/usr/bin/env python3
"""How evenly does the interleaved tile partition split a frame?

Runs, on ONE GPU, the share every rank of a world of W would render
(rm_frame_device with tile_first=r, tile_stride=W) and reports the slowest
share next to frame_time/W.  No collective is involved: this isolates the
compute side of the multi-GPU path (partition balance, launch tails) from the
gather, which only an 8-GPU node can time.

    python tools/part_timing.py [--workload c2] [--worlds 1,2,4,8] [--reps 10]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--frames-in-flight", type=int, default=1)
    args = ap.parse_args()

    import torch

    import bench
    from raymarchcl_amd import multigpu

    wl = bench.WORKLOADS[args.workload]
    vox, vres, opts, mc = bench.build_inputs(wl)
    n, width = wl["w"] * wl["h"], wl["w"]
    dev = torch.device("cuda", 0)
    base = None
    for world in [int(w) for w in args.worlds.split(",")]:
        times = []
        for rank in range(world):
            fr = multigpu.FrameRenderer(vox, vres, opts, mc, n, width, rank=rank, world=world,
                                        device=dev, want_pixels=True, want_argb=False,
                                        frames_in_flight=args.frames_in_flight)

            tiles = [torch.zeros(fr.tpp * 256, dtype=torch.float32, device=dev) for _ in fr.slots]

            def share():
                k = fr.frame % len(fr.slots)
                slot = fr.slots[k]
                fr.frame += 1
                with torch.cuda.stream(slot.stream):
                    slot.ctx.frame_device(fr.d_opts.data_ptr(), fr.d_mc.data_ptr(), fr.iters, fr.n,
                                          fr.width, tiles[k].data_ptr(), rank, world)

            for _ in range(len(fr.slots)):
                share()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(args.reps):
                share()
            torch.cuda.synchronize(dev)
            times.append((time.perf_counter() - t0) / args.reps * 1e3)
            fr.close()
        worst = max(times)
        if base is None:
            base = worst * world
        print(f"world {world}: slowest share {worst:.3f} ms, mean {sum(times) / world:.3f} ms, "
              f"ideal {base / world:.3f} ms, compute-side speed-up {base / worst:.2f}x "
              f"(shares: {' '.join(f'{t:.3f}' for t in times)})", flush=True)


if __name__ == "__main__":
    main()
