#!/usr/bin/env python3
"""bench.py -- the reference's headline workload on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one frame of BASELINE.json configs[1]: the 256^3 gyroid byte
volume (generators.clj:27-42), 1280x720, 16 passes (spp) with DOF 0.025,
preset :orange-stripes, camera of the README example -- i.e. the whole
pipeline of core.clj:76-97 (zeroed accumulator, 16 RenderImage passes in
order, TonemapImage) with every input already resident in HBM.  With N > 1
(one rank per GPU: launched by torch.distributed.run, or -- when called as plain
`python bench.py --gpus N` -- re-launched by this script under it) the frame's 8x8 tiles
are interleaved over the ranks and the tile accumulators are gathered on rank
0 over RCCL -- the total work is fixed, so scaling is "strong".

Prints ONE JSON line on rank 0.  `value` = primary rays (= samples) per second
of the whole job with ONE BLOCKING FRAME PER STEP (launch .. result on the device, the way the
reference's loop runs its pipeline, core.clj:203-208; the frame period with several frames in
flight is reported under `pipelined`, never as `value`); `roofline` prices the dominant kernel (render_frame_kernel:
all RenderImage passes of the frame in one launch) against HBM bandwidth using
the ALGORITHMIC bytes of the reference algorithm for that launch;
`cpu_baseline` is the CPU restatement of the reference kernel (oracle/) timed
on this box's host cores on a bounded sample -- reported, not the target.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

WORKLOADS = {
    # name: volume, vres, width, height, spp, render-option extras
    "c2": dict(desc="256^3 gyroid, 1280x720, 16 spp + DOF 0.025, :orange-stripes", vol="gyroid",
               vres=256, w=1280, h=720, spp=16, mat="orange-stripes", dof=0.025),
    "c1": dict(desc="64^3 gyroid, 256x256, 1 spp, :orange-stripes", vol="gyroid", vres=64, w=256,
               h=256, spp=1, mat="orange-stripes"),
    "c3": dict(desc="512^3 procedural blob volume (bunny stand-in, ~8 % fill), 1920x1080, 16 spp, :metal",
               vol="blobs", vres=512, w=1920, h=1080, spp=16, mat="metal"),
    "c4": dict(desc="256^3 gyroid, 3840x2160, 64 spp + DOF 0.025, :orange-stripes", vol="gyroid",
               vres=256, w=3840, h=2160, spp=64, mat="orange-stripes", dof=0.025),
    "c5": dict(desc="1024^3 gyroid (dragon stand-in, generated on the device), 1920x1080, 25 spp, :metal",
               vol="gyroid-gpu", vres=1024, w=1920, h=1080, spp=25, mat="metal"),
}


def build_inputs(wl):
    import raymarchcl_amd as rm
    from raymarchcl_amd import generators as gen
    from raymarchcl_amd import structs

    vres = wl["vres"]
    if wl["vol"] == "gyroid":
        vox = gen.make_gyroid_volume(vres)
    elif wl["vol"] == "gyroid-gpu":
        from raymarchcl_amd import _native

        with _native.Context(int(os.environ.get("LOCAL_RANK", "0"))) as gctx:
            vox = gctx.make_gyroid_volume(vres)
    else:
        import torch

        # (numpy takes two minutes for 512^3 x 160 blobs; same formula through torch on the GPU)
        vox = gen.make_blob_volume(vres, radius=(0.01, 0.03),
                                   device=f"cuda:{os.environ.get('LOCAL_RANK', '0')}" if torch.cuda.is_available() else None)
    extra = {k: wl[k] for k in ("dof",) if k in wl}
    opts = b"".join(
        structs.encode_bytes(rm.render_options(
            width=wl["w"], height=wl["h"], vres=[vres] * 3, t=i * 0.333, iter=wl["spp"],
            eyepos=rm.compute_eyepos(-45, 2.25, 0.35), targetpos=[0, -0.4, 0], mat=wl["mat"], **extra))
        for i in range(wl["spp"]))
    mc = np.stack([gen.generate_scatter_offsets(0x4000, seed=1000 + i) for i in range(wl["spp"])])
    return vox, (vres,) * 3, opts, mc


def source_digest(rocm_path=None):
    """Digest of everything the hot kernel is compiled from -- the TRACKED kernel sources + the compiler flags with the
    location of the ROCm tree normalised out: a property of the sources, not of the box (a git-ignored scratch file in
    csrc/ or another ROCM_PATH does not change it).  Identifies the build a PMC measurement belongs to."""
    import hashlib

    from raymarchcl_amd import _native

    root = rocm_path if rocm_path is not None else _native.ROCM_PATH
    flags = [f.replace(root, "$ROCM") if root else f for f in _native.hipcc_flags(rocm_path)]
    h = hashlib.sha256(" ".join(flags).encode())
    for name in _native.kernel_source_files():
        h.update(name.encode())
        h.update(open(os.path.join(_native.CSRC, name), "rb").read())
    return h.hexdigest()[:16]


def load_traffic(path):
    """HBM bytes per launch of the frame kernel from a rocprofv3 --pmc run (tools/pmc_traffic.sh
    writes the JSON together with the digest of the sources it measured).  Reported only when
    that digest is the one of the library being benchmarked: a number measured on another build
    says nothing about this one.  -> (bytes or None, note)"""
    if not (path and os.path.exists(path)):
        return None, "no PMC measurement on file"
    try:
        j = json.load(open(path))
    except Exception:
        return None, "unreadable PMC file"
    if j.get("source_digest") != source_digest():
        return None, f"PMC file {os.path.basename(path)} was measured on another build ({j.get('source_digest')})"
    return j.get("hbm_bytes_per_launch"), f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this build ({os.path.basename(path)})"


def animated_volume_leg(local_rank, vox, vres, opts, mc, n, width, contract, frames=12):
    """A NEW volume every frame (the reference's heat-map animation, meshvoxel.clj:85-89 -> core.clj:181-213): the frame
    period when the tables of volume k+1 are built on the library's own stream beside frame k (rm_stage_volume_device /
    rm_commit_staged_volume), next to the serial form (rm_set_volume_device, tables inside the frame).  Blocking frames:
    the caller waits for frame k before it commits volume k+1.  Inputs resident in HBM; never `value`."""
    import torch

    from raymarchcl_amd import _native

    dev = torch.device("cuda", local_rank)
    res = vres[0]
    base = torch.from_numpy(np.ascontiguousarray(vox)).to(dev).reshape(res, res, res)
    vols = [torch.roll(base, shifts=5 * k, dims=k % 3).contiguous().reshape(-1) for k in range(3)]  # three different volumes
    iters = len(opts) // 544
    d_opts = torch.from_numpy(np.frombuffer(opts, dtype=np.uint8).copy()).to(dev)
    d_mc = torch.from_numpy(np.ascontiguousarray(mc, dtype=np.float32)).to(dev)
    d_px = torch.zeros(4 * n, dtype=torch.float32, device=dev)
    d_argb = torch.zeros(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize(dev)
    iso = 32
    with _native.Context(local_rank, contract=contract) as ctx:
        def frame():
            ctx.frame_device_full(d_opts.data_ptr(), d_mc.data_ptr(), iters, n, width, d_px.data_ptr(), d_argb.data_ptr())

        # serial: the volume changes, the next frame builds its tables
        ctx.set_volume_device(vols[0].data_ptr(), vres)
        ctx.check_device_opts(d_opts.data_ptr(), iters, n, width)
        frame()
        ctx.synchronize()
        t0 = time.perf_counter()
        for k in range(frames):
            ctx.set_volume_device(vols[(k + 1) % 3].data_ptr(), vres)
            ctx.check_device_opts(d_opts.data_ptr(), iters, n, width)  # (builds the tables)
            frame()
            ctx.synchronize()
        serial_ms = (time.perf_counter() - t0) / frames * 1e3
        build_ms = ctx.last_table_build_ms()
        # staged: tables of volume k+1 beside frame k
        ctx.stage_volume_device(vols[0].data_ptr(), vres, iso)
        ctx.commit_staged_volume()
        ctx.check_device_opts(d_opts.data_ptr(), iters, n, width)
        frame()
        ctx.stage_volume_device(vols[1].data_ptr(), vres, iso)
        ctx.synchronize()
        ctx.commit_staged_volume()
        t0 = time.perf_counter()
        for k in range(frames):
            frame()
            ctx.stage_volume_device(vols[(k + 2) % 3].data_ptr(), vres, iso)
            ctx.synchronize()
            ctx.commit_staged_volume()
        staged_ms = (time.perf_counter() - t0) / frames * 1e3
        hist = ctx.frame_timing_history(frames)
        kernel_beside_build = float(np.mean([ms / k for ms, k in hist]))
        staged_build_ms = ctx.last_table_build_ms()
    return {"frame_period_ms_staged": round(staged_ms, 4), "frame_period_ms_serial": round(serial_ms, 4),
            "kernel_ms_beside_the_build": round(kernel_beside_build, 4), "table_build_ms_alone": round(build_ms, 3),
            "table_build_ms_beside_the_frame": round(staged_build_ms, 3), "frames": frames,
            "note": "a different volume every frame, blocking frames, inputs resident in HBM: staged = "
                    "rm_stage_volume_device(volume k+1) right after the launch of frame k, rm_commit_staged_volume after the "
                    "wait; serial = rm_set_volume_device + rm_check_device_opts (tables built before the frame's launch)"}


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-run this command line under
    torch.distributed.run with one rank per GPU (rank 0 prints the JSON line)."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this host driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="the extra `pipelined` leg: successive frames alternate between this many HIP streams "
                         "(default 3 on one GPU, 2 per rank on several -- measured best; 1 = leg skipped).  The "
                         "timed region behind `value` is always ONE BLOCKING FRAME AT A TIME, as the reference's loop "
                         "runs its pipeline (core.clj:203-208)")
    ap.add_argument("--backend", default="ranks", choices=["ranks", "library"],
                    help="ranks: one process per GPU, torch.distributed (RCCL) gather of the tile accumulators; "
                         "library: ONE process, the frame tiled over N devices inside the C library "
                         "(rm_create_multi: peer copies over xGMI, no RCCL) -- what a JNI caller gets")
    ap.add_argument("--contract", default="gfx950-default", choices=["cpu", "gfx950-default", "gfx950-strict", "gfx950"],
                    help="arithmetic contract of the kernels (include/raymarch_hip.h rm_set_contract): gfx950-default "
                         "(the library default) / gfx950-strict = the results of the reference kernel as ROCm's OpenCL "
                         "compiler builds it for this GPU with no options / with -ffp-contract=off and correctly "
                         "rounded divide+sqrt (each checked bit for bit against that build on the GPU; gfx950 = alias "
                         "of gfx950-strict, the name rounds 2-4 used); cpu = the results of an OpenCL CPU device "
                         "(checked against the CPU oracle).  Rounds <= 4 benchmarked the strict contract by default: "
                         "compare round-over-round numbers per contract (`other_contract` carries the others)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lean", action="store_true",
                    help="the timed frames and the roofline only: no host-boundary / animated-volume / other-contract / "
                         "reference-kernel legs, no CPU baseline (profiling runs: rocprofv3 then sees the frame kernel "
                         "of the timed region and nothing that runs beside it)")
    ap.add_argument("--cpu-passes", type=int, default=4, help="passes of the workload the CPU baseline renders")
    ap.add_argument("--traffic", default=os.path.join(ROOT, "profiles", "r06_pmc_traffic.json"))
    args = ap.parse_args()
    if args.lean:
        args.no_cpu_baseline = True
        args.frames_in_flight = 1
    if args.contract == "gfx950":  # the name rounds 2-4 used
        args.contract = "gfx950-strict"

    import torch
    import torch.distributed as dist

    if "WORLD_SIZE" not in os.environ and args.gpus > 1 and args.backend == "ranks":
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.backend == "library":
        sys.exit(library_bench(args))
    if world != args.gpus:
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the render path has no CPU fallback")
    # BENCH_ONE_DEVICE=1 (rehearsal of the N > 1 path on a single-GPU box): every rank on
    # cuda:0 and gloo instead of RCCL, which cannot place two ranks on one device
    rehearsal = os.environ.get("BENCH_ONE_DEVICE", "0") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from raymarchcl_amd import _native, multigpu

    if args.frames_in_flight <= 0:
        # measured: overlapping frames hides the ~0.4 ms tail of a launch (256^3: 3 streams
        # -5 %), but with volumes far beyond the caches concurrent frames evict each other
        # (512^3: +5 %, 1024^3: +7 %)
        big = WORKLOADS[args.workload]["vres"] ** 3 > (64 << 20)
        args.frames_in_flight = 1 if big else (3 if world == 1 else 2)  # (the `pipelined` leg only)

    # (no-op when the library is current; several ranks must not rebuild the same file at once)
    if world == 1 or rank == 0:
        _native.build()
    if world > 1:
        dist.barrier()
    wl = WORKLOADS[args.workload]
    vox, vres, opts, mc = build_inputs(wl)
    n, width, spp = wl["w"] * wl["h"], wl["w"], wl["spp"]
    # `value`: one frame at a time -- the reference's loop runs execute-pipeline and takes its result per frame
    # (core.clj:203-208), so a step is a BLOCKING frame: launch, (gather, resolve,) wait
    fr = multigpu.FrameRenderer(vox, vres, opts, mc, n, width, rank=rank, world=world, device=dev,
                                want_pixels=True, want_argb=True, frames_in_flight=1, contract=args.contract)

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def timed_steps(renderer, blocking):
        """W untimed + K timed steps between barrier + synchronize on both sides -> seconds, max over ranks."""
        for _ in range(args.warmup):
            renderer.render()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            renderer.render()
            if blocking:
                torch.cuda.synchronize(dev)  # this frame's result is there before the next frame starts
        sync_all()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    elapsed = timed_steps(fr, blocking=True)
    # Dominant kernel: render_frame_kernel.  The library records HIP events around the launch(es) of every frame on the
    # stream they run on and keeps the last 32 pairs: read here, AFTER the loop, they are the device times of the very
    # frames `elapsed` timed (the last min(K, 32) of them), so ms_per_step >= kernel_ms holds by construction.
    timed_hist = fr.ctx.frame_timing_history(min(args.steps, 32))
    # the same frames enqueued back to back on several streams (an animation loop that does not wait per frame):
    # reported beside the headline, never `value`
    pipelined = None
    if args.frames_in_flight > 1:
        fp = multigpu.FrameRenderer(vox, vres, opts, mc, n, width, rank=rank, world=world, device=dev, want_pixels=True,
                                    want_argb=True, frames_in_flight=args.frames_in_flight, contract=args.contract)
        p_elapsed = timed_steps(fp, blocking=False)
        fp.close()
        pipelined = {"frames_in_flight": args.frames_in_flight, "ms_per_step": round(p_elapsed / args.steps * 1e3, 4),
                     "value": round(n * spp * args.steps / p_elapsed / 1e6, 3), "unit": "Mrays/s",
                     "note": "frame PERIOD with successive frames in flight on separate HIP streams (the tail of one "
                             "launch overlaps the head of the next); no frame finishes sooner than ms_per_step of the headline"}

    # The serial wall time per frame -- what a caller of the blocking pipeline (core.clj:171) sees -- from a few extra
    # frames with a host clock around each.
    serial = []
    for _ in range(6):
        sync_all()
        fr.frame = 0
        ts = time.perf_counter()
        fr.render()
        sync_all()
        serial.append((time.perf_counter() - ts) * 1e3)
    launches = timed_hist[-1][1]
    kernel_ms = [ms / k for ms, k in timed_hist]
    pass_ms = float(np.mean(kernel_ms))  # average duration of one render-kernel launch over the timed frames
    serial_ms = float(np.median(serial[1:]))

    # N > 1: where a frame's time goes on every rank (events on the rank's stream around its share, the
    # gather and the root's resolve; strictly one frame at a time), and the same frame exchanged as
    # tonemapped ARGB words instead of float4 accumulators (what a caller that wants only the image gets)
    per_rank = argb_exchange = None
    if world > 1:
        rows = []
        for _ in range(6):
            sync_all()
            fr.render(timed=True)
            torch.cuda.synchronize(dev)
            rows.append(fr.last_breakdown())
        mine = [float(v) for v in np.median(np.array(rows[1:]), axis=0)]
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        fa = multigpu.FrameRenderer(vox, vres, opts, mc, n, width, rank=rank, world=world, device=dev,
                                    want_pixels=False, want_argb=True, frames_in_flight=1, contract=args.contract)
        ta = timed_steps(fa, blocking=True)
        arows = []
        for _ in range(4):
            sync_all()
            fa.render(timed=True)
            torch.cuda.synchronize(dev)
            arows.append(fa.last_breakdown())
        amine = [float(v) for v in np.median(np.array(arows[1:]), axis=0)]
        aeveryone = [None] * world
        dist.all_gather_object(aeveryone, amine + [ta])
        fa.close()
        if rank == 0:
            def summary(rows_, k):
                v = [r[k] for r in rows_]
                return {"max": round(max(v), 4), "mean": round(sum(v) / len(v), 4), "by_rank": [round(x, 4) for x in v]}
            per_rank = {"share_render_ms": summary(everyone, 0), "gather_ms": summary(everyone, 1),
                        "resolve_ms_root": round(everyone[0][2], 4),
                        "note": "one frame at a time; gather_ms is the collective as the rank's stream sees it: on the root "
                                "it ends when the slowest rank's tiles have arrived, on the others when theirs are sent"}
            argb_exchange = {"ms_per_step": round(max(r[3] for r in aeveryone) / args.steps * 1e3, 4),
                             "share_render_ms": summary(aeveryone, 0), "gather_ms": summary(aeveryone, 1),
                             "resolve_ms_root": round(aeveryone[0][2], 4),
                             "note": "the same frames with want_pixels=False: every rank tonemaps its tiles, 4 B per pixel "
                                     "are gathered instead of 16, the root un-permutes (rm_frame_device_argb)"}

    out = None
    if rank == 0:
        samples_per_frame = n * spp
        value = samples_per_frame * args.steps / elapsed / 1e6
        # algorithmic bytes of the workload: exact event counts from the counting
        # variant of the kernel (same algorithm, +counters), untimed, all passes
        alg_bytes_frame, c = algorithmic_bytes(vox, vres, opts, mc, n, spp, local_rank)
        # one launch covers spp/launches passes of this rank's tiles
        alg_bytes_launch = alg_bytes_frame / launches / world
        achieved = alg_bytes_launch / (pass_ms * 1e-3) / 1e9
        traffic, traffic_note = (load_traffic(args.traffic) if (world == 1 and args.workload == "c2")
                                 else (None, "measured for the default workload on one GPU only"))
        out = {
            "metric": "Mrays/s (primary rays = pixel samples per second) + ms/frame, " +
                      ("256^3 gyroid 1280x720x16spp" if args.workload == "c2" else wl["desc"]),
            "value": round(value, 3),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "ms_per_frame": round(serial_ms, 4),         # the "ms/frame" of the metric: one blocking frame alone
            "ms_per_frame_serial": round(serial_ms, 4),  # (same number under its round-2 name)
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": wl["desc"], "volume": f"{vres[0]}^3 u8", "resolution": [wl["w"], wl["h"]],
                       "spp": spp, "partition": f"8x8 tiles interleaved over {world} GPU(s)" + ((", gloo gather through the host (BENCH_ONE_DEVICE rehearsal: all ranks on one GPU)" if rehearsal
                                     else ", RCCL gather to rank 0") if world > 1 else ""),
                       "contract": args.contract + {
                           "cpu": " (OpenCL CPU device semantics; bit-exact vs the CPU oracle)",
                           "gfx950-default": " (ROCm OpenCL on this GPU; bit-exact vs the reference kernel built for gfx950 with no "
                                             "options, which is within 1e-4 of the reference's own fast-math build on 100.0000 % of the pixels of every BASELINE configuration c1-c5, "
                                             "profiles/r06_pin_gfx950.txt)",
                           "gfx950-strict": " (ROCm OpenCL on this GPU; bit-exact vs the reference kernel built for gfx950 with "
                                            "-ffp-contract=off and correctly rounded divide/sqrt)"}[args.contract],
                       "frames_in_flight": 1,
                       "overlap": "none: every timed step is one blocking frame (launch .. result), as core.clj:203-208 runs it"},
            "all_rays_per_s_M": round((c["rays"] + c["ao_calls"]) * args.steps / elapsed / 1e6, 2),
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_note,
                "kernel": "render_frame_kernel<accel, 7 waves/SIMD>", "kernel_ms": round(pass_ms, 4),
                "kernel_ms_source": f"HIP events around the launches of the last {len(timed_hist)} of the {args.steps} timed "
                                    "blocking frames (rm_frame_timing_history, read after the loop): mean",
                "launches_per_frame": launches,
                "alg_bytes_per_launch": int(alg_bytes_launch),
                "alg_bytes_per_sample": round(alg_bytes_frame / samples_per_frame, 1),
                "vox_reads_per_sample": round(c["vox_reads"] / samples_per_frame, 1),
                "table_reads_per_sample": round(c["mc_reads"] / samples_per_frame, 2),
            },
        }
        if pipelined:
            out["pipelined"] = pipelined
        if per_rank:
            out["per_rank"] = per_rank
            out["argb_exchange"] = argb_exchange
        if world == 1 and not args.lean:
            # the same frame through the host-buffer boundary (rm_render_frame: tables +
            # records up, float4 accumulator + ARGB back over PCIe) -- never `value`
            hctx = _native.Context(local_rank)
            hctx.set_volume(vox, vres)
            # once per (volume, isoVal), outside every timed region: the tables derived from the
            # volume (dist8, oct8, surf32) -- reported so that nothing is hidden in the set-up
            tp = time.perf_counter()
            hctx.render_frame(opts, mc, n)  # the first frame of a fresh context builds the tables
            first_ms = (time.perf_counter() - tp) * 1e3
            if args.workload == "c2":
                out["animated_volume"] = animated_volume_leg(local_rank, vox, vres, opts, mc, n, width, args.contract)
            out["precompute"] = {"derived_tables_ms": round(hctx.last_table_build_ms(), 3),
                                 "first_frame_ms": round(first_ms, 3),
                                 "note": "dist8 + oct8 + surf32 of the resident volume (device time, HIP events), built "
                                         "once per (volume, isoVal) inside the first frame that needs them; "
                                         "first_frame_ms = that frame through the host-buffer boundary, tables included"}
            th = time.perf_counter()
            for _ in range(3):
                hctx.render_frame(opts, mc, n)
            host_ms = (time.perf_counter() - th) / 3 * 1e3
            # what the reference's own caller does: pageable buffers, only the ARGB image read back (core.clj:91-97)
            pg_mc = np.ascontiguousarray(mc, dtype=np.float32).reshape(-1).copy()
            pg_argb = np.zeros(n, np.uint32)
            hctx.render_frame_into(opts, pg_mc, n, None, pg_argb)
            th = time.perf_counter()
            for _ in range(5):
                hctx.render_frame_into(opts, pg_mc, n, None, pg_argb)
            pageable_argb_ms = (time.perf_counter() - th) / 5 * 1e3
            # the same with long-lived, page-locked caller buffers (what a JNI caller's direct buffers are)
            h_mc = np.ascontiguousarray(mc, dtype=np.float32).reshape(-1)
            h_px, h_argb = np.zeros(4 * n, np.float32), np.zeros(n, np.uint32)
            for a in (h_mc, h_px, h_argb):
                hctx.pin_host_buffer(a)
            hctx.render_frame_into(opts, h_mc, n, h_px, h_argb)
            th = time.perf_counter()
            for _ in range(5):
                hctx.render_frame_into(opts, h_mc, n, h_px, h_argb)
            pinned_ms = (time.perf_counter() - th) / 5 * 1e3
            # ... and only the ARGB image read back: what the reference's pipeline transfers (it reads q-buf, core.clj:91-97)
            hctx.render_frame_into(opts, h_mc, n, None, h_argb)
            th = time.perf_counter()
            for _ in range(5):
                hctx.render_frame_into(opts, h_mc, n, None, h_argb)
            argb_only_ms = (time.perf_counter() - th) / 5 * 1e3
            hctx.close()
            out["host_boundary"] = {"ms_per_frame": round(host_ms, 3),
                                    "Mrays_per_s": round(samples_per_frame / host_ms / 1e3, 2),
                                    "ms_per_frame_argb_only": round(pageable_argb_ms, 3),
                                    "Mrays_per_s_argb_only": round(samples_per_frame / pageable_argb_ms / 1e3, 2),
                                    "ms_per_frame_pinned": round(pinned_ms, 3),
                                    "ms_per_frame_pinned_argb_only": round(argb_only_ms, 3),
                                    "note": "rm_render_frame with host buffers the caller never page-locked (PCIe-inclusive: 4 MiB of "
                                            "tables up, 18 MB of pixels back); _argb_only: the same pageable buffers, pixels_out = NULL -- "
                                            "only the 3.7 MB ARGB image comes back, which is all the reference's pipeline reads "
                                            "(core.clj:91-97): what a JNI caller of the reference's render loop gets; _pinned: the "
                                            "caller's buffers registered with rm_pin_host_buffer"}
        if world == 1 and not args.lean:
            # the other arithmetic contracts, same frame, strictly serial (reported, never `value`)
            out["other_contract"] = []
            for other in ("gfx950-default", "gfx950-strict", "cpu"):
                if other == args.contract:
                    continue
                fo = multigpu.FrameRenderer(vox, vres, opts, mc, n, width, device=dev, frames_in_flight=1, contract=other)
                oms = []
                for _ in range(6):
                    fo.render()
                    torch.cuda.synchronize(dev)
                    oms.append(fo.ctx.last_frame_timing()[0])
                fo.close()
                out["other_contract"].append({"contract": other, "kernel_ms": round(float(np.median(oms[1:])), 4)})
            out["other_contract_note"] = ("every contract is parity-checked bit for bit against its own checker "
                                          "(tests/test_gpu_device_contract.py, test_gpu_configs.py)")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(vox, opts, mc, n, spp, args.cpu_passes)
    if rank == 0:
        print(json.dumps(out), flush=True)  # before the teardown: the line must not depend on it
    fr.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def algorithmic_bytes(vox, vres, opts, mc, n, spp, device=0):
    """Exact event counts of the REFERENCE algorithm on this workload, from the counting variant
    of the kernel (same algorithm + counters; identical to the oracle's counters,
    tests/test_gpu_parity.py), untimed, all passes.  -> (bytes per frame, counters dict)"""
    from raymarchcl_amd import _native

    cnt = _native.Counters()
    cctx = _native.Context(device)
    cctx.set_volume(vox, vres)
    scratch = np.zeros(4 * n, dtype=np.float32)
    for i in range(spp):
        cctx.render_image(np.ascontiguousarray(mc[i]), opts[i * 544:(i + 1) * 544], scratch, n=n, counters=cnt)
    cctx.close()
    c = cnt.as_dict()
    return c["vox_reads"] * 1 + c["mc_reads"] * 16 + n * spp * 32, c


def library_bench(args):
    """--backend library: ONE process; the frame is tiled over N devices inside the C library
    (rm_create_multi).  Inputs and outputs live in the root device's HBM (rm_frame_device_full);
    per frame each device renders its interleaved tiles and peer-copies its accumulators into
    the root (xGMI), which un-permutes + tonemaps.  Frames are enqueued back to back."""
    import torch

    from raymarchcl_amd import _native

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the render path has no CPU fallback")
    rehearsal = os.environ.get("BENCH_ONE_DEVICE", "0") == "1"
    ids = [0] * args.gpus if rehearsal else list(range(args.gpus))
    if not rehearsal and torch.cuda.device_count() < args.gpus:
        sys.exit(f"--gpus {args.gpus} but only {torch.cuda.device_count()} device(s) visible")
    _native.build()
    wl = WORKLOADS[args.workload]
    vox, vres, opts, mc = build_inputs(wl)
    n, width, spp = wl["w"] * wl["h"], wl["w"], wl["spp"]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    ctx = _native.Context(ids if len(ids) > 1 else 0)
    ctx.set_contract(args.contract)
    ctx.set_volume(vox, vres)  # replicated to every device of the context
    d_opts = torch.frombuffer(bytearray(opts), dtype=torch.uint8).to(dev)
    d_mc = torch.from_numpy(np.ascontiguousarray(mc, dtype=np.float32).reshape(-1)).to(dev)
    d_px = torch.empty(4 * n, dtype=torch.float32, device=dev)
    d_argb = torch.empty(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize(dev)
    ctx.check_device_opts(d_opts.data_ptr(), spp, n, width)

    def frame():
        ctx.frame_device_full(d_opts.data_ptr(), d_mc.data_ptr(), spp, n, width, d_px.data_ptr(), d_argb.data_ptr())

    for _ in range(args.warmup):
        frame()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):  # one blocking frame at a time (core.clj:203-208), like the ranks backend
        frame()
        ctx.synchronize()
    elapsed = time.perf_counter() - t0
    kms, serial = [], []
    for _ in range(5):
        ts = time.perf_counter()
        frame()
        ctx.synchronize()
        serial.append((time.perf_counter() - ts) * 1e3)
        ms, launches = ctx.last_frame_timing()  # the root's partition
        kms.append(ms / launches)
    shares, frame_ms = ctx.last_frame_breakdown()
    # the same frames with only the ARGB image wanted: tonemapped words cross the links (4 B per pixel)
    for _ in range(args.warmup):
        ctx.frame_device_full(d_opts.data_ptr(), d_mc.data_ptr(), spp, n, width, None, d_argb.data_ptr())
    ctx.synchronize()
    ta = time.perf_counter()
    for _ in range(args.steps):
        ctx.frame_device_full(d_opts.data_ptr(), d_mc.data_ptr(), spp, n, width, None, d_argb.data_ptr())
    ctx.synchronize()
    ta = time.perf_counter() - ta
    ashares, aframe_ms = ctx.last_frame_breakdown()
    ctx.close()
    alg_frame, c = algorithmic_bytes(vox, vres, opts, mc, n, spp)
    world = len(ids)
    pass_ms = float(np.median(kms))
    achieved = alg_frame / launches / world / (pass_ms * 1e-3) / 1e9
    out = {
        "metric": "Mrays/s (primary rays = pixel samples per second) + ms/frame, " +
                  ("256^3 gyroid 1280x720x16spp" if args.workload == "c2" else wl["desc"]),
        "value": round(n * spp * args.steps / elapsed / 1e6, 3), "unit": "Mrays/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "ms_per_frame_serial": round(float(np.median(serial[1:])), 4), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": wl["desc"], "volume": f"{vres[0]}^3 u8", "resolution": [wl["w"], wl["h"]], "spp": spp,
                   "backend": "library (rm_create_multi: one process, peer copies over xGMI, no RCCL)",
                   "partition": f"8x8 tiles interleaved over {world} device(s)" +
                                (" (BENCH_ONE_DEVICE rehearsal: every rank on device 0)" if rehearsal and world > 1 else "")},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                     "kernel": "render_frame_kernel (the root device's partition)", "kernel_ms": round(pass_ms, 4),
                     "launches_per_frame": launches, "alg_bytes_per_launch": int(alg_frame / launches / world),
                     "alg_bytes_per_sample": round(alg_frame / (n * spp), 1)},
        "per_rank": {"share_render_ms": {"max": round(max(shares), 4), "mean": round(sum(shares) / len(shares), 4),
                                         "by_rank": [round(v, 4) for v in shares]},
                     "frame_ms_root": round(frame_ms, 4),
                     "note": "device time of the last frame: each device's partition kernels; frame_ms_root = start of the "
                             "root's share to the end of its resolve (waits for the slowest device + peer copies + resolve)"},
        "argb_exchange": {"ms_per_step": round(ta / args.steps * 1e3, 4),
                          "share_render_ms": {"max": round(max(ashares), 4), "mean": round(sum(ashares) / len(ashares), 4)},
                          "frame_ms_root": round(aframe_ms, 4),
                          "note": "d_pixels = NULL: every device tonemaps its tiles, 4 B per pixel are peer-copied instead of 16"},
    }
    print(json.dumps(out), flush=True)
    return 0


def cpu_baseline(vox, opts, mc, n, spp, passes):
    """The reference path on the host cores, bounded sample = the first `passes` passes of the
    frame.  If the reference kernel itself is there (oracle/_ref: the unmodified renderer.cl
    compiled for x86-64 in the build container, it travels as a prebuilt file) that is what is
    timed -- work-items dealt to all cores in chunks (ref_render_image_mt) -- and `kind` is
    "reference"; otherwise
    the plain-C restatement (oracle/rm_restate.c, bit-identical to it), `kind` "port"."""
    import oracle

    oracle.build(ref=False)
    cores = int(oracle.restate_lib().rmo_hw_threads())
    passes = max(1, min(passes, spp))

    def run_port():
        px = np.zeros(4 * n, dtype=np.float32)
        t0 = time.perf_counter()
        for i in range(passes):
            oracle.render_image(vox, np.ascontiguousarray(mc[i]), opts[i * 544:(i + 1) * 544], px, n=n,
                                threads=cores)
        return time.perf_counter() - t0

    def run_reference():
        oracle.ref_lib(False)  # raises if the prebuilt reference is not there
        px = np.zeros(4 * n, dtype=np.float32)
        t0 = time.perf_counter()
        for i in range(passes):
            oracle.ref_render_image_mt(vox, np.ascontiguousarray(mc[i]), opts[i * 544:(i + 1) * 544], px,
                                       cores, n=n)
        return time.perf_counter() - t0

    # warm-up outside the timers: thread pools, page faults of the volume and the tables
    warm = np.zeros(4 * n, dtype=np.float32)
    oracle.render_image(vox, np.ascontiguousarray(mc[0]), opts[:544], warm, n=n, id0=0, id1=min(n, 4096),
                        threads=cores)
    port_dt = min(run_port(), run_port())  # best of two: the first run of a process is reliably slower
    out = {"value": round(n * passes / port_dt / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
           "sample": f"passes 0..{passes - 1} of {spp}, all {n} pixels each ({port_dt:.1f} s wall)"}
    try:  # the compiled reference kernel, when its prebuilt library is there
        oracle.ref_render_image_mt(vox, np.ascontiguousarray(mc[0]), opts[:544], warm, cores, n=n, id0=0,
                                   id1=min(n, 4096))
        ref_dt = min(run_reference(), run_reference())
        out["reference_build"] = {
            "value": round(n * passes / ref_dt / 1e6, 4),
            "note": "unmodified renderer.cl compiled for x86-64, same sample and threads; its OpenCL built-ins "
                    "are out-of-line calls into a shim.  PREBUILT in the build container (oracle/_ref is "
                    "git-ignored and travels with the snapshot): a clean clone without /root/reference "
                    "reports only the port"}
    except Exception:
        pass
    try:  # the reference kernel itself on THIS GPU (prebuilt code object of the unmodified renderer.cl)
        if oracle.have_gfx950_ref("fast"):
            _px, _argb, ms = oracle.gfx950_render_frame(vox, opts[:passes * 544], mc[:passes], n, build="fast", tonemap=False)
            _px, _argb, ms = oracle.gfx950_render_frame(vox, opts[:passes * 544], mc[:passes], n, build="fast", tonemap=False)
            out["reference_kernel_on_this_gpu"] = {
                "value": round(n * passes / ms / 1e3, 2), "unit": "Mrays/s", "ms_per_pass": round(ms / passes, 3),
                "note": "RenderImage of the unmodified renderer.cl built by ROCm's OpenCL compiler for gfx950 with the "
                        "reference's own options (-cl-fast-relaxed-math -cl-mad-enable, core.clj:128), one launch per "
                        "pass, device time of the same sample; prebuilt in the build container (oracle/_ref)"}
    except Exception:
        pass
    return out


if __name__ == "__main__":
    main()
