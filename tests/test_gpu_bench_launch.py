"""`python bench.py --gpus N` as the driver invokes it (no launcher): the script re-launches itself
under torch.distributed.run with one rank per GPU.  Rehearsed here on the single-GPU box with
BENCH_ONE_DEVICE=1 (every rank on cuda:0, gloo instead of RCCL -- RCCL cannot place two ranks on one
device), so that the launcher, the rendezvous, the partition and the gather are covered; and the
one-process client of the same partition (--backend library: rm_create_multi + rm_frame_device_full
tiled over the context's devices) against a single-device frame."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import scenes

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("cpu_contract")]  # this module checks against the CPU oracle
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(*flags):
    env = dict(os.environ)
    env["BENCH_ONE_DEVICE"] = "1"
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_launches_its_own_ranks():
    out = _run_bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "c1")
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["value"] > 0
    assert out["scaling"] == "strong" and "roofline" in out
    assert "gloo" in out["config"]["partition"]  # the rehearsal says what it is


def test_bench_library_backend():
    out = _run_bench("--gpus", "3", "--backend", "library", "--steps", "3", "--warmup", "1", "--workload", "c1")
    assert out["n_gpus"] == 3 and out["value"] > 0 and out["roofline"]["launches_per_frame"] == 1
    assert "rm_create_multi" in out["config"]["backend"]


@pytest.mark.parametrize("ranks", [2, 5])
def test_device_resident_frames_over_the_devices_of_a_context(native, oracle_mod, ranks):
    """rm_frame_device_full on a multi-device context: inputs and outputs in the root's memory,
    frames enqueued back to back (the gather buffer is reused: each peer copy waits for the
    previous frame's resolve), every frame == the oracle."""
    import torch

    sc = scenes.build("orange_dof_2spp")
    n, it = sc["n"], sc["iter"]
    want_px, want_argb = oracle_mod.render_frame(sc["vox"], sc["opts"], sc["mc"], n)
    dev = torch.device("cuda", 0)
    d_opts = torch.frombuffer(bytearray(sc["opts"]), dtype=torch.uint8).to(dev)
    d_mc = torch.from_numpy(np.ascontiguousarray(sc["mc"], np.float32).reshape(-1)).to(dev)
    outs = [(torch.zeros(4 * n, dtype=torch.float32, device=dev), torch.zeros(n, dtype=torch.int32, device=dev))
            for _ in range(4)]
    torch.cuda.synchronize()
    with native.Context([0] * ranks) as ctx:
        ctx.set_volume(sc["vox"], sc["vres"])
        ctx.check_device_opts(d_opts.data_ptr(), it, n, sc["w"])
        for px, argb in outs:  # four frames in a row, no host synchronisation in between
            ctx.frame_device_full(d_opts.data_ptr(), d_mc.data_ptr(), it, n, sc["w"], px.data_ptr(), argb.data_ptr())
        ctx.synchronize()
        for px, argb in outs:
            assert np.array_equal(px.cpu().numpy().view(np.uint32), want_px.view(np.uint32))
            assert np.array_equal(argb.cpu().numpy().view(np.uint32), want_argb)
        # new tables behind the same pointers: validated again -> replicated again
        mc2 = np.ascontiguousarray(sc["mc"][::-1], np.float32)
        d_mc.copy_(torch.from_numpy(mc2.reshape(-1)))
        torch.cuda.synchronize()
        ctx.check_device_opts(d_opts.data_ptr(), it, n, sc["w"])
        px, argb = outs[0]
        ctx.frame_device_full(d_opts.data_ptr(), d_mc.data_ptr(), it, n, sc["w"], px.data_ptr(), argb.data_ptr())
        ctx.synchronize()
        want2, _ = oracle_mod.render_frame(sc["vox"], sc["opts"], mc2, n)
        assert np.array_equal(px.cpu().numpy().view(np.uint32), want2.view(np.uint32))


def test_host_and_device_frames_interleaved_on_a_multi_device_context(native, oracle_mod):
    """rm_frame_device_full -> rm_render_frame (other records and tables, host buffers) ->
    rm_frame_device_full with the first call's pointers: the host call overwrites the other devices'
    copies of the records and tables, so the third call must replicate again."""
    import torch

    sc = scenes.build("orange_dof_2spp")
    n, it = sc["n"], sc["iter"]
    want_px, want_argb = oracle_mod.render_frame(sc["vox"], sc["opts"], sc["mc"], n)
    other = scenes.build("orange_dof_2spp", mc_seed=77)
    opts2 = bytearray(other["opts"])
    opts2[260:264] = np.float32(1.25).tobytes()  # another exposure in record 0 (Appendix A: offset 260)
    opts2[544 + 260:544 + 264] = np.float32(1.25).tobytes()
    want2, _ = oracle_mod.render_frame(sc["vox"], bytes(opts2), other["mc"], n)
    dev = torch.device("cuda", 0)
    d_opts = torch.frombuffer(bytearray(sc["opts"]), dtype=torch.uint8).to(dev)
    d_mc = torch.from_numpy(np.ascontiguousarray(sc["mc"], np.float32).reshape(-1)).to(dev)
    px = torch.zeros(4 * n, dtype=torch.float32, device=dev)
    argb = torch.zeros(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    with native.Context([0, 0, 0]) as ctx:
        ctx.set_volume(sc["vox"], sc["vres"])
        ctx.check_device_opts(d_opts.data_ptr(), it, n, sc["w"])
        ctx.frame_device_full(d_opts.data_ptr(), d_mc.data_ptr(), it, n, sc["w"], px.data_ptr(), argb.data_ptr())
        ctx.synchronize()
        assert np.array_equal(px.cpu().numpy().view(np.uint32), want_px.view(np.uint32))
        got2, _ = ctx.render_frame(bytes(opts2), other["mc"], n)
        assert np.array_equal(got2.view(np.uint32), want2.view(np.uint32))
        px.zero_()
        ctx.frame_device_full(d_opts.data_ptr(), d_mc.data_ptr(), it, n, sc["w"], px.data_ptr(), argb.data_ptr())
        ctx.synchronize()
        assert np.array_equal(px.cpu().numpy().view(np.uint32), want_px.view(np.uint32))
        assert np.array_equal(argb.cpu().numpy().view(np.uint32), want_argb)


@pytest.mark.parametrize("ranks", [2, 5])
def test_argb_only_device_resident_frames_inside_the_library(native, ranks):
    """rm_frame_device_full(d_pixels = NULL) on a multi-device context: the devices exchange tonemapped
    words instead of accumulators; same image as the float path, frame after frame; the per-device
    breakdown is reported."""
    import torch

    sc = scenes.build("metal_3spp")
    n, it = sc["n"], sc["iter"]
    dev = torch.device("cuda", 0)
    d_opts = torch.frombuffer(bytearray(sc["opts"]), dtype=torch.uint8).to(dev)
    d_mc = torch.from_numpy(np.ascontiguousarray(sc["mc"], np.float32).reshape(-1)).to(dev)
    px = torch.zeros(4 * n, dtype=torch.float32, device=dev)
    argb_f = torch.zeros(n, dtype=torch.int32, device=dev)
    argb_w = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(3)]
    torch.cuda.synchronize()
    with native.Context([0] * ranks) as ctx:
        ctx.set_volume(sc["vox"], sc["vres"])
        ctx.check_device_opts(d_opts.data_ptr(), it, n, sc["w"])
        ctx.frame_device_full(d_opts.data_ptr(), d_mc.data_ptr(), it, n, sc["w"], px.data_ptr(), argb_f.data_ptr())
        for a in argb_w:  # words-only frames back to back, then a float frame again
            ctx.frame_device_full(d_opts.data_ptr(), d_mc.data_ptr(), it, n, sc["w"], None, a.data_ptr())
        ctx.synchronize()
        shares, frame_ms = ctx.last_frame_breakdown()
        assert len(shares) == ranks and all(s > 0 for s in shares) and frame_ms >= max(shares) * 0.5
        want = argb_f.cpu().numpy()
        assert len(np.unique(want)) > 100
        for a in argb_w:
            assert np.array_equal(a.cpu().numpy(), want)
        px2 = torch.zeros_like(px)
        ctx.frame_device_full(d_opts.data_ptr(), d_mc.data_ptr(), it, n, sc["w"], px2.data_ptr(), None)
        ctx.synchronize()
        assert torch.equal(px, px2)
