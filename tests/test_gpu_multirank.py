"""The N > 1 code path end to end on the single-GPU box: two ranks share cuda:0,
each renders its interleaved tile partition with the HIP kernels, the tile-major
accumulators are gathered on rank 0 (gloo here, staged through the host -- RCCL
cannot place two ranks on one device), and rank 0's resolved frame must equal the
oracle bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gfx950_pin import pin_default as pin  # noqa: E402,F401  (the library default contract)

pytestmark = pytest.mark.gpu  # ranks run in the library's default contract (gfx950-default): checked against the `default` reference build / its recording


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmpdir, frames_in_flight=1, backend="gloo", want_pixels=True):
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    import scenes
    from raymarchcl_amd import multigpu

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # gloo: every rank on cuda:0 (single-GPU box); nccl == RCCL: one rank per device
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = scenes.build("metal_3spp")
    fr = multigpu.FrameRenderer(sc["vox"], sc["vres"], sc["opts"], sc["mc"], sc["n"], sc["w"], rank=rank,
                                world=world, device=dev, frames_in_flight=frames_in_flight, want_pixels=want_pixels)
    for _ in range(2 * frames_in_flight + 1):  # every slot reused at least once
        d_px, d_argb = fr.render()
    torch.cuda.synchronize()
    if rank == 0:
        if d_px is not None:
            np.save(os.path.join(tmpdir, "px.npy"), d_px.cpu().numpy())
        np.save(os.path.join(tmpdir, "argb.npy"), d_argb.cpu().numpy().view(np.uint32))
    dist.barrier()
    fr.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,frames_in_flight", [(2, 1), (3, 1), (2, 2)])
def test_tile_partition_over_ranks(tmp_path, pin, world, frames_in_flight):
    import scenes

    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), frames_in_flight), nprocs=world, join=True)
    sc = scenes.build("metal_3spp")
    want, want_argb = pin.frame("metal_3spp", sc["vox"], sc["opts"], sc["mc"], sc["n"])
    px = np.load(tmp_path / "px.npy")
    argb = np.load(tmp_path / "argb.npy")
    assert np.array_equal(px.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(argb, want_argb)


@pytest.mark.parametrize("frames_in_flight", [1, 2])
def test_tile_partition_over_rccl(tmp_path, pin, frames_in_flight):
    """The same with one rank per GPU and the gather on the `nccl` backend (= RCCL over xGMI),
    side streams included -- the path bench.py --gpus N takes.  Needs >= 2 devices: runs on the
    driver's multi-GPU node, skipped on the single-GPU box."""
    import scenes

    world = torch.cuda.device_count()
    if world < 2:
        pytest.skip(f"{world} device(s): RCCL cannot place two ranks on one GPU")
    world = min(world, 8)
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), frames_in_flight, "nccl"), nprocs=world, join=True)
    sc = scenes.build("metal_3spp")
    want, want_argb = pin.frame("metal_3spp", sc["vox"], sc["opts"], sc["mc"], sc["n"])
    assert np.array_equal(np.load(tmp_path / "px.npy").view(np.uint32), want.view(np.uint32))
    assert np.array_equal(np.load(tmp_path / "argb.npy"), want_argb)


@pytest.mark.parametrize("world,frames_in_flight", [(2, 1), (3, 2)])
def test_argb_only_frames_exchange_tonemapped_words(tmp_path, pin, world, frames_in_flight):
    """want_pixels=False: every rank tonemaps its own tiles in the frame kernel, the ARGB words (4 B per
    pixel) are gathered and un-permuted by the root (rm_frame_device_argb / rm_resolve_device_argb):
    the image equals the reference build's TonemapImage output word for word."""
    import scenes

    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), frames_in_flight, "gloo", False), nprocs=world, join=True)
    sc = scenes.build("metal_3spp")
    _want, want_argb = pin.frame("metal_3spp", sc["vox"], sc["opts"], sc["mc"], sc["n"])
    assert not os.path.exists(tmp_path / "px.npy")
    assert np.array_equal(np.load(tmp_path / "argb.npy"), want_argb)


def _rccl_one_rank(rank, port, tmpdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)  # bench.py's and _worker's call
    ok = []
    for stream in (torch.cuda.current_stream(dev), torch.cuda.Stream(dev)):  # one frame at a time / a slot's side stream
        with torch.cuda.stream(stream):
            # the collective exactly as multigpu.FrameRenderer._gather issues it on `nccl`: the root's receive buffer made
            # once, its per-rank views as the gather list
            mine = torch.arange(64 * 64 * 4, dtype=torch.float32, device=dev) * 0.5
            d_all = torch.empty(1 * mine.numel(), dtype=mine.dtype, device=dev)
            chunks = list(d_all.view(1, -1).unbind(0))
            dist.gather(mine, gather_list=chunks, dst=0)
            words = torch.arange(4096, dtype=torch.int32, device=dev)  # ARGB-only frames exchange int32 words
            w_all = torch.empty(words.numel(), dtype=torch.int32, device=dev)
            dist.gather(words, gather_list=list(w_all.view(1, -1).unbind(0)), dst=0)
        stream.synchronize()
        ok.append(bool(torch.equal(d_all, mine)) and bool(torch.equal(w_all, words)))
    tt = torch.tensor([1.25], dtype=torch.float64, device=dev)  # bench.py: max over ranks of the elapsed time
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dist.barrier()
    torch.cuda.synchronize(dev)
    ok.append(float(tt.item()) == 1.25)
    with open(os.path.join(tmpdir, "ok.txt"), "w") as f:
        f.write(" ".join(str(int(o)) for o in ok))
    dist.destroy_process_group()


def test_rccl_initialises_and_runs_the_renderers_collectives_on_one_rank(tmp_path):
    """What CAN be executed of the RCCL side on a single-GPU box: the process group of bench.py / FrameRenderer comes up
    on this box (`nccl` backend, device_id, the box's IPC mode) and the collectives the renderer issues -- gather into
    views of one receive buffer, on the current and on a side stream; all-reduce MAX; barrier -- run on it.  The N > 1
    exchange itself needs >= 2 devices (test_tile_partition_over_rccl)."""
    mp.spawn(_rccl_one_rank, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    assert open(tmp_path / "ok.txt").read() == "1 1 1"
