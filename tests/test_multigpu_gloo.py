"""The N > 1 path on CPU: two ranks (gloo) each fill the tile-major
accumulators of their interleaved tile partition, the accumulators are
gathered on rank 0 with the same torch.distributed call the GPU path uses, and
the host mirror of the un-permute reproduces the full frame.  The per-rank
pixels come from the oracle here (tests may use it; the product cannot render
on CPU)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raymarchcl_amd import multigpu


def test_partition_geometry():
    for width, n, world in [(1280, 921600, 8), (50, 50 * 37 - 13, 2), (64, 64 * 48, 3), (7, 7, 4)]:
        tiles_x, total = multigpu.tile_geometry(width, n)
        tpp = multigpu.tiles_per_part(width, n, world)
        assert tpp * world >= total
        idx = multigpu.gathered_index_map(width, n, world)
        assert idx.size == n and np.unique(idx).size == n and idx.max() < world * tpp * 64
        seen = np.zeros(n, bool)
        for r in range(world):
            ids, loc = multigpu.local_work_items(width, n, r, world)
            assert not seen[ids].any() and (loc < tpp * 64).all()
            seen[ids] = True
            x, y = ids % width, ids // width
            tile = (y // 8) * tiles_x + x // 8
            assert (tile % world == r).all()  # interleaved ownership
        assert seen.all()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    import oracle
    import scenes

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = scenes.build("ragged_50x37")
    n, w, it = sc["n"], sc["w"], sc["iter"]
    ids, loc = multigpu.local_work_items(w, n, rank, world)
    tpp = multigpu.tiles_per_part(w, n, world)
    # render only the work-items this rank owns (runs of consecutive ids)
    full = np.zeros(4 * n, np.float32)
    for i in range(it):
        run_lo = prev = int(ids[0])
        runs = []
        for v in ids[1:]:
            v = int(v)
            if v != prev + 1:
                runs.append((run_lo, prev + 1))
                run_lo = v
            prev = v
        runs.append((run_lo, prev + 1))
        for lo, hi in runs:
            oracle.render_image(sc["vox"], sc["mc"][i].copy(), sc["opts"][i * 544:(i + 1) * 544], full,
                                n=n, id0=lo, id1=hi, threads=1)
    tiles = np.zeros((tpp * 64, 4), np.float32)
    tiles[loc] = full.reshape(-1, 4)[ids]
    got = multigpu.gather_tiles(torch.from_numpy(tiles.reshape(-1)), rank, world)
    if rank == 0:
        allt = got.numpy().reshape(-1, 4)
        img = allt[multigpu.gathered_index_map(w, n, world)].reshape(-1)
        want, _ = oracle.render_frame(sc["vox"], sc["opts"], sc["mc"], n, threads=1, tonemap=False)
        ok = np.array_equal(img.view(np.uint32), want.view(np.uint32))
        open(os.path.join(tmpdir, "result"), "w").write("ok" if ok else "mismatch")
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_reassembles_the_frame(tmp_path, oracle_mod):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert (tmp_path / "result").read_text() == "ok"
