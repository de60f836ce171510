"""Image-tile partition of one frame over the GPUs of a node.

Not a reference feature (the reference drives a single device, core.clj:122);
it is BASELINE.json's multi-GPU requirement.  Every work-item of the render
kernel depends only on its own global id (renderer.cl:469-473, 267, 334), so
any partition of the image gives bit-identical pixels.  Layout:

* the image is cut into 8x8-pixel tiles, numbered row-major;
* rank r of ``world`` owns tiles r, r+world, r+2*world, ... (interleaved,
  because cost per tile is very uneven: sky vs. multi-bounce surfaces);
* a rank keeps its accumulators tile-major -- ``tiles_per_part`` tiles of 64
  float4 -- which is also the unit that is exchanged;
* ONE collective per frame: gather of the tile-major accumulators to rank 0
  (torch.distributed -> RCCL over xGMI: point-to-point sends into the root, all
  links concurrently); the root un-permutes + tonemaps (rm_resolve_device).
* with ONE rank there is no partition: the frame kernel writes the row-major image and the
  ARGB words itself (rm_frame_device_full), one launch per frame of up to 16 passes.

The volume, the scatter tables and the option records are replicated.
"""
import numpy as np

TILE = 8
TILE_PIXELS = TILE * TILE


def tile_geometry(width, n):
    """-> (tiles_x, tiles_total) covering work-items 0..n-1 of an image ``width`` wide."""
    rows = -(-n // width)
    tiles_x = -(-width // TILE)
    return tiles_x, tiles_x * (-(-rows // TILE))


def tiles_per_part(width, n, world):
    return -(-tile_geometry(width, n)[1] // world)


def owner_of_tile(tile, world):
    """-> (rank, local tile index)"""
    return tile % world, tile // world


def gathered_index_map(width, n, world):
    """For every work-item id < n: its float4 index in the gathered buffer
    ``[world][tiles_per_part][64]``.  Host mirror of what resolve_kernel
    computes on the device (used by the CPU tests and to check the kernel)."""
    tiles_x, _total = tile_geometry(width, n)
    tpp = tiles_per_part(width, n, world)
    ids = np.arange(n, dtype=np.int64)
    x, y = ids % width, ids // width
    tile = (y // TILE) * tiles_x + (x // TILE)
    lane = (y % TILE) * TILE + (x % TILE)
    return ((tile % world) * tpp + tile // world) * TILE_PIXELS + lane


def local_work_items(width, n, rank, world):
    """-> (ids, local float4 index) of the work-items rank owns (ids < n only)."""
    idx = gathered_index_map(width, n, world)
    tpp = tiles_per_part(width, n, world)
    lo, hi = rank * tpp * TILE_PIXELS, (rank + 1) * tpp * TILE_PIXELS
    mine = np.nonzero((idx >= lo) & (idx < hi))[0]
    return mine, idx[mine] - lo


def gather_tiles(local_tiles, rank, world, dst=0, group=None):
    """Gather every rank's tile-major accumulators on ``dst``.

    local_tiles: float32 tensor [tiles_per_part*64*4] (CPU with gloo, GPU with
    nccl == RCCL).  Returns the [world * tiles_per_part*64*4] tensor on ``dst``,
    None elsewhere.  world == 1 returns the input.
    """
    if world == 1:
        return local_tiles
    import torch
    import torch.distributed as dist

    if local_tiles.is_cuda and dist.get_backend(group) == "gloo":
        # debugging / single-GPU test path: gloo cannot gather device tensors, stage via host
        host = gather_tiles(local_tiles.cpu(), rank, world, dst=dst, group=group)
        return host.to(local_tiles.device) if host is not None else None
    if rank == dst:
        out = torch.empty(world * local_tiles.numel(), dtype=local_tiles.dtype,
                          device=local_tiles.device)
        chunks = list(out.view(world, -1).unbind(0))
        dist.gather(local_tiles, gather_list=chunks, dst=dst, group=group)
        return out
    dist.gather(local_tiles, gather_list=None, dst=dst, group=group)
    return None


class _Slot:
    """What one frame in flight owns: a HIP stream, a library context bound to it, the tile
    accumulators (partitioned frames) and the outputs.  The volume and the tables derived from
    it belong to the first slot's context; the others share them (rm_share_volume)."""

    def __init__(self, torch, _native, dev, stream, tpp, n, root, want_pixels, want_argb, world):
        self.stream = stream
        self.d_tiles = (torch.zeros(tpp * TILE_PIXELS * 4, dtype=torch.float32, device=dev)
                        if world > 1 else None)
        # frames that want the ARGB image only are exchanged as tonemapped words (4 B per pixel, not 16)
        self.d_argb_tiles = (torch.zeros(tpp * TILE_PIXELS, dtype=torch.int32, device=dev)
                             if world > 1 and want_argb and not want_pixels else None)
        self.events = None  # render(timed=True): share start / share end / gather end / resolve end
        self.d_pixels = torch.empty(4 * n, dtype=torch.float32, device=dev) if (root and want_pixels) else None
        self.d_argb = torch.empty(n, dtype=torch.int32, device=dev) if (root and want_argb) else None
        self.ctx = _native.Context(dev.index or 0)
        self.ctx.set_stream(stream.cuda_stream)


class FrameRenderer:
    """Device-resident pipeline of one rank: inputs live in HBM as torch
    tensors, kernels run on torch streams, and ``render()`` does
    rm_frame_device -> (gather) -> rm_resolve_device on the root.

    ``frames_in_flight`` > 1 renders successive frames (the reference's animation
    loop, core.clj:202-208) on alternating HIP streams, each with its own staging and
    output buffers: the last wavefronts of frame i (a sample takes ~0.4 ms from camera
    ray to colour, so every launch ends with a tail of that length during which most
    of the chip idles) overlap the first of frame i+1.  A frame's own kernels stay
    ordered on its stream; per-frame results are unchanged.
    """

    def __init__(self, vox, vres, opts_bytes, mc, n, width, rank=0, world=1, device=None,
                 want_pixels=True, want_argb=True, group=None, frames_in_flight=1, contract=None):
        import torch

        from . import _native

        self.torch = torch
        self.rank, self.world, self.group = rank, world, group
        self.n, self.width = int(n), int(width)
        self.iters = len(opts_bytes) // _native.OPTS_BYTES
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        dev = self.device
        self.d_vox = torch.from_numpy(np.ascontiguousarray(vox)).to(dev)
        self.d_opts = torch.frombuffer(bytearray(opts_bytes), dtype=torch.uint8).to(dev)
        self.d_mc = torch.from_numpy(np.ascontiguousarray(mc, dtype=np.float32).reshape(-1)).to(dev)
        assert self.d_mc.numel() == self.iters * _native.TABLE_FLOATS
        self.tpp = _native.tiles_per_part(self.width, self.n, world)
        assert self.tpp == tiles_per_part(self.width, self.n, world)
        torch.cuda.synchronize(dev)  # inputs are uploaded before any slot's stream reads them
        root = rank == 0
        self.slots = []
        nslots = max(1, int(frames_in_flight))
        # The tables derived from the volume are built per (volume, isoVal) and rebuilt IN PLACE when
        # a launch needs another isoVal (rm_share_volume's rule: sharers must not render with
        # different thresholds at the same time).  Records of one frame may differ in isoVal; slots
        # in flight would then rebuild the tables under each other's launches -- such a frame gives
        # every slot its own volume context (and its own tables) instead of sharing.
        iso = {opts_bytes[i * _native.OPTS_BYTES + 284] for i in range(self.iters)}
        self.shared_tables = len(iso) == 1
        for i in range(nslots):
            # One frame at a time: torch's current stream (what a caller with its own stream
            # expects).  Several: streams created here back to back -- HIP deals streams to its
            # few hardware queues round-robin, so these land on distinct queues, whereas the
            # caller's stream may share one with them (two slots on one queue do not overlap).
            stream = torch.cuda.current_stream(dev) if nslots == 1 else torch.cuda.Stream(dev)
            slot = _Slot(torch, _native, dev, stream, self.tpp, self.n, root, want_pixels, want_argb, world)
            if contract:  # None: _native.DEFAULT_CONTRACT, else the library default (RM_CONTRACT_GFX950_DEFAULT)
                slot.ctx.set_contract(contract)
            if i == 0 or not self.shared_tables:
                slot.ctx.set_volume_device(self.d_vox.data_ptr(), vres)
            else:
                slot.ctx.share_volume(self.slots[0].ctx)
            slot.ctx.check_device_opts(self.d_opts.data_ptr(), self.iters, self.n, self.width)
            self.slots.append(slot)
        self.frame = 0
        self._timed_slot = None  # the slot of the last render(timed=True)
        # first slot under the names single-frame callers use
        self.ctx, self.d_tiles = self.slots[0].ctx, self.slots[0].d_tiles
        self.d_pixels, self.d_argb = self.slots[0].d_pixels, self.slots[0].d_argb
        torch.cuda.synchronize(dev)

    def render(self, timed=False):
        """One frame.  Asynchronous; returns the (pixels, argb) tensors of the slot it
        used, valid on the root once that slot's stream (or the device) is synchronised.
        timed: record events around this rank's share, the gather and the resolve on the slot's
        stream (``last_breakdown()`` reads them)."""
        torch = self.torch
        slot = self.slots[self.frame % len(self.slots)]
        self.frame += 1
        self._timed_slot = slot if timed else None
        if timed:
            if slot.events is None:
                slot.events = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            mark = lambda i: slot.events[i].record(slot.stream)  # noqa: E731
        else:
            mark = lambda i: None  # noqa: E731
        with torch.cuda.stream(slot.stream):
            if self.world == 1:  # the whole frame incl. tonemap is one kernel launch
                mark(0)
                slot.ctx.frame_device_full(self.d_opts.data_ptr(), self.d_mc.data_ptr(), self.iters, self.n,
                                           self.width,
                                           slot.d_pixels.data_ptr() if slot.d_pixels is not None else None,
                                           slot.d_argb.data_ptr() if slot.d_argb is not None else None)
                mark(1); mark(2); mark(3)  # (no gather, no separate resolve)
                return slot.d_pixels, slot.d_argb
            mark(0)
            if slot.d_argb_tiles is not None:  # ARGB only: every rank tonemaps its tiles, words are exchanged
                slot.ctx.frame_device_argb(self.d_opts.data_ptr(), self.d_mc.data_ptr(), self.iters, self.n,
                                           self.width, slot.d_tiles.data_ptr(), slot.d_argb_tiles.data_ptr(),
                                           self.rank, self.world)
                mark(1)
                allt = self._gather(slot, words=True)
                mark(2)
                if self.rank == 0:
                    slot.ctx.resolve_device_argb(allt.data_ptr(), self.world, self.n, self.width, slot.d_argb.data_ptr())
                mark(3)
                return slot.d_pixels, slot.d_argb
            slot.ctx.frame_device(self.d_opts.data_ptr(), self.d_mc.data_ptr(), self.iters, self.n,
                                  self.width, slot.d_tiles.data_ptr(), self.rank, self.world)
            mark(1)
            allt = self._gather(slot)
            mark(2)
            if self.rank == 0:
                slot.ctx.resolve_device(allt.data_ptr(), self.world, self.d_opts.data_ptr(), self.n,
                                        self.width,
                                        slot.d_pixels.data_ptr() if slot.d_pixels is not None else None,
                                        slot.d_argb.data_ptr() if slot.d_argb is not None else None)
            mark(3)
        return slot.d_pixels, slot.d_argb

    def last_breakdown(self):
        """(share_ms, gather_ms, resolve_ms) of the last ``render(timed=True)`` on this rank: device time of
        the rank's own kernels, of the collective as this rank's stream sees it (for the root: until the
        last rank's tiles have arrived, i.e. including the wait for the slowest rank), of the root's resolve."""
        slot = self._timed_slot
        if slot is None or slot.events is None:
            raise RuntimeError("last_breakdown(): the last frame was not rendered with render(timed=True)")
        slot.events[3].synchronize()
        e = slot.events
        return e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]), e[2].elapsed_time(e[3])

    def _gather(self, slot, words=False):
        """The frame's one collective (of the float4 accumulators, or -- words -- of the ARGB words).  With RCCL the root's receive buffer and its per-rank
        views are made once per slot (a frame of an 8-GPU share lasts well under a millisecond:
        per-frame allocations and list building would show); other backends go through
        gather_tiles()."""
        if self.world == 1:
            return slot.d_tiles
        import torch.distributed as dist

        mine = slot.d_argb_tiles if words else slot.d_tiles
        if dist.get_backend(self.group) != "nccl":
            return gather_tiles(mine, self.rank, self.world, group=self.group)
        if self.rank == 0:
            if getattr(slot, "d_all", None) is None:
                slot.d_all = self.torch.empty(self.world * mine.numel(), dtype=mine.dtype, device=self.device)
                slot.chunks = list(slot.d_all.view(self.world, -1).unbind(0))
            dist.gather(mine, gather_list=slot.chunks, dst=0, group=self.group)
            return slot.d_all
        dist.gather(mine, gather_list=None, dst=0, group=self.group)
        return None

    def close(self):
        self.torch.cuda.synchronize(self.device)
        for slot in self.slots:
            slot.ctx.close()
