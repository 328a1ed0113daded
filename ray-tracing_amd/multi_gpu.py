"""Multi-GPU combine steps: one process per GPU, scene replicated.

  tiles   : every rank renders the 64x64 screen tiles it owns (rt_render shard_rank/shard_count);
            tiles nobody else touches stay zero in each rank's buffer, so ONE all_reduce(sum) of the
            radiance buffers (RCCL over xGMI on GPUs, gloo in the CPU tests) is a gather of the tiles.
  samples : every rank renders all pixels with its own sample slice (sample_first, stream_id = rank);
            the f32 accumulation buffers are summed with one all_reduce and divided by the total
            sample count (summation order differs from one GPU: RMSE gate, not bits).
  tiles, gathered : the same tiles, but each rank sends ONLY its own pixels, packed, straight to rank 0
            (TileGather).  xGMI is point-to-point: a ring all_reduce of the whole frame moves
            2*(N-1)/N frames over every link (1080p RGBA f32, 8 GPUs: 58 MB per link, ~0.38 ms at 153 GB/s),
            the direct gather 1/N of a frame over each of the N-1 links into rank 0 (4.1 MB, ~0.03 ms).
  samples, direct : SampleReduce -- the same sum as a direct reduce-scatter: every rank sends slice j of its running
            sums straight to rank j (all_to_all: N-1 sends of 1/N of the buffer, one per xGMI link, all links busy at
            once), adds the N slices it received, and the slices are gathered.  A ring moves 2*(N-1)/N of the buffer
            over every link one hop at a time (8K RGBA f32, 8 GPUs: 0.93 GB per link, ~6 ms); the direct form moves
            66 MB per link per phase (~0.45 ms each).  SURVEY 8e.
These helpers only move tensors; they work on any torch device.
"""
import numpy as np


def tile_owner(width, height, tile, count):
    """(H, W) int array: rank that owns each pixel under round-robin tile sharding
    (tile index = ty * tiles_x + tx; owner = index % count), as rt_render does."""
    tiles_x = (width + tile - 1) // tile
    ys, xs = np.mgrid[0:height, 0:width]
    return ((ys // tile) * tiles_x + (xs // tile)) % max(count, 1)


def combine_tiles(radiance, group=None):
    """radiance: this rank's frame tensor with zeros outside its tiles. In-place sum over ranks."""
    import torch.distributed as dist
    dist.all_reduce(radiance, op=dist.ReduceOp.SUM, group=group)
    return radiance


def combine_samples(accum, total_samples, group=None):
    """accum: this rank's running sums (W*H*4). Returns the resolved radiance = sum / total_samples."""
    import torch.distributed as dist
    dist.all_reduce(accum, op=dist.ReduceOp.SUM, group=group)
    return accum / float(total_samples)


class SampleReduce:
    """Direct reduce-scatter + gather of the accumulation buffers over point-to-point links (all_to_all), instead of a
    ring all_reduce.  combine(accum, total_samples) returns the resolved radiance (sum / total_samples): on every rank
    with gather_to=None (all_gather), else only on rank gather_to (None elsewhere).  Summation order: ranks 0..N-1 per
    element, the same on every rank (a ring's order depends on the element's position) -- still not the order of one
    GPU doing all samples, so the RMSE gate applies, not bits."""

    def __init__(self, n_floats, world, rank, device="cpu", group=None, gather_to=None):
        import torch
        self.world, self.rank, self.group, self.gather_to, self.n = world, rank, group, gather_to, n_floats
        self.slice = (n_floats + world - 1) // world                       # padded so that every slice has one size
        self.send = torch.zeros(self.slice * world, dtype=torch.float32, device=device)
        self.recv = torch.zeros(self.slice * world, dtype=torch.float32, device=device)
        self.part = torch.zeros(self.slice, dtype=torch.float32, device=device)
        self.full = torch.zeros(self.slice * world, dtype=torch.float32, device=device)

    def combine(self, accum, total_samples):
        import torch.distributed as dist
        self.send[: self.n].copy_(accum.reshape(-1))
        dist.all_to_all_single(self.recv, self.send, group=self.group)      # slice j of every rank -> rank j
        parts = self.recv.view(self.world, self.slice)
        self.part.copy_(parts[0])
        for r in range(1, self.world):                                      # fixed order: rank 0, 1, ..., N-1
            self.part.add_(parts[r])
        self.part.div_(float(total_samples))
        if self.gather_to is None:
            dist.all_gather_into_tensor(self.full, self.part, group=self.group)
            return self.full[: self.n]
        out = list(self.full.view(self.world, self.slice).unbind(0)) if self.rank == self.gather_to else None
        dist.gather(self.part, out, dst=self.gather_to, group=self.group)
        return self.full[: self.n] if self.rank == self.gather_to else None


def wrap_device_buffer(ptr, n_floats, device_index):
    """View a device buffer of the library (rt_device_buffers) as a torch tensor without copying."""
    import torch

    class _Ptr:
        __cuda_array_interface__ = dict(shape=(n_floats,), typestr="<f4", data=(ptr, False), version=2)
    return torch.as_tensor(_Ptr(), device=f"cuda:{device_index}")


class TileGather:
    """Packed-tile gather to rank 0.  Every rank holds a full-frame tensor (H*W*C floats) in which only its own
    tiles are meaningful; combine() leaves the complete frame in rank 0's tensor.  Index tensors are built once."""

    def __init__(self, width, height, tile, world, rank, channels=4, device="cpu", group=None):
        import torch
        self.world, self.rank, self.channels, self.group = world, rank, channels, group
        owner = tile_owner(width, height, tile, world).reshape(-1)
        self.idx = [torch.from_numpy(np.flatnonzero(owner == r).astype(np.int64)).to(device) for r in range(world)]
        self.n_max = max(int(i.numel()) for i in self.idx)          # gather wants equal sizes: pad to the largest share
        self.send = torch.zeros(self.n_max * channels, dtype=torch.float32, device=device)
        self.recv = [torch.zeros_like(self.send) for _ in range(world)] if rank == 0 else None

    def combine(self, frame):
        import torch.distributed as dist
        px = frame.view(-1, self.channels)
        mine = self.idx[self.rank]
        self.send.view(-1, self.channels)[: mine.numel()].copy_(px.index_select(0, mine))
        dist.gather(self.send, self.recv, dst=0, group=self.group)
        if self.rank == 0:
            for r in range(1, self.world):
                px.index_copy_(0, self.idx[r], self.recv[r].view(-1, self.channels)[: self.idx[r].numel()])
        return frame
