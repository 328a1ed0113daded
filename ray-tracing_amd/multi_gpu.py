"""Multi-GPU combine steps: one process per GPU, scene replicated.

  tiles   : every rank renders the 64x64 screen tiles it owns (rt_render shard_rank/shard_count);
            tiles nobody else touches stay zero in each rank's buffer, so ONE all_reduce(sum) of the
            radiance buffers (RCCL over xGMI on GPUs, gloo in the CPU tests) is a gather of the tiles.
  samples : every rank renders all pixels with its own sample slice (sample_first, stream_id = rank);
            the f32 accumulation buffers are summed with one all_reduce and divided by the total
            sample count (summation order differs from one GPU: RMSE gate, not bits).
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


def wrap_device_buffer(ptr, n_floats, device_index):
    """View a device buffer of the library (rt_device_buffers) as a torch tensor without copying."""
    import torch

    class _Ptr:
        __cuda_array_interface__ = dict(shape=(n_floats,), typestr="<f4", data=(ptr, False), version=2)
    return torch.as_tensor(_Ptr(), device=f"cuda:{device_index}")
