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


def launch_ranks(n, argv, poll_s=0.2, grace_s=5.0, env_extra=None):
    """Start n rank processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment; rank 0 inherits stdout, the
    others' stdout is dropped) and wait for them by POLLING: the first rank that exits with a non-zero status ends the run --
    its siblings, which would otherwise sit in a collective until its timeout, are terminated (they are fresh child
    processes of a parent that never touches a GPU; nothing is exec'ed over a process that did) -- and that status is
    returned at once.  0 when every rank exited with 0.  bench.py --gpus N without a launcher runs through this."""
    import os
    import socket
    import subprocess
    import time
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if env_extra:
            env.update(env_extra)
        procs.append(subprocess.Popen(list(argv), env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    live = set(range(n))
    while live and rc == 0:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0:
                rc = abs(code) or 1
                break
        if live and rc == 0:
            time.sleep(poll_s)
    if rc != 0:
        for r in live:
            procs[r].terminate()
        deadline = time.time() + grace_s
        for r in live:
            try:
                procs[r].wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                procs[r].kill()
                procs[r].wait()
    return rc


def frames_identical(a, b):
    """Bitwise comparison of two frames (numpy arrays of one dtype and shape): what bench.py prints at N > 1 in tiles mode
    as parity.combined_equals_unsharded -- the frame rank 0 holds after the exchange against the same frame rendered by one
    rank alone.  Tiles are disjoint and every pixel is computed by exactly one rank with the same arithmetic: bits, not RMSE."""
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    same = a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()
    differing = 0 if same else (int((a.reshape(-1).view(np.uint8) != b.reshape(-1).view(np.uint8)).sum()) if a.shape == b.shape and a.dtype == b.dtype else -1)
    return {"identical": bool(same), "differing_bytes": differing, "pixels": int(a.size // 4) if a.ndim else 0}


def frames_rmse(a, b):
    """RMSE of two radiance frames in float64: the gate of the sample-sliced mode (summation order differs from one GPU doing
    all samples, so bits are not expected; BASELINE.json: <= 1e-3)."""
    d = np.asarray(a, np.float64).reshape(-1) - np.asarray(b, np.float64).reshape(-1)
    return float(np.sqrt(np.mean(d * d))) if d.size else 0.0


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


def _tiles_of(vw, vh, tile):
    return (vw + tile - 1) // tile, (vh + tile - 1) // tile


def tile_slot_floats(vw, vh, tile, count):
    """Floats in one rank's packed-tile message, sized for the rank that owns the most tiles (rank 0): what
    rt_tile_slot_bytes returns, / 4.  (vw, vh) = the covered rect."""
    tx, ty = _tiles_of(vw, vh, tile)
    return ((tx * ty + count - 1) // count) * tile * tile * 4


def pack_tiles_host(frame, rect, tile, count, rank):
    """The packed layout of rt_pack_tiles / k_tile_pack restated on the host (numpy): rank's owned tiles g = rank,
    rank + count, ... (row-major over the covered rect = (x, y, w, h)) one after the other, tile*tile RGBA f32 each,
    zeros where an edge tile leaves the rect.  Used by the gloo rehearsal / CPU tests, and the GPU tests hold the HIP
    kernel to it byte for byte.  frame: (H, W, 4) float32."""
    vx, vy, vw, vh = rect
    tx, ty = _tiles_of(vw, vh, tile)
    owned = range(rank, tx * ty, count)
    out = np.zeros((len(owned), tile, tile, 4), np.float32)
    for k, g in enumerate(owned):
        x0, y0 = (g % tx) * tile, (g // tx) * tile
        w, h = min(tile, vw - x0), min(tile, vh - y0)
        out[k, :h, :w] = frame[vy + y0:vy + y0 + h, vx + x0:vx + x0 + w]
    return out.reshape(-1)


def unpack_tiles_host(frame, slots, rect, tile, count, ranks):
    """The scatter of rt_unpack_tiles / k_tile_unpack on the host: slots[i] holds the packed tiles of ranks[i]."""
    vx, vy, vw, vh = rect
    tx, ty = _tiles_of(vw, vh, tile)
    for slot, rank in zip(slots, ranks):
        owned = range(rank, tx * ty, count)
        tiles = np.asarray(slot[: len(owned) * tile * tile * 4]).reshape(len(owned), tile, tile, 4)
        for k, g in enumerate(owned):
            x0, y0 = (g % tx) * tile, (g // tx) * tile
            w, h = min(tile, vw - x0), min(tile, vh - y0)
            frame[vy + y0:vy + y0 + h, vx + x0:vx + x0 + w] = tiles[k, :h, :w]
    return frame


class TileGather:
    """Packed-tile gather to rank 0: every rank packs the tiles it rendered into one contiguous message,
    ONE dist.gather moves the N - 1 messages straight to rank 0 (RCCL: one send per peer, one xGMI link each), and rank 0
    scatters them into its frame.  No per-frame Python loop over ranks and no full-frame copy.

    Device path (ctx given): rt_pack_tiles / rt_unpack_tiles -- two HIP kernels on the context's stream, the packed
    buffers are torch tensors on that device, and combine() must be called with that stream current (the caller runs its
    loop under torch.cuda.stream(s) with ctx.set_stream(s.cuda_stream)), so pack -> gather -> unpack are ordered.
    Host path (ctx None; gloo rehearsal and CPU tests): the same layout through pack_tiles_host / unpack_tiles_host on a
    host frame the caller passes in."""

    def __init__(self, width, height, tile, world, rank, ctx=None, device="cpu", rect=None, group=None, in_flight=1):
        import torch
        self.world, self.rank, self.tile, self.group, self.ctx = world, rank, tile, group, ctx
        self.rect = tuple(rect) if rect is not None else (0, 0, width, height)
        self.slot = tile_slot_floats(self.rect[2], self.rect[3], tile, world)
        if ctx is not None:
            assert ctx.tile_slot_bytes(world, tile) == self.slot * 4
        # one set of message buffers per frame in flight: frame k's gather may still be under way when frame k + 1 packs.
        # On the device path a set belongs to ONE stream (combine_in_flight hands them out per rt_get_stream handle), so
        # a set's reuse is ordered behind its last use by the stream itself.
        self.in_flight = max(1, in_flight)
        self._device = device
        self.sets = [self._new_set() for _ in range(self.in_flight)]
        self.send, self.recv, self.recv_list = self.sets[0]
        self._by_stream = {}

    def _new_set(self):
        import torch
        send = torch.zeros(self.slot, dtype=torch.float32, device=self._device)
        recv = torch.zeros(self.slot * self.world, dtype=torch.float32, device=self._device) if self.rank == 0 else None
        return (send, recv, list(recv.view(self.world, self.slot).unbind(0)) if self.rank == 0 else None)

    def combine_in_flight(self):
        """Device path: pack, gather and unpack of the frame rendered last, ordered on THAT frame's stream (rt_get_stream,
        made torch's current stream for the three calls, so the collective waits for the pack and the unpack for the
        collective) with the message buffers that belong to that stream.  With rt_set_frames_in_flight(n > 1) and the
        context on its own streams the next rt_render goes to the next set's stream and overlaps with this exchange;
        with one frame in flight (or a caller's stream) it is combine() on the context's stream."""
        import torch
        ptr = self.ctx.get_stream()
        if ptr not in self._by_stream:
            k = len(self._by_stream)
            bufs = self.sets[k] if k < len(self.sets) else self._new_set()     # a stream not seen before (the context was re-configured)
            self._by_stream[ptr] = (torch.cuda.ExternalStream(ptr) if ptr else torch.cuda.default_stream(), bufs)
        st, (self.send, self.recv, self.recv_list) = self._by_stream[ptr]
        with torch.cuda.stream(st):
            return self.combine(None)

    def combine(self, frame=None):
        """Device path: frame is ignored (the context's radiance buffer is source and destination).  Host path: frame is
        this rank's (H, W, 4) float32 numpy array; rank 0's is completed in place and returned."""
        import torch
        import torch.distributed as dist
        if self.ctx is not None:
            self.ctx.pack_tiles(self.rank, self.world, self.tile, self.send.data_ptr(), self.slot * 4)
        else:
            self.send.copy_(torch.from_numpy(np.resize(pack_tiles_host(frame, self.rect, self.tile, self.world, self.rank), self.slot)))
        dist.gather(self.send, self.recv_list, dst=0, group=self.group)
        if self.rank == 0 and self.world > 1:
            if self.ctx is not None:   # every peer's tiles in one launch; rank 0's own tiles are already in place
                self.ctx.unpack_tiles(1, self.world - 1, self.world, self.tile, self.recv.data_ptr() + self.slot * 4, self.slot * 4)
            else:
                unpack_tiles_host(frame, self.recv.view(self.world, self.slot)[1:].numpy(), self.rect, self.tile, self.world, range(1, self.world))
        return frame
