"""Multi-GPU sharding of the path-trace job: two decompositions, one exchange each, over RCCL/xGMI.

The reference is single-GPU (no collective anywhere).  Every (pixel, sample) pair is an
independent path and every rank holds a full scene replica, so the job can be cut either way:

* by PIXELS -- interleaved row bands (band b belongs to rank b % world; `shard_rows`, `gather_frame`,
  `FrameGatherer`): each rank renders its rows into a packed local float4 buffer with RNG streams keyed on the
  *global* pixel index, so an N-rank image is bit-identical to the 1-rank image; the exchange is one gather of the
  packed bands to rank 0 (grouped ncclSend/ncclRecv, one xGMI link per peer into the root).  This is the north-star
  decomposition and what `bench.py --gpus N` runs: every rank traces ALL samples of its rows with (4x4 chunk, sample)
  work items (BM_FLAG_SAMPLE_ITEMS), which keeps the persistent waves fed on 1/N of the pixels -- a 1/8 shard at 8 spp
  costs 1.80 ms against 1.78 ms for the full frame at 1 spp (tools/shard_time.py; with pixel items it was 4.29 ms).
* by SAMPLES -- every rank renders the full frame with its own slice of the sample indices (`sample_base` of
  bm_frame_params; `FrameReducer`): the exchange is ONE sum-reduction of the float4 frames to rank 0 after the last
  step (the per-rank buffers are additive).  `bench.py --decomposition samples`.  The sum of N partial frames differs
  from one N x spp render only in floating-point association (~1e-7 relative).

No collective sits on the data path of the render itself.
"""
import numpy as np

DEFAULT_BAND_ROWS = 16  # one row of 16x16 tiles
_cache = {}  # receive buffers / row indices, keyed by the gather's shape (per-frame gathers reuse them)


def shard_rows(height, band_rows, rank, world):
    """Global row indices owned by `rank`, in the order they are packed in its local buffer."""
    rows = np.arange(height)
    return rows[(rows // band_rows) % world == rank]


def assembly_index(height, band_rows, world):
    """For every global row y: its position in the stacked receive buffer [world * max_rows] (rank-major, every rank's
    packed rows padded to the largest shard) -- the frame is assembled with ONE index_select over it."""
    counts = [len(shard_rows(height, band_rows, r, world)) for r in range(world)]
    max_rows = max(counts)
    index = np.empty(height, np.int64)
    for r in range(world):
        index[shard_rows(height, band_rows, r, world)] = r * max_rows + np.arange(counts[r])
    return index


def gather_frame(local, height, band_rows, group=None, dst=0):
    """Gather every rank's packed rows ([local_rows, W, C] tensors, any backend) to `dst`.

    Returns the assembled [height, W, C] frame on `dst` and None elsewhere.  Ranks may own
    different numbers of rows (ragged last band); buffers are padded to the largest shard so a
    single equal-sized gather can be used.
    """
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        assert local.shape[0] == height
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    counts = [len(shard_rows(height, band_rows, r, world)) for r in range(world)]
    assert local.shape[0] == counts[rank], f"rank {rank}: {local.shape[0]} local rows, expected {counts[rank]}"
    max_rows = max(counts)
    out_device = local.device
    if local.is_cuda and dist.get_backend(group) == "gloo":
        local = local.cpu()  # test-only path: gloo has no device gather; RCCL ("nccl") gathers device buffers directly
    if local.shape[0] != max_rows:
        pad = torch.zeros((max_rows - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        send = torch.cat([local, pad], dim=0)
    else:
        send = local.contiguous()
    if rank == dst:
        key = (tuple(send.shape), send.dtype, str(send.device), height, band_rows, world)
        if key not in _cache:
            recv_all = torch.empty((world,) + tuple(send.shape), dtype=send.dtype, device=send.device)
            _cache[key] = (recv_all, [recv_all[r] for r in range(world)],
                           torch.as_tensor(assembly_index(height, band_rows, world), device=local.device, dtype=torch.long),
                           torch.empty((height,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device))
        recv_all, recv, index, out = _cache[key]
        dist.gather(send, gather_list=recv, dst=dst, group=group)
        torch.index_select(recv_all.view((-1,) + tuple(send.shape[1:])), 0, index, out=out)  # one kernel: row y <- (its rank, its packed row)
        return out.to(out_device)
    dist.gather(send, gather_list=None, dst=dst, group=group)
    return None


class FrameGatherer:
    """Pipelined form of gather_frame for a render loop: `start(local)` snapshots this rank's packed rows into a send
    buffer and launches the gather asynchronously (on RCCL's stream, behind the work already queued on the current
    stream), so that it overlaps the next frame's rendering; `finish()` waits for it and, on `dst`, returns the
    assembled [height, W, C] frame (None elsewhere; the returned tensor is reused by the next finish()).  At most one
    gather is in flight.  The root receives every peer's packed bands into one stacked buffer and assembles the frame
    with ONE precomputed permutation gather (index_select), not one scatter per peer.
    force_collective: issue the collective also when the group has a single rank (the 1-rank RCCL dry run of the tests:
    communicator, device buffers, asynchronous work handle -- everything except a second rank)."""

    def __init__(self, height, width, channels=4, band_rows=DEFAULT_BAND_ROWS, dtype=None, device=None, group=None, dst=0, force_collective=False):
        import torch
        import torch.distributed as dist
        self.height, self.band_rows, self.group, self.dst = height, band_rows, group, dst
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.collective = self.world > 1 or (force_collective and dist.is_initialized())
        self.counts = [len(shard_rows(height, band_rows, r, self.world)) for r in range(self.world)]
        self.out_device = torch.device(device) if device is not None else torch.device("cpu")
        dtype = dtype or torch.float32
        # gloo (CPU tests / single-GPU smoke runs) has no device gather: stage through host memory there
        self.stage_on_cpu = self.collective and dist.get_backend(group) == "gloo"
        buf_device = torch.device("cpu") if self.stage_on_cpu else self.out_device
        shape = (max(self.counts), width, channels)
        self.send = torch.zeros(shape, dtype=dtype, device=buf_device)
        root = self.rank == dst
        self.recv_all = torch.empty((self.world,) + shape, dtype=dtype, device=buf_device) if root else None
        self.recv = [self.recv_all[r] for r in range(self.world)] if root else None
        self.index = torch.as_tensor(assembly_index(height, band_rows, self.world), device=buf_device, dtype=torch.long) if root else None
        self.out = torch.empty((height, width, channels), dtype=dtype, device=buf_device) if root else None
        self.work = None
        self.local = None

    def start(self, local):
        import torch.distributed as dist
        assert self.work is None and self.local is None, "finish() the previous gather first"
        assert local.shape[0] == self.counts[self.rank]
        if not self.collective:
            self.local = local
            return
        self.send[: local.shape[0]].copy_(local)  # snapshot: the caller may keep accumulating into `local`
        self.work = dist.gather(self.send, gather_list=self.recv, dst=self.dst, group=self.group, async_op=True)

    def finish(self):
        import torch
        if not self.collective:
            out, self.local = self.local, None
            return out
        if self.work is None:
            return None
        self.work.wait()
        self.work = None
        if self.rank != self.dst:
            return None
        torch.index_select(self.recv_all.view((-1,) + tuple(self.send.shape[1:])), 0, self.index, out=self.out)
        return self.out if self.out.device == self.out_device else self.out.to(self.out_device)


class FrameReducer:
    """Sample-sharded frames: every rank holds a full [height, W, C] accumulation of ITS samples; `start(local)`
    snapshots it and launches the sum-reduction to `dst` asynchronously (RCCL's stream, behind the work already queued on
    the current stream) so that it overlaps the next frame's rendering; `finish()` waits for it and returns the summed
    frame on `dst` (None elsewhere; the returned tensor is reused by the next start()).  At most one reduction in flight."""

    def __init__(self, height, width, channels=4, dtype=None, device=None, group=None, dst=0, force_collective=False):
        import torch
        import torch.distributed as dist
        self.group, self.dst = group, dst
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.collective = self.world > 1 or (force_collective and dist.is_initialized())  # see FrameGatherer
        self.out_device = torch.device(device) if device is not None else torch.device("cpu")
        # gloo (CPU tests / single-GPU smoke runs) reduces host tensors: stage through host memory there
        self.stage_on_cpu = self.collective and dist.get_backend(group) == "gloo"
        buf_device = torch.device("cpu") if self.stage_on_cpu else self.out_device
        self.buf = torch.zeros((height, width, channels), dtype=dtype or torch.float32, device=buf_device)
        self.work = None
        self.local = None

    def start(self, local):
        import torch.distributed as dist
        assert self.work is None and self.local is None, "finish() the previous reduction first"
        assert tuple(local.shape) == tuple(self.buf.shape)
        if not self.collective:
            self.local = local
            return
        self.buf.copy_(local)  # snapshot: the caller may keep accumulating into `local`
        self.work = dist.reduce(self.buf, dst=self.dst, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self):
        if not self.collective:
            out, self.local = self.local, None
            return out
        if self.work is None:
            return None
        self.work.wait()
        self.work = None
        if self.rank != self.dst:
            return None
        return self.buf if self.buf.device == self.out_device else self.buf.to(self.out_device)
