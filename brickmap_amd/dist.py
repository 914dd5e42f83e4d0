"""Multi-GPU sharding of the path-trace job: two decompositions, one exchange each, over RCCL/xGMI.

The reference is single-GPU (no collective anywhere).  Every (pixel, sample) pair is an
independent path and every rank holds a full scene replica, so the job can be cut either way:

* by PIXELS -- interleaved row bands (band b belongs to rank b % world; `shard_rows`, `gather_frame`,
  `FrameGatherer`): each rank renders its rows into a packed local float4 buffer with RNG streams keyed on the
  *global* pixel index, so an N-rank image is bit-identical to the 1-rank image; the exchange is one gather of the
  packed bands to rank 0 (grouped ncclSend/ncclRecv, one xGMI link per peer into the root).  This is the
  latency decomposition: one frame finishes N times sooner.
* by SAMPLES -- every rank renders the full frame with its own slice of the sample indices (`sample_base` of
  bm_frame_params; `FrameReducer`): the exchange is one sum-reduction of the float4 frames to rank 0.  This is the
  throughput decomposition (N x the samples per frame): each rank's launch keeps the single-GPU shape -- the
  persistent kernel needs many more pixels than lanes to keep its waves full, and N x spp samples on 1/N of the rows
  cost 1.2x / 1.5x / 2.5x the single-GPU kernel time at N = 2 / 4 / 8 (tools/shard_time.py) -- so `bench.py` uses it.
  The sum of N partial frames differs from one N x spp render only in floating-point association (~1e-7 relative).

No collective sits on the data path of the render itself.
"""
import numpy as np

DEFAULT_BAND_ROWS = 16  # one row of 16x16 tiles
_cache = {}  # receive buffers / row indices, keyed by the gather's shape (per-frame gathers reuse them)


def shard_rows(height, band_rows, rank, world):
    """Global row indices owned by `rank`, in the order they are packed in its local buffer."""
    rows = np.arange(height)
    return rows[(rows // band_rows) % world == rank]


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
            _cache[key] = ([torch.empty_like(send) for _ in range(world)],
                           [torch.as_tensor(shard_rows(height, band_rows, r, world), device=local.device, dtype=torch.long) for r in range(world)],
                           torch.empty((height,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device))
        recv, row_index, out = _cache[key]
        dist.gather(send, gather_list=recv, dst=dst, group=group)
        for r in range(world):
            if row_index[r].numel():
                out.index_copy_(0, row_index[r], recv[r][: counts[r]])
        return out.to(out_device)
    dist.gather(send, gather_list=None, dst=dst, group=group)
    return None


class FrameGatherer:
    """Pipelined form of gather_frame for a render loop: `start(local)` snapshots this rank's packed rows into a send
    buffer and launches the gather asynchronously (on RCCL's stream, behind the work already queued on the current
    stream), so that it overlaps the next frame's rendering; `finish()` waits for it and, on `dst`, returns the
    assembled [height, W, C] frame (None elsewhere; the returned tensor is reused by the next finish()).  At most one
    gather is in flight."""

    def __init__(self, height, width, channels=4, band_rows=DEFAULT_BAND_ROWS, dtype=None, device=None, group=None, dst=0):
        import torch
        import torch.distributed as dist
        self.height, self.band_rows, self.group, self.dst = height, band_rows, group, dst
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.counts = [len(shard_rows(height, band_rows, r, self.world)) for r in range(self.world)]
        self.out_device = torch.device(device) if device is not None else torch.device("cpu")
        dtype = dtype or torch.float32
        # gloo (CPU tests / single-GPU smoke runs) has no device gather: stage through host memory there
        self.stage_on_cpu = self.world > 1 and dist.get_backend(group) == "gloo"
        buf_device = torch.device("cpu") if self.stage_on_cpu else self.out_device
        shape = (max(self.counts), width, channels)
        self.send = torch.zeros(shape, dtype=dtype, device=buf_device)
        self.recv = [torch.empty(shape, dtype=dtype, device=buf_device) for _ in range(self.world)] if self.rank == dst else None
        self.rows = ([torch.as_tensor(shard_rows(height, band_rows, r, self.world), device=buf_device, dtype=torch.long)
                      for r in range(self.world)] if self.rank == dst else None)
        self.out = torch.empty((height, width, channels), dtype=dtype, device=buf_device) if self.rank == dst else None
        self.work = None
        self.local = None

    def start(self, local):
        import torch.distributed as dist
        assert self.work is None and self.local is None, "finish() the previous gather first"
        assert local.shape[0] == self.counts[self.rank]
        if self.world == 1:
            self.local = local
            return
        self.send[: local.shape[0]].copy_(local)  # snapshot: the caller may keep accumulating into `local`
        self.work = dist.gather(self.send, gather_list=self.recv, dst=self.dst, group=self.group, async_op=True)

    def finish(self):
        if self.world == 1:
            out, self.local = self.local, None
            return out
        if self.work is None:
            return None
        self.work.wait()
        self.work = None
        if self.rank != self.dst:
            return None
        for r in range(self.world):
            if self.rows[r].numel():
                self.out.index_copy_(0, self.rows[r], self.recv[r][: self.counts[r]])
        return self.out.to(self.out_device)


class FrameReducer:
    """Sample-sharded frames: every rank holds a full [height, W, C] accumulation of ITS samples; `start(local)`
    snapshots it and launches the sum-reduction to `dst` asynchronously (RCCL's stream, behind the work already queued on
    the current stream) so that it overlaps the next frame's rendering; `finish()` waits for it and returns the summed
    frame on `dst` (None elsewhere; the returned tensor is reused by the next start()).  At most one reduction in flight."""

    def __init__(self, height, width, channels=4, dtype=None, device=None, group=None, dst=0):
        import torch
        import torch.distributed as dist
        self.group, self.dst = group, dst
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.out_device = torch.device(device) if device is not None else torch.device("cpu")
        # gloo (CPU tests / single-GPU smoke runs) reduces host tensors: stage through host memory there
        self.stage_on_cpu = self.world > 1 and dist.get_backend(group) == "gloo"
        buf_device = torch.device("cpu") if self.stage_on_cpu else self.out_device
        self.buf = torch.zeros((height, width, channels), dtype=dtype or torch.float32, device=buf_device)
        self.work = None
        self.local = None

    def start(self, local):
        import torch.distributed as dist
        assert self.work is None and self.local is None, "finish() the previous reduction first"
        assert tuple(local.shape) == tuple(self.buf.shape)
        if self.world == 1:
            self.local = local
            return
        self.buf.copy_(local)  # snapshot: the caller may keep accumulating into `local`
        self.work = dist.reduce(self.buf, dst=self.dst, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self):
        if self.world == 1:
            out, self.local = self.local, None
            return out
        if self.work is None:
            return None
        self.work.wait()
        self.work = None
        if self.rank != self.dst:
            return None
        return self.buf.to(self.out_device)
