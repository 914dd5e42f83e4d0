"""Multi-GPU sharding of the frame: interleaved row bands + one gather over RCCL/xGMI.

The reference is single-GPU (no collective anywhere).  Pixels are independent, so
the frame is split into row bands dealt round-robin to the ranks (band b belongs to rank
b % world): each rank holds a full scene replica, renders its rows into a packed local float4
buffer with RNG streams keyed on the *global* pixel index (so an N-rank image is bit-identical
to the 1-rank image), and the only exchange is one gather of the packed bands to rank 0
(`torch.distributed.gather` = grouped ncclSend/ncclRecv on the nccl/RCCL backend, one xGMI
link per peer into the root).  No collective sits on the data path of the render itself.
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
