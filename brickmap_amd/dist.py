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

The exchange itself lives behind the C-ABI (include/brickmap.h "multi-GPU": bm_comm_create / bm_gather_frame / bm_reduce_frame,
csrc/comm.hip -- grouped ncclSend / ncclRecv into one stacked buffer + one assembly kernel): `Comm` wraps it, and FrameGatherer /
FrameReducer use it whenever the process group's backend is RCCL ("nccl") and the frames are device tensors.  torch.distributed
then only carries the 128-byte communicator id from rank 0 to the others.  The torch.distributed.gather / reduce path remains for
gloo (the CPU tests) and can be forced with BM_DIST_TORCH=1.
"""
import ctypes as C
import os

import numpy as np

# rows per band: 1080 rows are 135 bands of 8 -- 17 or 16 per rank at N = 8 -- and every rank sees every part of the image; with 16-row
# bands the slowest of eight ranks takes 1.280 ms per step against 1.232 (the job is as fast as its slowest rank), with 4-row bands 1.243
DEFAULT_BAND_ROWS = 8
_cache = {}  # receive buffers / row indices, keyed by the gather's shape (per-frame gathers reuse them)
last_comm_error = None  # why _make_comm last gave up on the C-ABI exchange (this rank's own error text, or that another rank failed)


class Comm:
    """bm_comm (include/brickmap.h): an RCCL communicator of the C-ABI, one rank per GPU.  `unique_id` is the 128 bytes of
    bm_comm_unique_id made by ONE rank and carried to the others by the caller."""

    def __init__(self, device, rank, world, unique_id):
        from . import _lib
        self._L = _lib.load()
        self._check = _lib.check
        self.device, self.rank, self.world = int(device), int(rank), int(world)
        assert len(unique_id) == 128
        self._id = (C.c_ubyte * 128).from_buffer_copy(bytes(unique_id))
        h = C.c_void_p()
        self._check(self._L.bm_comm_create(self.device, self.rank, self.world, self._id, C.byref(h)))
        self.handle = h

    @staticmethod
    def unique_id():
        from . import _lib
        buf = (C.c_ubyte * 128)()
        _lib.check(_lib.load().bm_comm_unique_id(buf))
        return bytes(buf)

    @classmethod
    def from_process_group(cls, device, group=None):
        """One communicator over the ranks of a torch.distributed group: rank 0 makes the id, a broadcast carries it."""
        import torch
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        on_device = dist.get_backend(group) == "nccl"
        t = torch.zeros(128, dtype=torch.uint8, device=torch.device("cuda", device) if on_device else torch.device("cpu"))
        if rank == 0:
            t.copy_(torch.frombuffer(bytearray(cls.unique_id()), dtype=torch.uint8))
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(device, rank, world, bytes(t.cpu().numpy().tobytes()))

    def gather_frame(self, packed, frame, height, width, band_rows, root=0, stream=None):
        """bm_gather_frame: `packed` = this rank's [local_rows, width, 4] float32 device tensor, `frame` = [height, width, 4] on the root."""
        import torch
        if stream is None:
            stream = torch.cuda.current_stream(packed.device).cuda_stream
        self._check(self._L.bm_gather_frame(self.handle, C.c_void_p(packed.data_ptr()), C.c_void_p(frame.data_ptr()) if frame is not None else None,
                                            int(height), int(width), int(band_rows), int(root), C.c_void_p(stream)))

    def gather_frames(self, packed, frames, count, height, width, band_rows, root=0, stream=None):
        """bm_gather_frames: a batch -- `packed` = [count, local_rows, width, 4] (one allocation, what ONE bm_render_frames launch filled),
        `frames` = [count, height, width, 4] on the root; one message per peer for the whole batch."""
        import torch
        if stream is None:
            stream = torch.cuda.current_stream(packed.device).cuda_stream
        self._check(self._L.bm_gather_frames(self.handle, C.c_void_p(packed.data_ptr()), C.c_void_p(frames.data_ptr()) if frames is not None else None,
                                             int(count), int(height), int(width), int(band_rows), int(root), C.c_void_p(stream)))

    def reduce_frame(self, src, dst, root=0, stream=None):
        import torch
        if stream is None:
            stream = torch.cuda.current_stream(src.device).cuda_stream
        self._check(self._L.bm_reduce_frame(self.handle, C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()) if dst is not None else None,
                                            int(src.numel()), int(root), C.c_void_p(stream)))

    def info(self):
        """bm_comm_info: (rank, world) as the communicator itself reports them (ncclCommUserRank / ncclCommCount)."""
        r, w = C.c_int(-1), C.c_int(-1)
        self._check(self._L.bm_comm_info(self.handle, C.byref(r), C.byref(w)))
        return int(r.value), int(w.value)

    def barrier(self, stream=None):
        import torch
        self._check(self._L.bm_comm_barrier(self.handle, C.c_void_p(stream if stream is not None else torch.cuda.current_stream(self.device).cuda_stream)))

    def selftest(self, stream=None):
        import torch
        self._check(self._L.bm_comm_selftest(self.handle, C.c_void_p(stream if stream is not None else torch.cuda.current_stream(self.device).cuda_stream)))

    def close(self):
        if getattr(self, "handle", None):
            self._L.bm_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _make_comm(device_index, group):
    """The C-ABI communicator for a process group, checked before it is relied on: every rank runs bm_comm_selftest (a grouped
    send / receive round the ring of ranks + an all-reduce, data verified) and the ranks agree on the outcome -- if the
    communicator cannot be made or the check fails on ANY rank, all of them fall back to torch.distributed's exchange."""
    import sys
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", device_index) if dist.get_backend(group) == "nccl" else torch.device("cpu")

    def agreed(ok):
        t = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        return int(t.item()) == 1

    # 1. can every rank bind the library at all?  (A rank that failed later, inside from_process_group, would leave the others
    #    waiting in the id's broadcast or in ncclCommInitRank: settle it while nobody depends on anybody yet.)
    global last_comm_error
    last_comm_error = None
    why = None
    try:
        from . import _lib
        if not _lib.load().bm_comm_available():  # binds the library, starts nothing (ncclGetUniqueId would start a bootstrap listener per call)
            why = "RCCL library (librccl.so.1) not found"
    except Exception as e:  # noqa: BLE001
        why = e
    if not agreed(0 if why else 1):
        last_comm_error = str(why) if why is not None else "another rank could not bind the RCCL library"
        if why is not None:
            print(f"brickmap_amd.dist: C-ABI RCCL exchange unavailable ({why}); using torch.distributed", file=sys.stderr)
        return None
    # 2. the communicator, then its self-test
    comm, ok = None, 1
    try:
        comm = Comm.from_process_group(device_index, group)
        comm.selftest()
    except Exception as e:  # noqa: BLE001 -- anything at all: the render must not depend on it
        ok = 0
        last_comm_error = str(e)
        print(f"brickmap_amd.dist: C-ABI RCCL exchange unavailable ({e}); using torch.distributed", file=sys.stderr)
    if not agreed(ok):
        if last_comm_error is None:
            last_comm_error = "another rank's bm_comm_create / bm_comm_selftest failed"
        if comm is not None:
            comm.close()
        return None
    return comm


def _use_capi(collective, group, device):
    """The C-ABI exchange serves RCCL groups with device frames; gloo (CPU tests) and BM_DIST_TORCH=1 keep torch.distributed's."""
    import torch
    import torch.distributed as dist
    if not (collective and dist.is_initialized() and torch.device(device).type == "cuda") or os.environ.get("BM_DIST_TORCH", "0") == "1":
        return False
    # BM_DIST_CAPI=1: also on a gloo group (the tests' shared-GPU runs, with BM_RCCL_LIBRARY naming a stand-in transport)
    return dist.get_backend(group) == "nccl" or os.environ.get("BM_DIST_CAPI", "0") == "1"


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
    frames > 1: the unit of exchange is a BATCH of that many frames -- what a rank rendered with one Scene.render_frames launch
    (the frame ring): start() takes [frames, local_rows, W, C] (or a list of that many [local_rows, W, C] tensors), every peer sends
    its whole batch as ONE message (bm_gather_frames), finish() returns [frames, height, W, C].
    force_collective: issue the collective also when the group has a single rank (the 1-rank RCCL dry run of the tests:
    communicator, device buffers, asynchronous work handle -- everything except a second rank)."""

    def __init__(self, height, width, channels=4, band_rows=DEFAULT_BAND_ROWS, dtype=None, device=None, group=None, dst=0, force_collective=False,
                 side_stream=None, frames=1):
        import torch
        import torch.distributed as dist
        self.height, self.band_rows, self.group, self.dst = height, band_rows, group, dst
        self.frames = int(frames)
        assert 1 <= self.frames <= 256
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.collective = self.world > 1 or (force_collective and dist.is_initialized())
        self.counts = [len(shard_rows(height, band_rows, r, self.world)) for r in range(self.world)]
        self.out_device = torch.device(device) if device is not None else torch.device("cpu")
        dtype = dtype or torch.float32
        # the exchange behind the C-ABI (bm_gather_frames) where it applies: it runs on a side stream, behind a snapshot of the packed
        # rows and beside the next frame.  (Made first: the buffers below live on the device with it, on the host for gloo.)
        self.comm = None
        if _use_capi(self.collective, group, self.out_device) and channels == 4 and dtype == torch.float32:
            self.comm = _make_comm(self.out_device.index if self.out_device.index is not None else torch.cuda.current_device(), group)
        # gloo (CPU tests / single-GPU smoke runs) has no device gather: stage through host memory there
        self.stage_on_cpu = self.collective and self.comm is None and dist.get_backend(group) == "gloo"
        buf_device = torch.device("cpu") if self.stage_on_cpu else self.out_device
        torch_path = self.comm is None  # (with the C-ABI exchange the root's receive buffer belongs to the communicator)
        max_rows = max(self.counts)
        # the batch as it is sent: tightly packed for the C-ABI exchange (one message of frames x local_rows rows), padded to the
        # largest shard for torch.distributed's equal-sized gather
        self.send = torch.zeros((self.frames, max_rows if torch_path else self.counts[self.rank], width, channels), dtype=dtype, device=buf_device)
        root = self.rank == dst
        self.recv_all = torch.empty((self.world,) + tuple(self.send.shape), dtype=dtype, device=buf_device) if root and torch_path else None
        self.recv = [self.recv_all[r] for r in range(self.world)] if root and torch_path else None
        self.index = None
        if root and torch_path:  # frame k, row y <- stacked row (its rank) * frames * max_rows + k * max_rows + (its packed row)
            base = assembly_index(height, band_rows, self.world)
            r_of, l_of = base // max_rows, base % max_rows
            idx = np.concatenate([r_of * self.frames * max_rows + k * max_rows + l_of for k in range(self.frames)])
            self.index = torch.as_tensor(idx, device=buf_device, dtype=torch.long)
        self.out = torch.empty((self.frames, height, width, channels), dtype=dtype, device=buf_device) if root else None
        self.work = None
        self.local = None
        self.width = width
        if self.comm is not None:
            # side_stream: HIP maps streams onto a few hardware queues, and a queue runs its packets in order -- a gather that waits for
            # its frame at the head of a queue holds up whatever another stream put behind it there (the NEXT frame, if the render
            # stream shares that queue).  A caller that pipelines frames over several streams passes a stream it has probed to run
            # beside them (bm_probe_streams).
            self.side = side_stream if side_stream is not None else torch.cuda.Stream(device=self.out_device)
            self.snap = torch.cuda.Event()
            self.pending = False

    def _snapshot(self, local):
        """copy the caller's packed rows (one frame, a batch tensor, or a list of frames) into the send buffer"""
        rows = self.counts[self.rank]
        if isinstance(local, (list, tuple)):
            assert len(local) == self.frames
            for k, t in enumerate(local):
                assert t.shape[0] == rows
                self.send[k, :rows].copy_(t)
        elif local.dim() == self.send.dim():
            assert local.shape[0] == self.frames and local.shape[1] == rows
            self.send[:, :rows].copy_(local)
        else:
            assert self.frames == 1 and local.shape[0] == rows
            self.send[0, :rows].copy_(local)

    def _result(self, out):
        return out if self.frames > 1 else out[0]

    def start(self, local):
        import torch
        import torch.distributed as dist
        assert self.work is None and self.local is None, "finish() the previous gather first"
        if not self.collective:
            self.local = torch.stack(list(local)) if isinstance(local, (list, tuple)) else local
            return
        if self.comm is not None:
            assert not self.pending, "finish() the previous gather first"
            cur = torch.cuda.current_stream(self.out_device)
            cur.wait_stream(self.side)  # the previous gather has read `send` (whichever stream its finish() was called on)
            self._snapshot(local)  # the caller may keep accumulating into `local`
            self.snap.record(cur)
            self.side.wait_event(self.snap)
            self.comm.gather_frames(self.send, self.out, self.frames, self.height, self.width, self.band_rows, root=self.dst, stream=self.side.cuda_stream)
            self.pending = True
            return
        self._snapshot(local)  # the caller may keep accumulating into `local`
        self.work = dist.gather(self.send, gather_list=self.recv, dst=self.dst, group=self.group, async_op=True)

    def finish(self, wait=True):
        """wait=False: the caller does not touch the frame on the current stream (a render loop that only starts the next gather:
        start() orders itself behind this one) -- no stream is made to wait."""
        import torch
        if not self.collective:
            out, self.local = self.local, None
            return out
        if self.comm is not None:
            if not self.pending:
                return None
            if wait:
                torch.cuda.current_stream(self.out_device).wait_stream(self.side)  # like Work.wait(): the current stream, not the host
            self.pending = False
            return self._result(self.out) if self.rank == self.dst else None
        if self.work is None:
            return None
        self.work.wait()
        self.work = None
        if self.rank != self.dst:
            return None
        torch.index_select(self.recv_all.view((-1,) + tuple(self.send.shape[2:])), 0, self.index, out=self.out.view((-1,) + tuple(self.send.shape[2:])))
        out = self._result(self.out)
        return out if out.device == self.out_device else out.to(self.out_device)


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
        self.comm = None
        if _use_capi(self.collective, group, self.out_device) and (dtype or torch.float32) == torch.float32:
            self.comm = _make_comm(self.out_device.index if self.out_device.index is not None else torch.cuda.current_device(), group)
        # gloo (CPU tests / single-GPU smoke runs) reduces host tensors: stage through host memory there
        self.stage_on_cpu = self.collective and self.comm is None and dist.get_backend(group) == "gloo"
        buf_device = torch.device("cpu") if self.stage_on_cpu else self.out_device
        self.buf = torch.zeros((height, width, channels), dtype=dtype or torch.float32, device=buf_device)
        self.work = None
        self.local = None
        if self.comm is not None:
            self.side = torch.cuda.Stream(device=self.out_device)
            self.snap = torch.cuda.Event()
            self.pending = False

    def start(self, local):
        import torch
        import torch.distributed as dist
        assert self.work is None and self.local is None, "finish() the previous reduction first"
        assert tuple(local.shape) == tuple(self.buf.shape)
        if not self.collective:
            self.local = local
            return
        if self.comm is not None:
            assert not self.pending, "finish() the previous reduction first"
            cur = torch.cuda.current_stream(self.out_device)
            cur.wait_stream(self.side)  # the previous reduction has read / written `buf` (whichever stream its finish() was called on)
            self.buf.copy_(local)  # snapshot: the caller may keep accumulating into `local`
            self.snap.record(cur)
            self.side.wait_event(self.snap)
            self.comm.reduce_frame(self.buf, self.buf, root=self.dst, stream=self.side.cuda_stream)  # in place, like dist.reduce
            self.pending = True
            return
        self.buf.copy_(local)  # snapshot: the caller may keep accumulating into `local`
        self.work = dist.reduce(self.buf, dst=self.dst, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self):
        if not self.collective:
            out, self.local = self.local, None
            return out
        if self.comm is not None:
            import torch
            if not self.pending:
                return None
            torch.cuda.current_stream(self.out_device).wait_stream(self.side)
            self.pending = False
            return self.buf if self.rank == self.dst else None
        if self.work is None:
            return None
        self.work.wait()
        self.work = None
        if self.rank != self.dst:
            return None
        return self.buf if self.buf.device == self.out_device else self.buf.to(self.out_device)
