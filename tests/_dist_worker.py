"""Worker for tests/test_dist_gloo.py: one process per rank, gloo backend, CPU tensors.

The sharding / gather logic (brickmap_amd/dist.py) is backend-agnostic; the per-rank renderer here
is the CPU oracle standing in for the HIP kernel (allowed in tests), so that the N-rank frame can be
checked against the 1-rank frame without a GPU."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from brickmap_amd import dist as bdist  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    W, H, band = 40, 70, 16  # ragged: 70 rows = 4 full bands + a 6-row tail
    w = oracle.World(128, 128, threads=2)
    w.reset_device(True)
    cam = oracle.make_camera((64, 16, 102.4), oracle.camera_direction(0.8, -0.5))
    full = np.zeros((H, W, 4), np.float32)
    w.render(cam, oracle.make_frame(W, H, spp=2, band_rows=band, shard_rank=rank, shard_count=world), accum=full, want_dbg=False)
    rows = bdist.shard_rows(H, band, rank, world)
    local = torch.from_numpy(np.ascontiguousarray(full[rows]))
    out = bdist.gather_frame(local, H, band, dst=0)
    if rank == 0:
        want, _, _, _ = w.render(cam, oracle.make_frame(W, H, spp=2), want_dbg=False)
        assert out is not None and tuple(out.shape) == (H, W, 4)
        assert np.array_equal(out.numpy(), want), "gathered N-rank frame differs from the 1-rank frame"
        print("DIST_OK", world)
    else:
        assert out is None
    # pipelined form: two frames through the same gatherer, the second started before the first is read back
    g = bdist.FrameGatherer(H, W, band_rows=band, device="cpu")
    g.start(local)
    first = g.finish()
    first = first.clone() if first is not None else None  # the returned frame is only valid until the next finish()
    g.start(local * 2)
    second = g.finish()
    if rank == 0:
        assert np.array_equal(first.numpy(), want) and np.array_equal(second.numpy(), want * 2)
        print("PIPE_OK", world)
    else:
        assert first is None and second is None
    # batches (what a rank renders with one frame-ring launch): three frames per exchange, as one tensor and as a list of frames
    gb = bdist.FrameGatherer(H, W, band_rows=band, device="cpu", frames=3)
    batch = torch.stack([local, local * 3, local * 5])
    gb.start(batch)
    got_b = gb.finish()
    got_b = got_b.clone() if got_b is not None else None
    gb.start([local * 7, local, local * 2])
    got_l = gb.finish()
    if rank == 0:
        assert tuple(got_b.shape) == (3, H, W, 4)
        for k, f in enumerate((1, 3, 5)):
            assert np.array_equal(got_b[k].numpy(), want * f)
        for k, f in enumerate((7, 1, 2)):
            assert np.array_equal(got_l[k].numpy(), want * f)
        print("BATCH_OK", world)
    else:
        assert got_b is None and got_l is None
    # sample sharding: every rank renders the full frame with its own sample indices; the frames are summed on rank 0
    mine = np.zeros((H, W, 4), np.float32)
    w.render(cam, oracle.make_frame(W, H, spp=2, sample_base=2 * rank), accum=mine, want_dbg=False)
    red = bdist.FrameReducer(H, W, device="cpu")
    red.start(torch.from_numpy(mine))
    total = red.finish()
    if rank == 0:
        want_all, _, _, _ = w.render(cam, oracle.make_frame(W, H, spp=2 * world), want_dbg=False)
        got = total.numpy()
        assert np.array_equal(got[..., 3], want_all[..., 3])  # terminated-path counts are integers: exact
        assert np.allclose(got, want_all, rtol=1e-5, atol=1e-7), "sum of the sample shards differs from the single render"
        print("SAMPLES_OK", world)
    else:
        assert total is None
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
