"""One rank of the multi-rank C-ABI exchange test (tests/test_gpu_dist.py): several of these share GPU 0 over the host-staged
stand-in for RCCL (tests/fake_rccl.cpp, BM_RCCL_LIBRARY).  usage: _fake_rccl_worker.py <rank> <world> <id_file> <root>"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import brickmap_amd as bm  # noqa: E402


def main():
    rank, world, id_file, root = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    torch.cuda.set_device(0)
    if rank == 0:
        uid = bm.dist.Comm.unique_id()
        with open(id_file + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(id_file + ".tmp", id_file)
    else:
        t0 = time.time()
        while not os.path.exists(id_file):
            assert time.time() - t0 < 60, "no communicator id"
            time.sleep(0.02)
        uid = open(id_file, "rb").read()
    comm = bm.dist.Comm(0, rank, world, uid)
    comm.selftest()   # ring of grouped send / receive + all-reduce through the C-ABI, data verified on every rank
    comm.barrier()
    G, W, H, band = 256, 200, 120, 16  # 120 rows: 7 bands of 16 + a ragged one -> unequal shards
    scene = bm.Scene(G, G, device=0).generate().preload_all()
    cam = bm.Camera(position=(G / 2, G / 8, 0.8 * G), horizontal_angle=0.8, vertical_angle=-0.5).update()
    state = bm.State(W, H, device=0, band_rows=band, shard_rank=rank, shard_count=world)
    p = bm.FrameParams(W, H, spp=2, max_bounces=3, band_rows=band, shard_rank=rank, shard_count=world, flags=bm.BM_FLAG_ORDERED)  # ordered sums: the gathered frame is compared bit for bit
    frame = torch.full((H, W, 4), float("nan"), dtype=torch.float32, device="cuda:0") if rank == root else None
    for step in range(2):  # two frames through the same communicator (the root's stacked buffer is reused)
        scene.render(cam, p, state.blit_buffer)
        comm.gather_frame(state.blit_buffer, frame, H, W, band, root=root)
    torch.cuda.synchronize()
    if rank == root:
        want = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
        for step in range(2):
            scene.render(cam, bm.FrameParams(W, H, spp=2, max_bounces=3, flags=bm.BM_FLAG_ORDERED), want)
        torch.cuda.synchronize()
        assert torch.equal(frame.view(torch.int32), want.view(torch.int32)), "gathered frame differs from the unsharded render"
    # a batch: three frames (own sample_base each) rendered by ONE frame-ring launch into one allocation, exchanged as one message
    # per peer (bm_gather_frames) -- each gathered frame is the unsharded frame bit for bit
    K = 3
    rows = state.local_rows
    packed = torch.zeros((K, rows, W, 4), dtype=torch.float32, device="cuda:0")
    ps = [bm.FrameParams(W, H, spp=2, sample_base=10 + 2 * k, max_bounces=3, band_rows=band, shard_rank=rank, shard_count=world, flags=bm.BM_FLAG_ORDERED) for k in range(K)]
    scene.render_frames(cam, ps, [packed[k] for k in range(K)])
    frames = torch.full((K, H, W, 4), float("nan"), dtype=torch.float32, device="cuda:0") if rank == root else None
    comm.gather_frames(packed, frames, K, H, W, band, root=root)
    torch.cuda.synchronize()
    if rank == root:
        for k in range(K):
            want = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
            scene.render(cam, bm.FrameParams(W, H, spp=2, sample_base=10 + 2 * k, max_bounces=3, flags=bm.BM_FLAG_ORDERED), want)
            torch.cuda.synchronize()
            assert torch.equal(frames[k].view(torch.int32), want.view(torch.int32)), f"frame {k} of the gathered batch differs from the unsharded render"
    # the sample decomposition's exchange: every rank contributes rank + 1
    src = torch.full((H, W, 4), float(rank + 1), dtype=torch.float32, device="cuda:0")
    dst = torch.zeros_like(src) if rank == root else None
    comm.reduce_frame(src, dst, root=root)
    torch.cuda.synchronize()
    if rank == root:
        assert float(dst.min()) == float(dst.max()) == world * (world + 1) / 2
    comm.barrier()
    comm.close()
    scene.close()
    print(f"FAKE_RCCL_RANK_OK {rank}", flush=True)


if __name__ == "__main__":
    main()
