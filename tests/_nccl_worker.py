"""1-rank RCCL dry run (tests/test_gpu_dist.py): everything bench.py --gpus N does on the "nccl" backend except the second
rank -- process group with device_id, device-tensor collectives with asynchronous work handles (FrameGatherer /
FrameReducer with force_collective), barrier, the max-over-ranks all_reduce -- on one frame of the HIP path."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import brickmap_amd as bm  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    assert dist.get_backend() == "nccl"
    G, W, H = 256, 200, 120  # 120 rows = 7 full bands of 16 + a ragged one
    scene = bm.Scene(G, G, device=0).generate().preload_all()
    cam = bm.Camera(position=(G / 2, G / 8, 0.8 * G), horizontal_angle=0.8, vertical_angle=-0.5).update()
    band = bm.dist.DEFAULT_BAND_ROWS
    state = bm.State(W, H, device=0, band_rows=band, shard_rank=0, shard_count=1)
    p = bm.FrameParams(W, H, spp=2, max_bounces=3, flags=bm.BM_FLAG_SAMPLE_ITEMS, band_rows=band, shard_rank=0, shard_count=1)
    gatherer = bm.dist.FrameGatherer(H, W, band_rows=band, device=dev, force_collective=True)
    # RCCL group + device frames: the exchange is the C-ABI's (bm_comm_create / bm_gather_frame, csrc/comm.hip)
    assert gatherer.collective and not gatherer.stage_on_cpu and gatherer.send.is_cuda and gatherer.comm is not None
    assert (gatherer.comm.rank, gatherer.comm.world) == (0, 1)
    gatherer.comm.selftest()  # grouped ncclSend / ncclRecv (to itself) + all-reduce through the communicator, data verified
    gatherer.comm.barrier()
    frames = []
    for step in range(3):  # the pipelined loop of bench.py: finish the previous gather, start the next one behind the frame
        scene.render(cam, p, state.blit_buffer)
        got = gatherer.finish()
        if got is not None:
            frames.append(got.clone())
        gatherer.start(state.blit_buffer)
    frames.append(gatherer.finish().clone())
    torch.cuda.synchronize()
    assert len(frames) == 3 and all(f.is_cuda and tuple(f.shape) == (H, W, 4) for f in frames)
    assert torch.equal(frames[-1], state.blit_buffer)  # world 1: the gathered frame is the local one, bit for bit
    assert float(frames[0][..., 3].min()) == 2.0 and float(frames[2][..., 3].min()) == 6.0
    reducer = bm.dist.FrameReducer(H, W, device=dev, force_collective=True)
    assert reducer.collective and reducer.buf.is_cuda and reducer.comm is not None
    reducer.start(state.blit_buffer)
    red = reducer.finish()
    torch.cuda.synchronize()
    assert torch.equal(red, state.blit_buffer)
    dist.barrier()
    t = torch.tensor([1.25], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t.item()) == 1.25
    # the torch.distributed exchange stays available (gloo tests, BM_DIST_TORCH=1): same frames
    os.environ["BM_DIST_TORCH"] = "1"
    g2 = bm.dist.FrameGatherer(H, W, band_rows=band, device=dev, force_collective=True)
    assert g2.comm is None and g2.recv_all.is_cuda
    g2.start(state.blit_buffer)
    assert torch.equal(g2.finish(), state.blit_buffer)
    del os.environ["BM_DIST_TORCH"]
    # a second communicator from raw C-ABI calls, as a C++ host would make it (no torch.distributed involved)
    c2 = bm.dist.Comm(0, 0, 1, bm.dist.Comm.unique_id())
    out = torch.empty_like(state.blit_buffer)
    c2.gather_frame(state.blit_buffer, out, H, W, band)
    torch.cuda.synchronize()
    assert torch.equal(out, state.blit_buffer)
    c2.close()
    one = bm.dist.gather_frame(state.blit_buffer, H, band)  # world 1 short-cut returns the local frame
    assert one is state.blit_buffer
    scene.close()
    dist.destroy_process_group()
    print("NCCL_DRY_RUN_OK", np.float32(frames[-1].sum().item()))


if __name__ == "__main__":
    main()
