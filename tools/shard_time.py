"""Per-rank time of the multi-GPU bench layout on ONE GPU: rank r of N traces all `spp` samples of its H/N rows
(interleaved 8-row bands, (chunk, sample) work items), consecutive steps on one stream and pipelined over two.
usage: shard_time.py [spp=8] [steps=40]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, brickmap_amd as bm
G, W, H = 1024, 1920, 1080
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
scene = bm.Scene(G, G, device=0).generate().preload_all()
cam = bm.Camera(position=(G/2, G/8, 0.8*G), horizontal_angle=0.8, vertical_angle=-0.5).update()
pool = [torch.cuda.Stream() for _ in range(5)]


def run(st, bufs, N, rank, band, streams):
    flags = bm.BM_FLAG_SAMPLE_ITEMS if N > 1 else 0
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(steps):
        p = bm.FrameParams(W, H, spp=spp, sample_base=i * spp, max_bounces=3, band_rows=band, shard_rank=rank, shard_count=N, flags=flags)
        j = i % len(streams)
        scene.render(cam, p, bufs[j], stream=streams[j].cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / steps * 1e3


base = None
for N in (1, 2, 4, 8):
    rank = 0
    band = bm.dist.DEFAULT_BAND_ROWS if N > 1 else H
    st = bm.State(W, H, device=0, band_rows=band, shard_rank=rank, shard_count=N)
    bufs = [st.blit_buffer, torch.zeros_like(st.blit_buffer)]
    run(st, bufs, N, rank, band, pool[:1])
    one = run(st, bufs, N, rank, band, pool[:1])
    two = min(run(st, bufs, N, rank, band, [pool[a], pool[b]]) for a, b in ((0, 1), (0, 2), (1, 2), (0, 3)))
    if N == 1:
        base = (one, two)
    print(f"N={N} rank {rank} ({st.local_rows} rows x {spp} spp): one stream {one:.3f} ms/step (x{base[0]/one:.2f} of N=1), "
          f"two streams {two:.3f} ms/step (x{base[1]/two:.2f} of N=1 pipelined)")
