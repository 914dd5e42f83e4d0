"""Every rank's shard of the north-star job (1080p, 8 spp, (chunk, sample) items, N = 8) timed on ONE GPU for several band heights:
the job is as fast as its slowest rank (profiles/r04_rank_balance.txt).  usage: python tools/rank_times.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, brickmap_amd as bm
G, W, H, spp, steps = 1024, 1920, 1080, 8, 30
scene = bm.Scene(G, G, device=0).generate().preload_all()
cam = bm.Camera(position=(G/2, G/8, 0.8*G), horizontal_angle=0.8, vertical_angle=-0.5).update()
pool = [torch.cuda.Stream() for _ in range(1)]
def run(N, rank, band, streams):
    st = bm.State(W, H, device=0, band_rows=band, shard_rank=rank, shard_count=N)
    bufs = [st.blit_buffer, torch.zeros_like(st.blit_buffer)]
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        for i in range(steps):
            p = bm.FrameParams(W, H, spp=spp, sample_base=i * spp, max_bounces=3, band_rows=band, shard_rank=rank, shard_count=N, flags=bm.BM_FLAG_SAMPLE_ITEMS)
            j = i % len(streams)
            scene.render(cam, p, bufs[j], stream=streams[j].cuda_stream)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / steps * 1e3)
    return best
for N in (8,):
    for band in (4, 5, 8, 9, 12, 15, 16, 4, 8, 16):
        ts = [run(N, r, band, pool) for r in range(N)]
        print(f"N={N} band_rows={band}: two streams per rank " + " ".join(f"{t:.3f}" for t in ts) + f"  max {max(ts):.3f} mean {np.mean(ts):.3f}", flush=True)
