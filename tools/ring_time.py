"""The frame ring (bm_render_frames: K consecutive frames as ONE launch) against single launches and against two-stream issue, on one GPU.
usage: python tools/ring_time.py [config2|config2x4|shard8] [frames]
config2: 1080p, 1 spp, 4 segments; config2x4: the north-star shape (4 spp); shard8: rank 0's 1/8 shard (interleaved 8-row bands) of the 8-spp
multi-GPU job.  Prints wall ms per frame for: one launch per frame on one stream, two alternating streams, the ring at several K."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, brickmap_amd as bm
name = sys.argv[1] if len(sys.argv) > 1 else "config2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
W, H, mb, G = 1920, 1080, 3, 1024
spp, shard = {"config2": (1, (0, 1)), "config2x4": (4, (0, 1)), "shard8": (8, (0, 8)), "shard4": (8, (0, 4)), "shard2": (8, (0, 2))}[name]
flags = bm.BM_FLAG_SAMPLE_ITEMS if spp > 1 else 0
scene = bm.Scene(G, G, device=0).generate().preload_all()
cam = bm.Camera(position=(G/2, G/8, 0.8*G), horizontal_angle=0.8, vertical_angle=-0.5).update()
def P(i):
    return bm.FrameParams(W, H, spp=spp, sample_base=i * spp, max_bounces=mb, flags=flags, band_rows=8 if shard[1] > 1 else 0, shard_rank=shard[0], shard_count=shard[1])
rows = bm.local_rows(P(0))
acc = torch.zeros((rows, W, 4), dtype=torch.float32, device="cuda")
acc2 = torch.zeros_like(acc)
def timed(fn, reps=3):
    best = None
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / n * 1e3
        best = dt if best is None else min(best, dt)
    return best
def single():
    for i in range(n): scene.render(cam, P(i), acc)
single(); torch.cuda.synchronize()
t_single = timed(single)
k_single = float(np.median(scene.render_times(n)))
s2 = [torch.cuda.Stream(), torch.cuda.Stream()]
def two():
    for i in range(n): scene.render(cam, P(i), (acc, acc2)[i % 2], stream=s2[i % 2].cuda_stream)
two()
t_two = timed(two)
print(f"{name}: {rows} rows x {W}, {spp} spp.  single launches: {t_single:.4f} ms per frame (kernel median {k_single:.4f}); two streams: {t_two:.4f}")
for K in (2, 4, 8, 20, 40):
    if K > n: break
    def ring():
        for b in range(0, n, K): scene.render_frames(cam, [P(b + i) for i in range(min(K, n - b))], acc)
    ring()
    print(f"  ring K = {K:3d}: {timed(ring):.4f} ms per frame")
