"""Per-rank kernel time of the weak-scaling bench layout on ONE GPU: rank r of N traces N*spp samples for its H/N rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, brickmap_amd as bm
G, W, H = 1024, 1920, 1080
scene = bm.Scene(G, G, device=0).generate().preload_all()
cam = bm.Camera(position=(G/2, G/8, 0.8*G), horizontal_angle=0.8, vertical_angle=-0.5).update()
flags = bm.BM_FLAG_SAMPLE_ITEMS if (len(sys.argv) > 1 and sys.argv[1] == "items") else 0
print("work items:", "(chunk, sample)" if flags else "pixels")
for N in (1, 2, 4, 8):
    for rank in sorted({0, N - 1}):
        band = 16 if N > 1 else H
        st = bm.State(W, H, device=0, band_rows=band, shard_rank=rank, shard_count=N)
        ts = []
        for i in range(8):
            p = bm.FrameParams(W, H, spp=N, sample_base=i * N, max_bounces=3, band_rows=band, shard_rank=rank, shard_count=N, flags=flags)
            scene.render(cam, p, st.blit_buffer)
            ts.append(scene.last_render_ms())
        print(f"N={N} rank {rank}: {np.median(ts[2:]):.3f} ms per step ({st.local_rows} rows x {N} spp)")
