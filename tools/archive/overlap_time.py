"""Consecutive frames issued on two alternating streams: the next frame's workgroups start while the previous frame drains.
Accumulation with float atomics (BM_FLAG_SAMPLE_ITEMS), because two frames may touch a pixel at the same time.
usage: python tools/overlap_time.py [streams]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, brickmap_amd as bm
G, W, H = 1024, 1920, 1080
scene = bm.Scene(G, G, device=0).generate().preload_all()
cam = bm.Camera(position=(G/2, G/8, 0.8*G), horizontal_angle=0.8, vertical_angle=-0.5).update()
acc = torch.zeros((H, W, 4), dtype=torch.float32, device='cuda')
for n_streams in (1, 2, 3):
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    for flags in (0, bm.BM_FLAG_SAMPLE_ITEMS):
        if n_streams > 1 and not flags:
            continue
        K = 40
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(K):
                st = streams[i % n_streams]
                scene.render(cam, bm.FrameParams(W, H, spp=1, sample_base=i, max_bounces=3, flags=flags), acc, stream=st.cuda_stream)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / K
        ms = scene.render_times(20)
        print(f"{n_streams} stream(s), {'atomic' if flags else 'plain '} accumulation: {dt*1e3:.3f} ms per step, kernel events avg {ms.mean():.3f} ms -> {W*H*4/dt/1e6:.0f} Mrays/s nominal")
