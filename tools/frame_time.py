"""Kernel time of the config-2 frame (1080p, 1 spp, 4 segments, 1024^3 resident) on one stream: median of 30 launches.
usage: python tools/frame_time.py [workload=config2]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, brickmap_amd as bm
G, W, H = 1024, 1920, 1080
scene = bm.Scene(G, G, device=0).generate().preload_all()
cam = bm.Camera(position=(G/2, G/8, 0.8*G), horizontal_angle=0.8, vertical_angle=-0.5).update()
acc = torch.zeros((H, W, 4), dtype=torch.float32, device='cuda')
for i in range(36):
    scene.render(cam, bm.FrameParams(W, H, spp=1, sample_base=i, max_bounces=3), acc)
torch.cuda.synchronize()
ms = scene.render_times(30)
print(f"config2 frame: median {np.median(ms):.4f} ms, mean {ms.mean():.4f}, min {ms.min():.4f}")
