"""Kernel time of a bench workload's frame on one stream (median of the timed launches): the A/B yardstick.
usage: python tools/frame_time.py [config2|config3|config5] [launches]
config2: 1080p, 1 spp, 4 segments, 1024^3 resident; config3: 4K, 4 spp, 8 segments, 2048^3 (resident here: the kernel, not the
streaming); config5: 8K, 4 spp, 8 segments, LoD, 4096^3 resident."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, brickmap_amd as bm
name = sys.argv[1] if len(sys.argv) > 1 else "config2"
W, H, spp, mb, ns = {"config2": (1920, 1080, 1, 3, 8), "config2x4": (1920, 1080, 4, 3, 8), "config3": (3840, 2160, 4, 7, 16), "config5": (7680, 4320, 4, 7, 32)}[name]
n = int(sys.argv[2]) if len(sys.argv) > 2 else (30 if name.startswith("config2") else 4)
G = 128 * ns
scene = bm.Scene(G, G, device=0).generate().preload_all()
cam = bm.Camera(position=(G/2, G/8, 0.8*G), horizontal_angle=0.8, vertical_angle=-0.5).update()
acc = torch.zeros((H, W, 4), dtype=torch.float32, device='cuda')
for i in range(n + (6 if name.startswith("config2") else 1)):
    scene.render(cam, bm.FrameParams(W, H, spp=spp, sample_base=i * spp, max_bounces=mb), acc)
torch.cuda.synchronize()
ms = scene.render_times(n)
print(f"{name} frame: median {np.median(ms):.4f} ms, mean {ms.mean():.4f}, min {ms.min():.4f}  ({W*H*spp*(mb+1)/np.median(ms)/1e3:.0f} nominal Mrays/s)")
