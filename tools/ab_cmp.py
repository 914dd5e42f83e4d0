# python tools/ab_cmp.py <lib.so>: renders a 1080p frame on config2's world with <lib> and prints kernel ms + a hash of the ordered frame
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from brickmap_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import numpy as np, brickmap_amd as bm
for name, (W, H, spp, mb, ns, lod) in {"config2": (1920, 1080, 1, 3, 8, None), "config5": (7680, 4320, 4, 7, 32, None)}.items():
    if len(sys.argv) > 2 and name not in sys.argv[2:]: continue
    G = 128 * ns
    scene = bm.Scene(G, G, device=0).generate().preload_all()
    cam = bm.Camera(position=(G/2, G/8, 0.8*G), horizontal_angle=0.8, vertical_angle=-0.5).update()
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device='cuda')
    n = 30 if name == "config2" else 4
    for i in range(n + 3):
        scene.render(cam, bm.FrameParams(W, H, spp=spp, sample_base=i * spp, max_bounces=mb), acc)
    torch.cuda.synchronize()
    ms = scene.render_times(n)
    o = torch.zeros_like(acc)
    scene.render(cam, bm.FrameParams(W, H, spp=1, max_bounces=mb, flags=bm.BM_FLAG_ORDERED), o)
    torch.cuda.synchronize()
    print(f"{name}: median {np.median(ms):.4f} ms  min {ms.min():.4f}  ordered-frame sha {hashlib.sha256(o.cpu().numpy().tobytes()).hexdigest()[:16]}")
    scene.close()
