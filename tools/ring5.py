import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, brickmap_amd as bm
W, H, spp, mb, ns = 7680, 4320, 4, 7, 32
G = 128 * ns
scene = bm.Scene(G, G, device=0).generate().preload_all()
cam = bm.Camera(position=(G/2, G/8, 0.8*G), horizontal_angle=0.8, vertical_angle=-0.5).update()
acc = torch.zeros((H, W, 4), dtype=torch.float32, device='cuda')
P = lambda i: bm.FrameParams(W, H, spp=spp, sample_base=i * spp, max_bounces=mb)
for rep in range(2):
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(5): scene.render(cam, P(i), acc)
    torch.cuda.synchronize(); t1 = (time.perf_counter() - t) / 5 * 1e3
    torch.cuda.synchronize(); t = time.perf_counter()
    scene.render_frames(cam, [P(i) for i in range(5)], acc)
    torch.cuda.synchronize(); t2 = (time.perf_counter() - t) / 5 * 1e3
    print(f"config5: single launches {t1:.3f} ms per frame, uniform ring of 5 {t2:.3f}")
