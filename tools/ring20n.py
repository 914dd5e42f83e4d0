# non-uniform ring of 20 frames (two alternating buffers: the buffers do not step by a constant), kernel ms per frame
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, brickmap_amd as bm
G, W, H = 1024, 1920, 1080
scene = bm.Scene(G, G, device=0).generate().preload_all()
cam = bm.Camera(position=(G/2, G/8, 0.8*G), horizontal_angle=0.8, vertical_angle=-0.5).update()
a = torch.zeros((H, W, 4), dtype=torch.float32, device='cuda'); pad = torch.zeros(12345, device='cuda'); b = torch.zeros_like(a)
K = 20; best = 1e9
for rep in range(8):
    ps = [bm.FrameParams(W, H, spp=1, sample_base=rep * K + i, max_bounces=3) for i in range(K)]
    scene.render_frames(cam, ps, [(a, b, b, a)[i % 4] for i in range(K)])
    torch.cuda.synchronize()
    if rep >= 3: best = min(best, scene.last_render_ms() / K)
print(f"wave-level ring of 20: {best:.4f} ms per frame (kernel)")
