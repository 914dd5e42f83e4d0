# wall ms per frame of a uniform ring of 20 frames (config 2), best of 5 launches; usage via tools/ab_run.py <lib> tools/ring20.py [spp]
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, brickmap_amd as bm
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 1
G, W, H = 1024, 1920, 1080
scene = bm.Scene(G, G, device=0).generate().preload_all()
cam = bm.Camera(position=(G/2, G/8, 0.8*G), horizontal_angle=0.8, vertical_angle=-0.5).update()
acc = torch.zeros((H, W, 4), dtype=torch.float32, device='cuda')
K = 20
best = 1e9
for rep in range(8):
    ps = [bm.FrameParams(W, H, spp=spp, sample_base=(rep * K + i) * spp, max_bounces=3) for i in range(K)]
    scene.render_frames(cam, ps, acc)
    torch.cuda.synchronize()
    if rep >= 3: best = min(best, scene.last_render_ms() / K)
print(f"ring of 20, {spp} spp: {best:.4f} ms per frame (kernel)")
