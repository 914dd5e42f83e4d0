import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, brickmap_amd as bm
G, W, H = 1024, 1920, 1080
scene = bm.Scene(G, G, device=0).generate().preload_all()
cam = bm.Camera(position=(G/2, G/8, 0.8*G), horizontal_angle=0.8, vertical_angle=-0.5).update()
acc = torch.zeros((H, W, 4), dtype=torch.float32, device='cuda')
for i in range(13):
    scene.render(cam, bm.FrameParams(W, H, spp=1, sample_base=i, max_bounces=0, flags=bm.BM_FLAG_PRIMARY_ONLY), acc)
torch.cuda.synchronize()
print("primary-only %.3f ms" % scene.render_times(10).mean())
