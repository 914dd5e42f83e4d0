import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, brickmap_amd as bm
G, W, H = 1024, 1920, 1080
scene = bm.Scene(G, G, device=0).generate().preload_all()
acc = torch.zeros((H, W, 4), dtype=torch.float32, device='cuda')
def t(cam, **kw):
    for i in range(13):
        scene.render(cam, bm.FrameParams(W, H, spp=1, sample_base=i, **kw), acc)
    torch.cuda.synchronize()
    return scene.render_times(10).mean()
away = bm.Camera(position=(G/2, -3*G, 0.8*G), horizontal_angle=3.14159, vertical_angle=0.3).update()  # looks away from the world: every ray misses the box
cam = bm.Camera(position=(G/2, G/8, 0.8*G), horizontal_angle=0.8, vertical_angle=-0.5).update()
print("all-miss floor %.3f ms | primary-only %.3f ms | 4 segments %.3f ms" % (t(away, max_bounces=3), t(cam, max_bounces=0, flags=bm.BM_FLAG_PRIMARY_ONLY), t(cam, max_bounces=3)))
