import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, brickmap_amd as bm
G = 256
scene = bm.Scene(G, G, device=0).generate().preload_all()
cam = bm.Camera(position=(G/2, -3*G, 0.8*G), horizontal_angle=3.14159, vertical_angle=0.3).update()
for (W, H, spp) in [(64, 64, 1), (256, 256, 1), (1024, 512, 1), (1920, 1080, 1), (3840, 2160, 1), (1920, 1080, 4), (1920, 1080, 16)]:
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device='cuda')
    for i in range(8):
        scene.render(cam, bm.FrameParams(W, H, spp=spp, sample_base=i*spp), acc)
    torch.cuda.synchronize()
    ms = scene.render_times(5).mean()
    print("%5dx%-5d spp %2d : %.4f ms  -> %.3f ns per path" % (W, H, spp, ms, ms * 1e6 / (W * H * spp)))
