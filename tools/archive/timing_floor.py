import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, brickmap_amd as bm
G, W, H = 1024, 1920, 1080
scene = bm.Scene(G, G, device=0).generate().preload_all()
cam = bm.Camera(position=(G/2, -3*G, 0.8*G), horizontal_angle=3.14159, vertical_angle=0.3).update()
acc = torch.zeros((H, W, 4), dtype=torch.float32, device='cuda')
for i in range(5):
    scene.render(cam, bm.FrameParams(W, H, spp=1, sample_base=i), acc)
scene.counters_reset()
scene.render(cam, bm.FrameParams(W, H, spp=1, sample_base=9), acc)
torch.cuda.synchronize()
print("kernel ms", scene.last_render_ms())
s = scene.sched_stats()
for k in ("step", "candidate", "shade", "connect"):
    r, l, cy = s[k+"_runs"], s[k+"_lanes"], s[k+"_cycles"]
    print("%-10s runs %10d  avg lanes %5.1f  cycles/run %8.1f  share %5.1f%%" % (k, r, l/max(r,1), cy/max(r,1), 100.0*cy/max(s["total_cycles"],1)))
print("waves", s["waves"], "avg wave cycles", s["total_cycles"]/max(s["waves"],1))
