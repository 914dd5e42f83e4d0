"""Prints bench-style timing plus the wave-scheduler statistics of the instrumented kernel (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, brickmap_amd as bm
G, W, H = 1024, 1920, 1080
scene = bm.Scene(G, G, device=0).generate().preload_all()
cam = bm.Camera(position=(G/2, G/8, 0.8*G), horizontal_angle=0.8, vertical_angle=-0.5).update()
acc = torch.zeros((H, W, 4), dtype=torch.float32, device='cuda')
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for i in range(13):
    scene.render(cam, bm.FrameParams(W, H, spp=spp, sample_base=i*spp, max_bounces=3), acc)
torch.cuda.synchronize()
ms = scene.render_times(10)
print("kernel ms avg %.4f  min %.4f  (spp=%d)  -> %.1f Mrays/s nominal" % (ms.mean(), ms.min(), spp, W*H*spp*4/ms.mean()/1e3))
scene.counters_reset()
scene.render(cam, bm.FrameParams(W, H, spp=spp, max_bounces=3, flags=bm.BM_FLAG_COUNTERS), acc)
c = scene.counters(); s = scene.sched_stats()
print(c)
for k in ("step", "candidate", "shade"):
    r, l = s[k+"_runs"], s[k+"_lanes"]
    cy = s[k+"_cycles"]
    print("%-10s runs %10d  avg active lanes %5.1f   cycles/run %8.1f   share of wave time %5.1f%%" % (k, r, l/max(r,1), cy/max(r,1), 100.0*cy/s["total_cycles"]))
print("connect    in %d shade passes, %5.1f lanes each" % (s["connect_runs"], s["connect_lanes"]/max(s["connect_runs"],1)))
print("jump       runs %10d  avg active lanes %5.1f" % (s["jump_runs"], s["jump_lanes"]/max(s["jump_runs"],1)))
print("drain      %5.1f%% of a wave's lifetime lies after its last refill attempt" % (100.0*s["drain_cycles"]/s["total_cycles"]))
print("waves", s["waves"], "avg wave cycles", s["total_cycles"]/s["waves"])
