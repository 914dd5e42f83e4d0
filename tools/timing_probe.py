"""With a -DBM_PHASE_TIMING build: phase time shares of the PLAIN kernel (no traversal counters, no hit records)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, brickmap_amd as bm
G, W, H = 1024, 1920, 1080
scene = bm.Scene(G, G, device=0).generate().preload_all()
cam = bm.Camera(position=(G/2, G/8, 0.8*G), horizontal_angle=0.8, vertical_angle=-0.5).update()
acc = torch.zeros((H, W, 4), dtype=torch.float32, device='cuda')
kw = dict(max_bounces=int(sys.argv[1]) if len(sys.argv) > 1 else 3, flags=int(sys.argv[2]) if len(sys.argv) > 2 else 0)
for i in range(5):
    scene.render(cam, bm.FrameParams(W, H, spp=1, sample_base=i, **kw), acc)
scene.counters_reset()
scene.render(cam, bm.FrameParams(W, H, spp=1, sample_base=9, **kw), acc)
torch.cuda.synchronize()
print("kernel ms", scene.last_render_ms())
s = scene.sched_stats()
for k in ("step", "candidate", "shade"):
    r, l, cy = s[k+"_runs"], s[k+"_lanes"], s[k+"_cycles"]
    print("%-10s runs %10d  avg lanes %5.1f  cycles/run %8.1f  share %5.1f%%" % (k, r, l/max(r,1), cy/max(r,1), 100.0*cy/s["total_cycles"]))
print("jump       runs %10d  avg lanes %5.1f   (time inside 'step')" % (s["jump_runs"], s["jump_lanes"]/max(s["jump_runs"],1)))
print("drain (from a wave's last refill attempt to its exit): %5.1f%% of its lifetime on average" % (100.0 * s["drain_cycles"] / s["total_cycles"]))
other = s["total_cycles"] - s["drain_cycles"] - sum(s[k+"_cycles"] for k in ("step", "candidate", "shade"))
print("scheduler + refill share %5.1f%%" % (100.0*other/s["total_cycles"]))
print("waves", s["waves"], "avg wave cycles", s["total_cycles"]/max(s["waves"],1))
