import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, brickmap_amd as bm
G, W, H = 1024, 1920, 1080
scene = bm.Scene(G, G, device=0).generate().preload_all()
cam = bm.Camera(position=(G/2, G/8, 0.8*G), horizontal_angle=0.8, vertical_angle=-0.5).update()
acc = torch.zeros((H, W, 4), dtype=torch.float32, device='cuda')
for K in (1, 20):
    scene.counters_reset()
    ps = [bm.FrameParams(W, H, spp=1, sample_base=100 + i, max_bounces=3, flags=bm.BM_FLAG_COUNTERS) for i in range(K)]
    if K == 1: scene.render(cam, ps[0], acc)
    else: scene.render_frames(cam, ps, acc)
    torch.cuda.synchronize()
    s = scene.sched_stats(); c = scene.counters()
    print(f"K = {K}: per frame:")
    for k, name in (("jump", "J"), ("step", "S"), ("candidate", "B"), ("shade", "C")):
        r, l = s[k + "_runs"] / K, s[k + "_lanes"] / K
        print(f"   {name}: {r/1e6:.3f} M passes at {l/max(r,1):.1f} lanes")
    print(f"   connect lanes {s['connect_lanes']/K/1e6:.3f} M in {s['connect_runs']/K/1e6:.3f} M passes; rays {(c['extend_rays']+c['shadow_rays'])/K/1e6:.3f} M; drain share {100*s['drain_cycles']/max(s['total_cycles'],1):.1f} %")
