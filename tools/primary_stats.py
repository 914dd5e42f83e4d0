import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch, brickmap_amd as bm
G, W, H = 1024, 1920, 1080
scene = bm.Scene(G, G, device=0).generate().preload_all()
cam = bm.Camera(position=(G/2, G/8, 0.8*G), horizontal_angle=0.8, vertical_angle=-0.5).update()
acc = torch.zeros((H, W, 4), dtype=torch.float32, device='cuda')
for label, kw in (("primary-only", dict(max_bounces=0, flags=bm.BM_FLAG_PRIMARY_ONLY)), ("full", dict(max_bounces=3, flags=0))):
    for i in range(8):
        scene.render(cam, bm.FrameParams(W, H, spp=1, sample_base=i, **kw), acc)
    torch.cuda.synchronize()
    ms = scene.render_times(6)
    scene.counters_reset()
    kw2 = dict(kw); kw2["flags"] = kw["flags"] | bm.BM_FLAG_COUNTERS
    scene.render(cam, bm.FrameParams(W, H, spp=1, sample_base=9, **kw2), acc)
    torch.cuda.synchronize()
    s, c = scene.sched_stats(), scene.counters()
    tot = max(1, s["total_cycles"])
    print(label, "%.3f ms" % np.median(ms), " ".join(f"{k} {s[k+'_runs']/1e6:.3f}M at {s[k+'_lanes']/max(1,s[k+'_runs']):.1f}" for k in ("jump", "step", "candidate", "shade")),
          f"drain {100.0*s['drain_cycles']/tot:.1f}% rays {(c['extend_rays']+c['shadow_rays'])/1e6:.2f}M cells {c['index_loads']/1e6:.1f}M tests {c['brick_tests']/1e6:.2f}M")
