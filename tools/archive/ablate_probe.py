"""Ablation timings on the GPU box: resident scene vs nothing resident (every non-empty brick = solid hit, no 8^3 walk)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, brickmap_amd as bm
G, W, H = 1024, 1920, 1080
scene = bm.Scene(G, G, device=0).generate()
cam = bm.Camera(position=(G/2, G/8, 0.8*G), horizontal_angle=0.8, vertical_angle=-0.5).update()
acc = torch.zeros((H, W, 4), dtype=torch.float32, device='cuda')
def run(label, flags=0, **kw):
    for i in range(8):
        scene.render(cam, bm.FrameParams(W, H, spp=1, sample_base=i, flags=flags, **kw), acc)
    torch.cuda.synchronize()
    ms = scene.render_times(5)
    scene.counters_reset()
    scene.render(cam, bm.FrameParams(W, H, spp=1, flags=flags | bm.BM_FLAG_COUNTERS, **kw), acc)
    c = scene.counters()
    print("%-28s %.3f ms   cells %6.1fM  brick_tests %5.2fM  voxel_steps %5.1fM  rays %.2fM" % (label, ms.mean(), c["index_loads"]/1e6, c["brick_tests"]/1e6, c["voxel_steps"]/1e6, (c["extend_rays"]+c["shadow_rays"])/1e6))
run("unloaded 4seg", max_bounces=3)
run("unloaded primary-only", max_bounces=0, flags=bm.BM_FLAG_PRIMARY_ONLY)
scene.preload_all()
run("resident 4seg", max_bounces=3)
run("resident primary-only", max_bounces=0, flags=bm.BM_FLAG_PRIMARY_ONLY)
run("resident 1seg (shadow)", max_bounces=0)
run("resident 2seg", max_bounces=1)
