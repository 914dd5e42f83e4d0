"""The reference's fly-through (performance_measure.h:4-25) on its native world (4096 x 4096 x 512 voxels, all bricks
resident): ms per 1920x1080 frame (1 spp, 4 segments) for each of the 9 viewpoints -- the regression scene of
SURVEY.md section 8(f) item 3.  Usage (GPU box): python tools/flythrough.py [frames_per_view]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, brickmap_amd as bm

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 10
W, H = 1920, 1080
scene = bm.Scene(4096, 512, device=0).generate().preload_all()
info = scene.info()
print("world 4096x4096x512: %d bricks (%.0f MiB), index grid %.0f MiB" % (info["total_bricks"], info["brick_bytes"] / 2**20, info["index_bytes"] / 2**20))
acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
for i in range(len(bm.FLYTHROUGH_VIEWS)):
    cam = bm.flythrough_camera(i)
    acc.zero_()
    for f in range(frames + 2):
        scene.render(cam, bm.FrameParams(W, H, spp=1, sample_base=f, max_bounces=3), acc)
    torch.cuda.synchronize()
    ms = scene.render_times(frames)
    scene.counters_reset()
    scene.render(cam, bm.FrameParams(W, H, spp=1, max_bounces=3, flags=bm.BM_FLAG_COUNTERS), acc)
    c = scene.counters()
    rays = c["extend_rays"] + c["shadow_rays"]
    print("view %d  pos %-34s  %.3f ms/frame  %.2f M rays  %.0f M cells  -> %.0f Mrays/s nominal" %
          (i, str(tuple(round(v, 1) for v in cam.position)), ms.mean(), rays / 1e6, c["index_loads"] / 1e6, W * H * 4 / ms.mean() / 1e3))
