"""The north-star's LDS sentence, measured: "the 64-byte 8^3 bitmask of the hit brick staged into LDS and broadcast across the 64-lane
wavefront".  With a -DBM_PHASE_TIMING -DBM_SHARE_PROBE build (tools/build_variants.sh share "-DBM_PHASE_TIMING -DBM_SHARE_PROBE"),
for the bench frame (config 2) and for its primary rays alone: in how many candidate passes do two or more lanes test the same brick
cell, and how many of the lanes of a candidate pass test a cell that a lower lane of the same pass tests as well (the brick fetches a
broadcast would save)?   usage: python tools/share_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, brickmap_amd as bm
G, W, H = 1024, 1920, 1080
scene = bm.Scene(G, G, device=0).generate().preload_all()
cam = bm.Camera(position=(G/2, G/8, 0.8*G), horizontal_angle=0.8, vertical_angle=-0.5).update()
acc = torch.zeros((H, W, 4), dtype=torch.float32, device='cuda')
for name, kw in (("full frame (1 spp, 4 segments, helper lanes)", dict(max_bounces=3)),
                 ("full frame, ordered (no helper lanes)", dict(max_bounces=3, flags=bm.BM_FLAG_ORDERED)),
                 ("primary rays only", dict(max_bounces=0, flags=bm.BM_FLAG_PRIMARY_ONLY)),
                 ("4 spp, pixel items", dict(max_bounces=3, spp=4))):
    spp = kw.pop("spp", 1)
    for i in range(2):
        scene.render(cam, bm.FrameParams(W, H, spp=spp, sample_base=i, **kw), acc)
    scene.counters_reset()
    scene.render(cam, bm.FrameParams(W, H, spp=spp, sample_base=9, **kw), acc)
    torch.cuda.synchronize()
    s, d = scene.sched_stats(), scene.sched_detail()
    passes, shared, dup = d["brick_passes"], d["brick_loop_trips"], d["brick_lane_steps"]
    lanes = s["candidate_lanes"]
    print(f"{name}: {passes} candidate passes at {lanes / max(passes, 1):.1f} lanes; in {100.0 * shared / max(passes, 1):.1f} % of them two or more lanes "
          f"test the same cell; {dup} of {lanes} candidate lanes ({100.0 * dup / max(lanes, 1):.1f} %) test a cell a lower lane tests too")
