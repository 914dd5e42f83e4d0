"""Steady-state timing of the wavefront mode against the fused per-pixel kernel on the bench workload (config 2).
usage: python tools/wavefront_time.py [frames]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import brickmap_amd as bm

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 24
G, W, H, Q = 1024, 1920, 1080, 2 * 1048576
scene = bm.Scene(G, G, device=0).generate()
scene.preload_all()
cam = bm.Camera(position=(G / 2, G / 8, 0.8 * G), horizontal_angle=0.8, vertical_angle=-0.5).update()
acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")

# per-pixel reference point: actual rays per frame and time
p = bm.FrameParams(W, H, spp=1, max_bounces=3, flags=bm.BM_FLAG_COUNTERS)
scene.counters_reset()
scene.render(cam, p, acc); torch.cuda.synchronize()
c = scene.counters()
rays_pp = c["extend_rays"] + c["shadow_rays"]
p = bm.FrameParams(W, H, spp=1, max_bounces=3)
for _ in range(5): scene.render(cam, p, acc)
torch.cuda.synchronize()
ts = []
for _ in range(10):
    scene.render(cam, p, acc); ts.append(scene.last_render_ms())
ms_pp = float(np.median(ts))
print(f"per-pixel : {ms_pp:.3f} ms/frame, {rays_pp/1e6:.2f} M rays -> {rays_pp/ms_pp/1e3:.0f} Mrays/s actual")

wf = bm.Wavefront(scene, Q)
acc.zero_()
pw = bm.FrameParams(W, H, max_bounces=3)
rows = []
for f in range(frames):
    wf.frame(cam, pw, acc)
    t = wf.times(); st = wf.stats()
    rows.append((t, st))
for f, (t, st) in enumerate(rows):
    if f < 6 or f >= frames - 3:
        rays = Q + st["shadow"]
        print(f"frame {f+1:2d}: total {t['total']:.3f} primary {t['primary']:.3f} extend {t['extend']:.3f} shade {t['shade']:.3f} connect {t['connect']:.3f} ms"
              f" | survivors {st['survivors']} shadow {st['shadow']} -> {rays/t['total']/1e3:.0f} Mrays/s")
tail = rows[8:]
tot = sum(t["total"] for t, _ in tail); rays = sum(Q + st["shadow"] for _, st in tail)
print(f"steady state: {tot/len(tail):.3f} ms/frame, {rays/len(tail)/1e6:.2f} M rays/frame -> {rays/tot/1e3:.0f} Mrays/s actual (per-pixel {rays_pp/ms_pp/1e3:.0f})")

# scheduler statistics of the instrumented kernels over 4 steady-state frames
wf.counters_reset()
pc = bm.FrameParams(W, H, max_bounces=3, flags=bm.BM_FLAG_COUNTERS)
for _ in range(4):
    wf.frame(cam, pc, acc)
for which in ("extend", "connect"):
    s, c = wf.sched_stats(which), wf.counters(which)
    rays = c["extend_rays"] + c["shadow_rays"]
    print(f"{which}: {rays/4/1e6:.2f} M rays/frame, cells/ray {c['index_loads']/rays:.1f}, candidates/ray {s['candidate_lanes']/rays:.2f}, "
          f"A lanes/run {s['step_lanes']/max(s['step_runs'],1):.1f}, B lanes/run {s['candidate_lanes']/max(s['candidate_runs'],1):.1f}, "
          f"refill rays/run {s['refill_rays']/max(s['refills'],1):.1f}, A runs/ray {s['step_runs']/rays*64:.1f}x64, B runs/ray {s['candidate_runs']/rays*64:.2f}x64")
