"""Cost of the pure brick-grid walk: rays that cross only EMPTY supercells (no mask reloads besides the per-supercell
record, no candidates) vs the terrain view.  Primary rays only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, brickmap_amd as bm
G, W, H = 1024, 1920, 1080
scene = bm.Scene(G, G, device=0).generate().preload_all()
acc = torch.zeros((H, W, 4), dtype=torch.float32, device='cuda')
def run(label, cam):
    kw = dict(max_bounces=0, flags=bm.BM_FLAG_PRIMARY_ONLY)
    for i in range(10):
        scene.render(cam, bm.FrameParams(W, H, spp=1, sample_base=i, **kw), acc)
    torch.cuda.synchronize()
    ms = scene.render_times(6).mean()
    scene.counters_reset()
    scene.render(cam, bm.FrameParams(W, H, spp=1, max_bounces=0, flags=bm.BM_FLAG_PRIMARY_ONLY | bm.BM_FLAG_COUNTERS), acc)
    c = scene.counters(); s = scene.sched_stats()
    print("%-22s %.3f ms  cells %6.1fM  tests %5.2fM  step runs %5.2fM (%.1f lanes)  => %.1f ps per cell, %.0f SIMD-cycles per step run" %
          (label, ms, c["index_loads"]/1e6, c["brick_tests"]/1e6, s["step_runs"]/1e6, s["step_lanes"]/max(s["step_runs"],1),
           (ms-0.105)*1e9/max(c["index_loads"],1), (ms-0.105)*1e-3*1024*2.4e9/max(s["step_runs"],1)))
run("terrain view", bm.Camera(position=(G/2, G/8, 0.8*G), horizontal_angle=0.8, vertical_angle=-0.5).update())
run("sky layer, level", bm.Camera(position=(G/2, 1.0, 990.0), horizontal_angle=0.0, vertical_angle=0.25).update())
run("sky layer, diagonal", bm.Camera(position=(1.0, 1.0, 960.0), horizontal_angle=0.785, vertical_angle=0.2).update())
