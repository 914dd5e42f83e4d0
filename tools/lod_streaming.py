"""README.md:9 of the reference: "[with LoD] the number of chunks requested to be streamed did decrease significantly".
Measured here: bricks streamed in until steady state, LoD on (reference thresholds) vs off, on the BASELINE config-5
world (4096^3 voxels) and on the reference's native world, 1920x1080, 1 spp, 4 segments, overlapped servicing."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, brickmap_amd as bm
W, H = 1920, 1080
def run(label, gs, gh, cam, lod_on):
    scene = bm.Scene(gs, gh, device=0)
    scene.set_queue_capacity(1 << 20)
    if not lod_on:
        scene.set_lod(2**31 - 1, 2**31 - 1)
    scene.generate()
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    frames = 0
    for f in range(200):
        scene.render(cam, bm.FrameParams(W, H, spp=1, sample_base=f, max_bounces=3), acc)
        frames += 1
        if scene.process_load_queue() == 0 and f > 2:
            break
    torch.cuda.synchronize()
    info = scene.info()
    ms = scene.render_times(3).mean()
    print("%-34s LoD %-3s: %8d of %9d bricks resident after %3d frames (%.1f MiB), %.2f ms/frame at steady state" %
          (label, "on" if lod_on else "off", info["resident_bricks"], info["total_bricks"], frames, info["resident_bricks"] * 64 / 2**20, ms))
    scene.close()
cam5 = bm.Camera(position=(2048.0, 512.0, 3276.8), horizontal_angle=0.8, vertical_angle=-0.5).update()
for lod in (True, False):
    run("config-5 world 4096^3", 4096, 4096, cam5, lod)
for lod in (True, False):
    run("native world, fly-through view 0", 4096, 512, bm.flythrough_camera(0), lod)
for lod in (True, False):
    run("native world, fly-through view 4", 4096, 512, bm.flythrough_camera(4), lod)
