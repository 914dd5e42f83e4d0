"""Streaming as a RATE: the reference's fly-through (performance_measure.h:4-25, nine viewpoints) on its
native 4096 x 4096 x 512 world, starting from EMPTY residency; every view is held until nothing is requested any more.
Per servicing mode (reference order: the host waits for the frame, Scene.cpp:200-252; overlapped: two rings, the host never
waits) it reports frames, wall ms per frame while bricks stream in and at steady state, bricks uploaded and bricks/s, the
host time bm_scene_process_load_queue spends staging (us per 1000 bricks), arena growths (none of them copying).
usage (GPU box): python tools/stream_flythrough.py [out.json] [ring_capacity]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, brickmap_amd as bm

out_path = sys.argv[1] if len(sys.argv) > 1 else None
ring = int(sys.argv[2]) if len(sys.argv) > 2 else (1 << 16)
W, H = 1920, 1080
scene = bm.Scene(4096, 512, device=0)
scene.set_queue_capacity(ring)
t0 = time.time()
scene.generate()
build_s = time.time() - t0
acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
result = {"world": "4096 x 4096 x 512 voxels (reference native), %d bricks" % scene.info()["total_bricks"], "frame": f"{W}x{H}, 1 spp, 4 segments",
          "ring_capacity": ring, "world_build_s": round(build_s, 1), "modes": {}}
for mode in ("reference-order", "overlapped"):
    scene.set_streaming_mode(False)
    scene.reset_residency()
    scene.set_streaming_mode(mode == "overlapped")
    views = []
    total_bricks, total_frames, fill_s, fill_frames = 0, 0, 0.0, 0
    for v in range(len(bm.FLYTHROUGH_VIEWS)):
        cam = bm.flythrough_camera(v)
        p = bm.FrameParams(W, H, spp=1, max_bounces=3)
        n_frames, bricks, idle = 0, 0, 0
        torch.cuda.synchronize()
        t_view = time.perf_counter()
        t_steady = None
        while n_frames < 3000:
            scene.render(cam, p, acc)
            n = scene.process_load_queue()
            n_frames += 1
            bricks += n
            idle = idle + 1 if n == 0 else 0
            if idle >= 3:
                break
        torch.cuda.synchronize()
        fill = time.perf_counter() - t_view
        # steady state of this view: 20 more frames
        t1 = time.perf_counter()
        for _ in range(20):
            scene.render(cam, p, acc)
            scene.process_load_queue()
        torch.cuda.synchronize()
        steady_ms = (time.perf_counter() - t1) / 20 * 1e3
        views.append({"view": v, "frames_until_steady": n_frames, "bricks_uploaded": bricks, "fill_ms_per_frame": round(fill / n_frames * 1e3, 3),
                      "steady_ms_per_frame": round(steady_ms, 3), "fill_bricks_per_s": round(bricks / fill) if bricks else 0})
        total_bricks += bricks; total_frames += n_frames; fill_s += fill; fill_frames += n_frames
    info = scene.info()
    result["modes"][mode] = {
        "views": views, "bricks_uploaded": total_bricks, "resident_bricks": info["resident_bricks"],
        "frames_while_filling": fill_frames, "ms_per_frame_while_filling": round(fill_s / fill_frames * 1e3, 3),
        "bricks_per_s_while_filling": round(total_bricks / fill_s),
        "upload_batches": info["stream_batches"], "host_staging_us_per_1000_bricks": round(info["stream_host_ns"] / 1e3 / max(total_bricks, 1) * 1000, 1),
        "host_staging_ms_per_batch": round(info["stream_host_ns"] / 1e6 / max(info["stream_batches"], 1), 3),
        "arena": {"virtual": info["arena_virtual"], "growths": info["arena_growths"], "copy_growths": info["arena_copy_growths"],
                  "arena_MiB": round(info["brick_bytes"] / 2**20, 1), "pool_MiB": round(info["pool_bytes"] / 2**20, 1)},
    }
    print(mode, json.dumps({k: v for k, v in result["modes"][mode].items() if k != "views"}))
    for vw in views:
        print("   ", vw)
if out_path:
    json.dump(result, open(out_path, "w"), indent=1)
