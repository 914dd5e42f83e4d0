"""Streaming stress (GPU box): the 2048^3 world streamed from nothing with frames round-robin on three streams, a small ring (many
batches, pool moves, arena growths), both servicing modes; the steady-state frame must equal the resident scene's bit for bit.
usage: python tools/stress_stream.py [ring=4096]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, brickmap_amd as bm
ring = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
G, W, H = 2048, 800, 450
cam = bm.Camera(position=(G / 2, G / 8, 0.8 * G), horizontal_angle=0.8, vertical_angle=-0.5).update()
ref = bm.Scene(G, G, device=0).generate().preload_all()
want = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
wdbg = torch.zeros((H, W, 8), dtype=torch.int32, device="cuda:0")
p0 = bm.FrameParams(W, H, spp=1, max_bounces=7)
ref.render(cam, p0, want, debug=wdbg); torch.cuda.synchronize(); ref.close()
ok = True
for overlapped in (False, True):
    s = bm.Scene(G, G, device=0); s.set_queue_capacity(ring); s.generate(); s.set_streaming_mode(overlapped)
    streams = [torch.cuda.Stream() for _ in range(3)]
    bufs = [torch.zeros_like(want) for _ in range(3)]
    torch.cuda.synchronize()
    total = idle = 0
    for k in range(20000):
        j = k % 3
        s.render(cam, p0, bufs[j], stream=streams[j].cuda_stream)
        n = s.process_load_queue(); total += n
        idle = idle + 1 if n == 0 else 0
        if idle >= 6: break
    torch.cuda.synchronize()
    i = s.info()
    got = torch.zeros_like(want); gdbg = torch.zeros_like(wdbg)
    s.render(cam, p0, got, debug=gdbg); torch.cuda.synchronize()
    same = torch.equal(got, want) and torch.equal(gdbg, wdbg)
    print(f"overlapped={overlapped}: {k + 1} frames, {total} bricks in {i['stream_batches']} batches, resident {i['resident_bricks']}, arena growths {i['arena_growths']} (copying {i['arena_copy_growths']}), "
          f"failed {i['failed']}, steady-state frame == resident frame: {same}")
    ok = ok and same and total == i["resident_bricks"] and i["failed"] == 0
    s.close()
print("STRESS", "OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
