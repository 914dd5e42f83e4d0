"""Kernel time against the refill threshold (FrameConstants::refill_min, overridden per process with BM_REFILL_MIN) for work items of
different lengths: 1080p on the 1024^3 world at 1 / 2 / 3 / 4 / 8 samples per pixel item and 4 or 8 segments per path, the 1/8 shard
with (chunk, sample) items, and config 3.  usage: python tools/refill_sweep.py [values...]   (one child process per value; 0 = the rule)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, brickmap_amd as bm
def t(G, W, H, spp, mb, n, **kw):
    scene = t.scenes.setdefault(G, bm.Scene(G, G, device=0).generate().preload_all())
    cam = bm.Camera(position=(G/2, G/8, 0.8*G), horizontal_angle=0.8, vertical_angle=-0.5).update()
    p0 = bm.FrameParams(W, H, spp=spp, max_bounces=mb, **kw)
    acc = torch.zeros((bm.local_rows(p0), W, 4), dtype=torch.float32, device="cuda")
    for i in range(n + 3):
        scene.render(cam, bm.FrameParams(W, H, spp=spp, sample_base=i * spp, max_bounces=mb, **kw), acc)
    torch.cuda.synchronize()
    return float(np.median(scene.render_times(n)))
t.scenes = {}
out = []
for spp in (1, 2, 3, 4, 8):
    out.append(("1080p %%d spp x 4 seg" %% spp, t(1024, 1920, 1080, spp, 3, 12)))
out.append(("1080p 1 spp x 8 seg", t(1024, 1920, 1080, 1, 7, 12)))
out.append(("1080p 2 spp x 8 seg", t(1024, 1920, 1080, 2, 7, 12)))
out.append(("1/8 shard 8 spp sample items", t(1024, 1920, 1080, 8, 3, 12, band_rows=16, shard_rank=0, shard_count=8, flags=bm.BM_FLAG_SAMPLE_ITEMS)))
out.append(("1080p 4 spp sample items", t(1024, 1920, 1080, 4, 3, 12, flags=bm.BM_FLAG_SAMPLE_ITEMS)))
t.scenes.clear()
out.append(("config3 (4K 4 spp x 8 seg)", t(2048, 3840, 2160, 4, 7, 3)))
print(" | ".join("%%s %%.3f" %% o for o in out))
''' % ROOT
vals = [int(v) for v in sys.argv[1:]] or [0, 4, 8, 12, 16, 24]
for v in vals:
    env = dict(os.environ)
    env.pop("BM_REFILL_MIN", None)
    if v:
        env["BM_REFILL_MIN"] = str(v)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print(f"refill_min {'rule' if not v else v:>4}: {r.stdout.strip() or r.stderr[-400:]}", flush=True)
