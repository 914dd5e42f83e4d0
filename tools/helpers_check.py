"""Helper lanes (trace.hip HELP, the default) against ordered accumulation (BM_FLAG_ORDERED): the same frame rendered both ways --
radiance equal up to summation order, terminated-path counts (alpha) identical, traversal counters identical -- and the kernel time
of both.  usage: python tools/helpers_check.py [config2|config3|config5] [launches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, brickmap_amd as bm
name = sys.argv[1] if len(sys.argv) > 1 else "config2"
W, H, spp, mb, ns = {"config2": (1920, 1080, 1, 3, 8), "config2x4": (1920, 1080, 4, 3, 8), "config3": (3840, 2160, 4, 7, 16), "config5": (7680, 4320, 4, 7, 32)}[name]
n = int(sys.argv[2]) if len(sys.argv) > 2 else (30 if name.startswith("config2") else 4)
G = 128 * ns
scene = bm.Scene(G, G, device=0).generate().preload_all()
cam = bm.Camera(position=(G/2, G/8, 0.8*G), horizontal_angle=0.8, vertical_angle=-0.5).update()
out = {}
for label, flags in (("ordered", bm.BM_FLAG_ORDERED), ("helpers", 0)):
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device='cuda')
    scene.counters_reset()
    scene.render(cam, bm.FrameParams(W, H, spp=spp, sample_base=0, max_bounces=mb, flags=flags | bm.BM_FLAG_COUNTERS), acc)
    torch.cuda.synchronize()
    cnt = scene.counters()
    acc2 = torch.zeros_like(acc)
    scene.render(cam, bm.FrameParams(W, H, spp=spp, sample_base=0, max_bounces=mb, flags=flags), acc2)
    torch.cuda.synchronize()
    warm = 6 if name.startswith("config2") else 1
    scratch = torch.zeros_like(acc)
    for i in range(n + warm):
        scene.render(cam, bm.FrameParams(W, H, spp=spp, sample_base=i * spp, max_bounces=mb, flags=flags), scratch)
    torch.cuda.synchronize()
    ms = scene.render_times(n)
    out[label] = (acc.cpu().numpy(), acc2.cpu().numpy(), cnt, float(np.median(ms)), float(ms.min()))
    print(f"{name} {label}: median {np.median(ms):.4f} ms, min {ms.min():.4f}  ({W*H*spp*(mb+1)/np.median(ms)/1e3:.0f} nominal Mrays/s)", flush=True)
a_i, a_p, c0, _, _ = out["ordered"]
b_i, b_p, c1, _, _ = out["helpers"]
assert np.array_equal(a_i.view(np.uint32), a_p.view(np.uint32)), "ordered: instrumented and plain kernels differ"
for b in (b_i, b_p):
    assert np.array_equal(a_i[..., 3], b[..., 3]), "alpha (terminated paths) differs"
    err = np.abs(b[..., :3] - a_i[..., :3]) / np.maximum(np.abs(a_i[..., :3]), 1e-6)
    print(f"  helpers vs ordered: max rel radiance difference {err.max():.3e}")
    assert err.max() < 2e-5
assert c0 == c1, (c0, c1)
print("  counters identical:", c0)
print(f"  speed-up {out['ordered'][3] / out['helpers'][3]:.4f} (median), {out['ordered'][4] / out['helpers'][4]:.4f} (min)")
