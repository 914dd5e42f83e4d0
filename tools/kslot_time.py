"""Default schedule (trace.hip) against the K-slot schedule (trace_k.hip, BM_FLAG_KSLOT) on a bench workload's frame:
kernel ms (median of the timed launches, one stream), frames bit-identical or not, and the instrumented kernels' pass statistics.
usage: python tools/kslot_time.py [config2|config3|config5] [launches] [spp]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, brickmap_amd as bm
name = sys.argv[1] if len(sys.argv) > 1 else "config2"
W, H, spp, mb, ns = {"config2": (1920, 1080, 1, 3, 8), "config3": (3840, 2160, 4, 7, 16), "config5": (7680, 4320, 4, 7, 32)}[name]
n = int(sys.argv[2]) if len(sys.argv) > 2 else (20 if name == "config2" else 4)
if len(sys.argv) > 3: spp = int(sys.argv[3])
G = 128 * ns
scene = bm.Scene(G, G, device=0).generate().preload_all()
cam = bm.Camera(position=(G/2, G/8, 0.8*G), horizontal_angle=0.8, vertical_angle=-0.5).update()
frames = {}
for label, flag in (("default", 0), ("kslot", bm.BM_FLAG_KSLOT)):
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device='cuda')
    for i in range(n + 3):
        scene.render(cam, bm.FrameParams(W, H, spp=spp, sample_base=i * spp, max_bounces=mb, flags=flag), acc)
    torch.cuda.synchronize()
    ms = scene.render_times(n)
    frames[label] = acc.cpu().numpy()
    print(f"{name} spp {spp} {label:8s}: median {np.median(ms):.4f} ms, min {ms.min():.4f}  ({W*H*spp*(mb+1)/np.median(ms)/1e3:.0f} nominal Mrays/s)")
    scene.counters_reset()
    acc2 = torch.zeros((H, W, 4), dtype=torch.float32, device='cuda')
    scene.render(cam, bm.FrameParams(W, H, spp=spp, max_bounces=mb, flags=flag | bm.BM_FLAG_COUNTERS), acc2)
    torch.cuda.synchronize()
    s, c = scene.sched_stats(), scene.counters()
    tot = max(1, s["total_cycles"])
    print("   instrumented: " + "  ".join(f"{k} {s[k+'_runs']/1e6:.3f}M at {s[k+'_lanes']/max(1,s[k+'_runs']):.1f}" for k in ("jump", "step", "candidate", "shade")) +
          "  | cycles/run " + " ".join(f"{k[:4]} {s[k+'_cycles']/max(1,s[r+'_runs']):.0f}" for k, r in (("candidate", "candidate"), ("shade", "shade"))) +
          f" walk/(jump+4step) {s['step_cycles']/max(1,s['jump_runs']+s['step_runs']):.0f} | shares A {100.0*s['step_cycles']/tot:.0f}% B {100.0*s['candidate_cycles']/tot:.0f}% C {100.0*s['shade_cycles']/tot:.0f}%" +
          f"  drain {100.0*s['drain_cycles']/tot:.1f}%  rays {(c['extend_rays']+c['shadow_rays'])/1e6:.2f}M cells {c['index_loads']/1e6:.1f}M")
same = np.array_equal(frames["default"].view(np.uint32), frames["kslot"].view(np.uint32))
print("frames bit-identical:", same)
