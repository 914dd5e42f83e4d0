"""With a -DBM_PHASE_TIMING build: phase time shares of the PLAIN kernel (no traversal counters, no hit records)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, brickmap_amd as bm
G, W, H = 1024, 1920, 1080
scene = bm.Scene(G, G, device=0).generate().preload_all()
cam = bm.Camera(position=(G/2, G/8, 0.8*G), horizontal_angle=0.8, vertical_angle=-0.5).update()
acc = torch.zeros((H, W, 4), dtype=torch.float32, device='cuda')
kw = dict(max_bounces=int(sys.argv[1]) if len(sys.argv) > 1 else 3, flags=int(sys.argv[2]) if len(sys.argv) > 2 else 0)
for i in range(5):
    scene.render(cam, bm.FrameParams(W, H, spp=1, sample_base=i, **kw), acc)
scene.counters_reset()
scene.render(cam, bm.FrameParams(W, H, spp=1, sample_base=9, **kw), acc)
torch.cuda.synchronize()
print("kernel ms", scene.last_render_ms())
s = scene.sched_stats()
for k in ("step", "candidate", "shade"):
    r, l, cy = s[k+"_runs"], s[k+"_lanes"], s[k+"_cycles"]
    print("%-10s runs %10d  avg lanes %5.1f  cycles/run %8.1f  share %5.1f%%" % (k, r, l/max(r,1), cy/max(r,1), 100.0*cy/s["total_cycles"]))
print("jump       runs %10d  avg lanes %5.1f   (time inside 'step')" % (s["jump_runs"], s["jump_lanes"]/max(s["jump_runs"],1)))
print("drain (from a wave's last refill attempt to its exit): %5.1f%% of its lifetime on average" % (100.0 * s["drain_cycles"] / s["total_cycles"]))
other = s["total_cycles"] - s["drain_cycles"] - sum(s[k+"_cycles"] for k in ("step", "candidate", "shade"))
print("scheduler + refill share %5.1f%%" % (100.0*other/s["total_cycles"]))
print("waves", s["waves"], "avg wave cycles", s["total_cycles"]/max(s["waves"],1))
d = scene.sched_detail()
csum = sum(d[k] for k in ("connect_cycles", "shade_hit_cycles", "sky_cycles", "primary_cycles", "setup_cycles"))
if csum:
    print("shade pass split: " + "  ".join("%s %.1f%%" % (k[:-7], 100.0 * d[k] / csum) for k in ("connect_cycles", "shade_hit_cycles", "sky_cycles", "primary_cycles", "setup_cycles")))
if d["brick_passes"]:
    print("candidate passes that walked a brick: %d, loop length (longest walk of the pass) %.2f cells, lanes' own walks %.2f cells on %.1f lanes" %
          (d["brick_passes"], d["brick_loop_trips"] / d["brick_passes"], d["brick_lane_steps"] / max(1, s["candidate_lanes"]), s["candidate_lanes"] / max(1, s["candidate_runs"])))
# the one-path-per-lane bound (docs/HISTORY.md 5.4): with pass types s of cost c_s (issue time of one pass) that a ray needs o_s
# times, the time per ray on a 64-lane wave is sum(o_s c_s / n_s) with sum(n_s) <= 64 lanes to share: minimal for
# n_s ~ sqrt(o_s c_s), i.e. lane utilisation <= sum(w_s) / (sum(sqrt(w_s)))^2 with w_s = o_s c_s
import math
rays = None
try:
    scene.counters_reset(); scene.render(cam, bm.FrameParams(W, H, spp=1, sample_base=9, flags=bm.BM_FLAG_COUNTERS, **{k: v for k, v in kw.items() if k != "flags"}), acc); c = scene.counters()
    rays = c["extend_rays"] + c["shadow_rays"]
except Exception as e:
    print("counters:", e)
if rays:
    w = {}
    for k, runs, lanes, cyc in (("jump", s["jump_runs"], s["jump_lanes"], None), ("step", s["step_runs"], s["step_lanes"], None),
                                ("candidate", s["candidate_runs"], s["candidate_lanes"], s["candidate_cycles"]), ("shade", s["shade_runs"], s["shade_lanes"] + s["connect_lanes"], s["shade_cycles"])):
        w[k] = (runs, lanes)
    # pass costs in cycles: walk passes share step_cycles in proportion 150 : 35 (instructions per jump pass : single move)
    tot_w = s["jump_runs"] * 150.0 + s["step_runs"] * 35.0
    cost = {"jump": s["step_cycles"] * 150.0 / tot_w, "step": s["step_cycles"] * 35.0 / tot_w, "candidate": s["candidate_cycles"] / max(1, s["candidate_runs"]), "shade": s["shade_cycles"] / max(1, s["shade_runs"])}
    ws = {k: (w[k][1] / rays) * cost[k] for k in w}  # o_s * c_s per ray
    bound = sum(ws.values()) / sum(math.sqrt(v) for v in ws.values()) ** 2
    used = sum(w[k][1] * cost[k] for k in w) / (64.0 * sum(w[k][0] * cost[k] for k in w))
    print("per ray: " + "  ".join("%s %.2f visits x %.0f cycles" % (k, w[k][1] / rays, cost[k]) for k in w))
    print("lane utilisation (time-weighted) %.1f%%; bound for one path per lane with these costs %.1f%%" % (100 * used, 100 * bound))
