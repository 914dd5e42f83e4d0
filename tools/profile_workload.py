#!/usr/bin/env python3
"""Profile evidence for one bench workload on the GPU box (run it through gpurun):

    python tools/profile_workload.py <workload> <round-tag> [bench args...]

writes, under gpurun_out/ (copy what is to be judged into profiles/):
  <tag>_bench_<workload>.json          the bench line of `python bench.py --workload <workload> --steps S --warmup W`
  <tag>_kernel_stats_<workload>.csv    rocprofv3 --kernel-trace --stats summary of the SAME command
  <tag>_pmc_summary_<workload>.json    per-launch averages of bm::trace_paths<false> from separate --pmc passes (kernel-trace
                                       only, one counter group per pass) + the derived figures DESIGN.md quotes
Counter units / corrections: FETCH_SIZE, WRITE_SIZE in KiB.  On gfx950 FETCH_SIZE = fabric read requests x 64 B; MI355X_MICROARCH.md
prescribes x2 for full-line (128-byte) coalesced streams and says other patterns must be calibrated -- done in
profiles/r03_fetch_calibration.txt (tools/ubench/fetch_calib.hip): every access of trace_paths (1-byte field lookups, 4-byte
index words, 64-byte bricks) is a single-SECTOR request, for which FETCH_SIZE is exact as reported: factor 1.0.  MALL hits
are included (the counters cannot separate them).  WRITE_SIZE as reported.
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
KERNEL = os.environ.get("BM_PROFILE_KERNEL", "trace_paths<false,")  # (every production instantiation: hand-out, helper lanes, frame ring)
PMC_FRAMES = 4  # the counter passes run `--steps 4 --warmup 4`: on resident workloads two launches of four frames each (the frame ring), same kernel
FETCH_FACTOR = 1.0  # profiles/r03_fetch_calibration.txt: single-sector requests are counted at their true 64 bytes
PMC_SETS = [
    "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY",
    "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR",
    "GRBM_GUI_ACTIVE TCC_HIT TCC_MISS TCC_REQ",
    "FETCH_SIZE",
    "WRITE_SIZE",
    "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ",
]


def main():
    workload, tag = sys.argv[1], sys.argv[2]
    extra = sys.argv[3:]
    os.makedirs(OUT, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    bench = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload] + extra
    steps = ["--steps", "5", "--warmup", "2"] if "--steps" not in extra else []
    # 1. a first bench line: the kernel duration the derived figures of the PMC summary are priced with
    def bench_line():
        r = subprocess.run(bench + steps, capture_output=True, text=True, cwd="/tmp", env=env)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(r.stdout[-2000:], r.stderr[-4000:])
            raise SystemExit("bench failed")
        return json.loads(line[-1])
    bench_json = bench_line()
    print("bench:", bench_json["value"], bench_json["unit"], bench_json["ms_per_step"], "ms/step, roofline frac", bench_json["roofline"]["frac"])
    # 2. kernel trace + stats of the same command (without the untimed extra measurements of the N = 1 line, so that the
    #    per-kernel averages are those of the warm-up + timed launches)
    d = f"/tmp/prof_{tag}_{workload}"
    shutil.rmtree(d, ignore_errors=True)
    subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "-o", "p", "--"] + bench + steps + ["--no-cpu-baseline", "--no-extras"],
                   capture_output=True, text=True, cwd="/tmp", env=env)
    stats = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    kernel_avg_ns = None
    # the TIMED launches alone, from the trace of the same run: the stats' average also covers the warm-up launches (on a streaming
    # workload those include the fill of the brick pools), the roofline prices the timed ones
    timed_ms = None
    n_timed = int(bench_json["roofline"].get("launches") or (extra + steps)[(extra + steps).index("--steps") + 1])  # timed LAUNCHES (a launch may hold several steps)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        rows = [r_ for r_ in csv.DictReader(open(f)) if KERNEL in r_.get("Kernel_Name", "")]
        rows.sort(key=lambda r_: int(r_["Start_Timestamp"]))
        last = rows[-n_timed:]
        if last:
            timed_ms = sum(int(r_["End_Timestamp"]) - int(r_["Start_Timestamp"]) for r_ in last) / len(last) / 1e6
            print(f"rocprofv3 kernel trace: the last {len(last)} launches (the timed ones) average {timed_ms:.4f} ms")
    if stats:
        shutil.copy(stats[0], os.path.join(OUT, f"{tag}_kernel_stats_{workload}.csv"))
        for row in csv.DictReader(open(stats[0])):
            if KERNEL in row.get("Name", ""):
                kernel_avg_ns = float(row["AverageNs"])
                print("rocprofv3 stats:", row["Name"][:60], "calls", row["Calls"], "avg ms", kernel_avg_ns / 1e6)
    # 3. PMC passes (short runs: 2 steps)
    tot, n = collections.defaultdict(float), collections.defaultdict(int)
    for i, group in enumerate(PMC_SETS):
        d = f"/tmp/pmc_{tag}_{workload}_{i}"
        shutil.rmtree(d, ignore_errors=True)
        subprocess.run(["rocprofv3", "--kernel-trace", "--pmc"] + group.split() + ["--output-format", "csv", "-d", d, "-o", "p", "--"] + bench +
                       ["--steps", str(PMC_FRAMES), "--warmup", str(PMC_FRAMES), "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, cwd="/tmp", env=env)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if KERNEL in row["Kernel_Name"]:
                    tot[row["Counter_Name"]] += float(row["Counter_Value"])
                    n[row["Counter_Name"]] += 1
    # frames per launch of those runs: what bench.py issues by default for this workload (resident: all four steps in one launch; streaming: one)
    pmc_frames = PMC_FRAMES if bench_json["config"].get("frames_per_launch", 1) > 1 else 1
    sys.path.insert(0, ROOT)
    import bench as bench_mod
    s = {"workload": workload, "frames_per_launch": pmc_frames, "collected_at_commit": bench_mod.git_head(),
         "per": f"launch of bm::{KERNEL} of {pmc_frames} frame(s) (average over the {max(n.values()) if n else 0} launches of a short bench run, warm-up and streaming fill included)"}
    for c in sorted(tot):
        s[c + ("_KiB" if c in ("FETCH_SIZE", "WRITE_SIZE") else "")] = tot[c] / n[c]
    if "FETCH_SIZE_KiB" in s and "WRITE_SIZE_KiB" in s:
        hbm = (FETCH_FACTOR * s["FETCH_SIZE_KiB"] + s["WRITE_SIZE_KiB"]) * 1024
        ms = bench_json["roofline"]["kernel_ms_per_step"] * pmc_frames  # duration of a launch of the counter passes' shape, at this run's rate
        s["derived"] = {
            "fabric_bytes_per_launch (1.0 x FETCH + WRITE; sector requests, MALL hits included)": hbm,
            "sector_requests_per_s_G": (s["TCC_MISS"] / (ms * 1e-3) / 1e9) if s.get("TCC_MISS") else None,
            "frac_of_measured_random_sector_ceiling (48 G requests/s)": (s["TCC_MISS"] / (ms * 1e-3) / 48e9) if s.get("TCC_MISS") else None,
            "kernel_ms_per_step (bench, HIP events)": bench_json["roofline"]["kernel_ms_per_step"],
            "kernel_ms_avg (bench, HIP events, per timed launch)": bench_json["roofline"]["kernel_ms_avg"],
            "kernel_ms_avg (rocprofv3 --stats)": kernel_avg_ns / 1e6 if kernel_avg_ns else None,
            "kernel_ms_avg (rocprofv3 kernel trace, the timed launches only)": timed_ms,
            "hbm_GBps": hbm / (ms * 1e-3) / 1e9,
            "frac_of_8TBps_by_counters": hbm / (ms * 1e-3) / 8e12,
            "frac_of_8TBps_algorithmic": bench_json["roofline"]["frac"],
            "algorithmic_bytes_per_launch": bench_json["roofline"]["algorithmic_bytes_per_launch"],
            "tcc_hit_rate": s["TCC_HIT"] / s["TCC_REQ"] if s.get("TCC_REQ") else None,
            "valu_lane_utilisation": s["SQ_THREAD_CYCLES_VALU"] / (64.0 * s["SQ_ACTIVE_INST_VALU"]) if s.get("SQ_ACTIVE_INST_VALU") else None,
        }
    json.dump(s, open(os.path.join(OUT, f"{tag}_pmc_summary_{workload}.json"), "w"), indent=1)
    print(json.dumps(s.get("derived", {}), indent=1))
    # 4. the bench line of step 1 (timed BEFORE the counter passes: rocprofv3's PMC collection leaves the GPU at its profiling clocks for a
    #    while, a line measured after it reads 15-20 % slow) with the counter-derived fields of its roofline taken from THIS summary --
    #    exactly what bench.py does with the committed file: the same function, pointed at the new file
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    shutil.copy(os.path.join(OUT, f"{tag}_pmc_summary_{workload}.json"), os.path.join(ROOT, "profiles", f"{tag}_pmc_summary_{workload}.json"))
    assert bench_mod.PROFILE_ROUNDS[0] == tag, f"bench.py PROFILE_ROUNDS must start with {tag}"
    rf = bench_json["roofline"]
    rf.update(bench_mod.counter_figures(workload, rf["kernel_ms_per_step"] * 1e-3, sum(rf["frames_per_launch"]) / len(rf["frames_per_launch"])))
    json.dump(bench_json, open(os.path.join(OUT, f"{tag}_bench_{workload}.json"), "w"), indent=1)
    print("bench line:", bench_json["value"], bench_json["unit"], bench_json["ms_per_step"], "ms/step, frac", rf["frac"], "frac_by_counters", rf.get("frac_by_counters"),
          "lanes", rf.get("valu_lanes"))
    if "derived" in s:
        want = int((FETCH_FACTOR * s["FETCH_SIZE_KiB"] + s["WRITE_SIZE_KiB"]) * 1024 / pmc_frames)
        assert rf["traffic_per_step"] == want and f"{tag}_pmc_summary_{workload}.json" in rf["traffic_source"], (rf["traffic"], want, rf["traffic_source"])

if __name__ == "__main__":
    main()
