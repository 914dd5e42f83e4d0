#!/usr/bin/env python3
"""bench.py -- Mrays/s of the brickmap path-trace hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one launch of the path-trace kernel over the workload's frame: `spp` complete paths
(primary + up to max_bounces bounces, one shadow ray per hit) for every pixel of this rank's rows,
with the scene already resident in HBM.  N = 1 runs BASELINE.json configs[1]:
1920x1080, 1 spp, 4-bounce (MAX_BOUNCES = 3 -> 4 segments), 8x8x8 superchunks, all bricks resident.
N > 1 keeps the per-GPU work fixed (weak scaling) by sharding the SAMPLES: every rank traces the full frame with its
own `spp` sample indices of each pixel (N*spp samples per pixel and step in total), then the N float4 frames are
summed onto rank 0 over RCCL (brickmap_amd/dist.py FrameReducer); the reduction of frame i overlaps the tracing of
frame i+1 and every reduction, including the last one, completes inside the timed region.  (Sharding the pixels
instead -- row bands + gather, also in dist.py -- gives a bit-identical image but leaves each rank N*spp samples on
1/N of the rows, a launch shape that costs the persistent kernel 1.2x / 1.5x / 2.5x at N = 2 / 4 / 8.)

metric: Mrays/s = width * height * spp_total * segments / seconds  (nominal rays).
The line also carries `roofline` (algorithmic bytes of the kernel / its HIP-event duration against
the 8 TB/s HBM peak) and, on rank 0 at N = 1, `cpu_baseline` (the oracle's scalar C port of the same
path timed on the host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def workload(name):
    """BASELINE.json configs (world = 128*N voxels per side for N^3 superchunks)."""
    table = {
        # name: (width, height, spp, max_bounces, superchunks per side, streaming)
        "config1": (256, 256, 1, 0, 1, False),
        "config2": (1920, 1080, 1, 3, 8, False),
        "config3": (3840, 2160, 4, 7, 16, True),
        # config 4 is config 3 at 16 spp over 8 GPUs (run with --gpus 8 --workload config4: 2 spp per rank-step)
        "config4": (3840, 2160, 2, 7, 16, True),
        # config 5: 8K, LoD on (reference thresholds), 32^3 superchunks; 32 spp over 8 GPUs = 4 spp per rank-step
        "config5": (7680, 4320, 4, 7, 32, False),
    }
    return table[name]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="config2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--schedule", choices=["fused", "wavefront"], default="fused",
                    help="fused = one persistent kernel tracing complete paths (the product's default); wavefront = the "
                         "reference's own queue schedule, one segment of every path in flight per step (N = 1 only)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import brickmap_amd as bm

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # BM_BENCH_SHARE_GPU=1 (testing on a 1-GPU box only): every rank uses GPU 0 and the gather runs over gloo
    share_gpu = os.environ.get("BM_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    W, H, spp, max_bounces, n_super, streaming = workload(args.workload)
    G = 128 * n_super
    segments = max_bounces + 1
    spp_step = spp * world  # weak scaling: N x the samples per pixel and step, each rank the full frame with its own sample slice

    # ---- scene replica on this GPU (world build is CPU plumbing and is not timed)
    t0 = time.time()
    scene = bm.Scene(G, G, device=local_rank).generate()
    if streaming:
        scene.set_queue_capacity(1 << 20)
        scene.reset_residency()
    else:
        scene.preload_all()
    build_s = time.time() - t0
    cam = bm.Camera(position=(G / 2, G / 8, 0.8 * G), horizontal_angle=0.8, vertical_angle=-0.5).update()
    state = bm.State(W, H, device=local_rank)
    accum = state.blit_buffer
    if args.schedule == "wavefront":
        if world != 1:
            raise SystemExit("--schedule wavefront does not shard (replicas only): run it with --gpus 1")
        return bench_wavefront(args, bm, torch, np, scene, cam, accum, W, H, max_bounces, n_super, G, streaming, build_s)

    def params(step, flags=0):  # rank r owns samples [step*N*spp + r*spp, ... + spp) of every pixel
        return bm.FrameParams(W, H, spp=spp, sample_base=(step * world + rank) * spp, max_bounces=max_bounces, flags=flags)

    # N > 1: the reduction of frame i runs on RCCL's stream while frame i+1 is being traced (one reduction in flight)
    gatherer = bm.dist.FrameReducer(H, W, device=dev) if world > 1 else None

    def one_step(step):
        scene.render(cam, params(step), accum)
        if streaming:
            scene.process_load_queue()
        if gatherer is not None:
            gatherer.finish()      # frame step-1 is complete on rank 0
            gatherer.start(accum)  # snapshot + asynchronous sum-reduction of this frame
        return accum

    if streaming:  # reach streaming steady state before anything is timed
        for i in range(64):
            scene.render(cam, params(0), accum)
            if scene.process_load_queue() == 0:
                break
        accum.zero_()

    for i in range(args.warmup):
        one_step(i)
    if gatherer is not None:
        gatherer.finish()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    for i in range(args.steps):
        one_step(args.warmup + i)
    if gatherer is not None:
        gatherer.finish()  # the last frame's reduction is inside the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    kernel_ms = scene.render_times(args.steps)  # HIP events on the launch stream, one pair per launch
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share_gpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- algorithmic bytes of exactly the timed launches, from the instrumented kernel variant (not timed)
    scene.counters_reset()
    scratch = torch.zeros_like(accum)
    for i in range(args.steps):
        scene.render(cam, params(args.warmup + i, flags=bm.BM_FLAG_COUNTERS), scratch)
    cnt = scene.counters()
    local_rows = state.local_rows
    alg_bytes = 4 * cnt["index_loads"] + 64 * cnt["brick_tests"] + 16 * W * local_rows * args.steps
    avg_kernel_s = float(np.mean(kernel_ms)) * 1e-3
    achieved_gbs = alg_bytes / args.steps / avg_kernel_s / 1e9
    actual_rays = cnt["extend_rays"] + cnt["shadow_rays"]

    nominal_rays_per_step = W * H * spp_step * segments
    value = nominal_rays_per_step * args.steps / elapsed / 1e6

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    out = {
        "metric": "Mrays/sec (primary x spp x bounces) at 1080p 4-bounce",
        "value": round(value, 3),
        "unit": "Mrays/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic (SimplexNoise terrain built on the CPU by the product generator; canonical xorshift RNG streams)",
        "config": {
            "workload": f"BASELINE {args.workload}: {W}x{H}, {spp} spp per GPU-step (x{world} ranks = {spp_step} spp), "
                        f"{segments} segments/path, {n_super}^3 superchunks ({G}^3 voxels), "
                        + ("brick streaming at steady state" if streaming else "all bricks pre-loaded"),
            "width": W, "height": H, "spp_per_step": spp_step, "segments": segments, "world_voxels": G,
            "sharding": f"{world} x sample shards (every rank the full frame, its own {spp} of the {spp_step} samples) + RCCL sum-reduce to rank 0" if world > 1 else "single GPU",
            "camera": {"position": list(cam.position), "angles": [0.8, -0.5]},
            "world_build_s": round(build_s, 2),
        },
        "rays": {"nominal_per_step": nominal_rays_per_step, "actual_per_step_rank0": actual_rays / args.steps,
                 "actual_Mrays_s_rank0_kernel": round(actual_rays / args.steps / avg_kernel_s / 1e6, 2)},
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved_gbs, 2),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved_gbs / HBM_PEAK_GBS, 5),
            "traffic": pmc_traffic(args.workload),
            "kernel": "bm::trace_paths<false>",
            "kernel_ms_avg": round(float(np.mean(kernel_ms)), 4),
            "algorithmic_bytes_per_launch": alg_bytes / args.steps,
            "bytes_per_actual_ray": round(alg_bytes / max(actual_rays, 1), 1),
            "counts_per_launch": {k: v / args.steps for k, v in cnt.items()},
        },
    }

    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(W, H, spp, max_bounces, G, cam)
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def bench_wavefront(args, bm, torch, np, scene, cam, accum, W, H, max_bounces, n_super, G, streaming, build_s):
    """The reference's queue schedule (bm_wavefront_*): a step is one launch_kernels call = one segment of every path in
    the 2 Mi-slot queue.  Nominal rays of the metric = paths retired during the timed steps x segments per path."""
    Q = 2 * 1048576  # ray_queue_buffer_size, variables.h:61
    segments = max_bounces + 1
    wf = bm.Wavefront(scene, Q)
    p = bm.FrameParams(W, H, max_bounces=max_bounces)

    def one_step():
        wf.frame(cam, p, accum)
        if streaming:
            scene.process_load_queue()

    if streaming:
        for _ in range(64):
            wf.frame(cam, p, accum)
            if scene.process_load_queue() == 0:
                break
    for _ in range(max(args.warmup, 2 * segments)):  # the bounce mix of the queue reaches its steady state
        one_step()
    torch.cuda.synchronize()
    retired0 = float(accum[..., 3].sum(dtype=torch.float64).item())  # alpha counts terminated paths (kernel.cu:301,322)
    t_start = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    retired = float(accum[..., 3].sum(dtype=torch.float64).item()) - retired0
    value = retired * segments / elapsed / 1e6
    # ---- per-kernel durations and algorithmic bytes of the same steady state (separate, untimed frames)
    pc = bm.FrameParams(W, H, max_bounces=max_bounces, flags=bm.BM_FLAG_COUNTERS)
    wf.counters_reset()
    for _ in range(args.steps):
        wf.frame(cam, pc, accum)
    ce, cc = wf.counters("extend"), wf.counters("connect")
    times = []
    for _ in range(args.steps):
        wf.frame(cam, p, accum)
        times.append(wf.times())
    ms = {k: float(np.mean([t[k] for t in times])) for k in times[0]}
    # extend kernel: index words + bricks + its queue traffic: 48 B read (origin, direction, normal) and
    # 16 B written (normal, distance) per slot
    ext_bytes = (4 * ce["index_loads"] + 64 * ce["brick_tests"]) / args.steps + Q * (48 + 16)
    achieved = ext_bytes / (ms["extend"] * 1e-3) / 1e9
    actual = (ce["extend_rays"] + cc["shadow_rays"]) / args.steps
    out = {
        "metric": "Mrays/sec (primary x spp x bounces) at 1080p 4-bounce",
        "value": round(value, 3), "unit": "Mrays/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic (SimplexNoise terrain built on the CPU by the product generator; reference RNG streams)",
        "config": {
            "workload": f"BASELINE {args.workload} scene and camera, reference wavefront schedule: {W}x{H}, queue of {Q} rays, "
                        f"one segment of every path in flight per step, {segments} segments/path, {n_super}^3 superchunks, "
                        + ("brick streaming at steady state" if streaming else "all bricks pre-loaded"),
            "schedule": "wavefront", "width": W, "height": H, "queue_size": Q, "segments": segments, "world_voxels": G,
            "sharding": "single GPU (the queue schedule does not shard: replicas only)", "world_build_s": round(build_s, 2),
        },
        "rays": {"nominal_per_step": retired * segments / args.steps, "paths_retired_per_step": retired / args.steps,
                 "actual_per_step": actual, "actual_Mrays_s_frame": round(actual / (ms["total"] * 1e-3) / 1e6, 2)},
        "frame_ms": {k: round(v, 4) for k, v in ms.items()},
        "roofline": {
            "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": None, "kernel": "bm::wf_trace<false, false> (extend)", "kernel_ms_avg": round(ms["extend"], 4),
            "algorithmic_bytes_per_launch": ext_bytes,
            "counts_per_launch": {k: v / args.steps for k, v in ce.items()},
        },
    }
    print(json.dumps(out), flush=True)


def cpu_baseline(W, H, spp, max_bounces, G, cam):
    """The oracle's scalar C port of the same per-pixel path (oracle/oracle.c orc_render) on all host cores.
    Sample: whole frames of the same workload -- one calibration frame, then enough samples per pixel for
    roughly 3 s of wall time (tens of CPU-seconds on a many-core host)."""
    import oracle
    cores = os.cpu_count() or 1
    world = oracle.World(G, G, threads=cores)
    world.reset_device(True)
    ocam = oracle.make_camera(cam.position, cam.direction)
    _, _, _, t1 = world.render(ocam, oracle.make_frame(W, H, spp=1, max_bounces=max_bounces), want_dbg=False, threads=cores)
    n = max(1, min(64, int(3.0 / max(t1, 1e-3))))
    _, _, cnt, secs = world.render(ocam, oracle.make_frame(W, H, spp=n, max_bounces=max_bounces, sample_base=1), want_dbg=False, threads=cores)
    nominal = W * H * n * (max_bounces + 1)
    return {
        "value": round(nominal / secs / 1e6, 4),
        "unit": "Mrays/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{n} samples/pixel of the same {W}x{H} frame ({nominal} nominal rays, "
                  f"{cnt['extend_rays'] + cnt['shadow_rays']} actual) in {secs:.2f} s on {cores} threads",
    }


def pmc_traffic(workload):
    """HBM bytes per launch from the committed rocprofv3 PMC summary of this workload (separate --pmc passes,
    tools/pmc.sh), or None.  Units and correction as MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE and
    WRITE_SIZE are KiB; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, i.e. reports half of a 16 B/lane
    stream, so it is doubled (this kernel's traffic is dominated by the 16-byte brick / mask-record reads);
    WRITE_SIZE is taken as reported (uncalibrated per the guide)."""
    path = os.path.join(ROOT, "profiles", "pmc_summary.json")
    try:
        with open(path) as f:
            d = json.load(f)
        if d.get("workload") != workload:
            return None
        return int((2.0 * d["FETCH_SIZE_KiB"] + d["WRITE_SIZE_KiB"]) * 1024)
    except (OSError, KeyError, ValueError):
        return None


if __name__ == "__main__":
    main()
