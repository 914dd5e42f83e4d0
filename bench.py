#!/usr/bin/env python3
"""bench.py -- Mrays/s of the brickmap path-trace hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one FRAME of the path-trace kernel over this rank's shard of the workload: `spp` complete paths (primary + up to
max_bounces bounces, one shadow ray per hit) for every pixel of the shard, with its own constants (sample_base) and its own
ticket counters, scene resident in HBM.  Consecutive steps are issued the way the reference's frame loop issues them -- one after
the other on ONE stream (main.cpp:117-147) -- but as ONE launch of the persistent kernel per `--frames-per-launch` steps
(bm_render_frames, the "frame ring"): a wave that finds frame i's ticket counters used up finishes its own paths
and starts on frame i+1 by itself, so the end of a frame (the latency of the paths that started last: a sixth of a 1080p / 1-spp
frame) is covered by the beginning of the next one instead of an idle GPU.  Nothing is skipped or shared between steps: every
frame traces all its rays and adds them to the accumulation buffer like consecutive frames of the reference's accumulation
(kernel.cu:319-322,341-343).  `--frames-per-launch 1` is one launch per step (what rounds 1-5 timed); the N = 1 line reports that
figure next to the headline as `one_frame_per_launch`.

N = 1 runs BASELINE.json configs[1] itself: 1920x1080, 1 spp, 4-bounce (MAX_BOUNCES = 3 -> 4 segments), 8x8x8 superchunks, all
bricks resident; by default all K timed steps are one launch.

N > 1 STRONG-scales one fixed job with the north-star decomposition: the same 1080p / 4-segment frame at
MULTI_GPU_SPP = 8 samples per pixel, cut into interleaved 8-row bands (band b belongs to rank b % N); every rank traces all 8
samples of its rows into a packed float4 buffer (work items = (4x4 chunk, sample) pairs, BM_FLAG_SAMPLE_ITEMS, so the persistent
waves stay fed on 1/N of the pixels), five steps per launch by default, each into its own packed buffer of one allocation, and the
batch is gathered to rank 0 over RCCL/xGMI as ONE message per peer (brickmap_amd/dist.py FrameGatherer -> bm_gather_frames:
grouped send/recv, 33 MB / N per peer per step).  The gather of batch i overlaps the tracing of batch i+1 (one render stream, the
exchange on a side stream); every gather, including the last, completes inside the timed region.  Rates are per nominal ray, but
the N = 1 line is a DIFFERENT work shape (1 spp, pixel items; coherent neighbouring samples run ~20 % faster per ray), so a
scaling efficiency must not be computed against it: every N > 1 line carries `same_job_single_gpu` (rank 0 renders the line's own
8-spp job unsharded, untimed) and the N = 1 line carries the same figure as `multi_gpu_job_on_one_gpu` -- that is the denominator.
`--decomposition samples` keeps the alternative cut (every rank the full frame with its own samples, ONE sum-reduction after the
last step -- the buffers are additive), `--scaling weak` makes the job grow with N (N spp in total).

metric: Mrays/s = width * height * spp_total * segments / seconds  (nominal rays).
The line also carries `roofline` (algorithmic bytes of the timed launches / their HIP-event duration against the 8 TB/s HBM
peak) and, on rank 0 at N = 1, `cpu_baseline` (the oracle's scalar C port of the same path timed on the host cores).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
# what bounds trace_paths on each workload (profiles/r0x_pmc_summary_<workload>.json, profiles/r03_fetch_calibration.txt); {waves} = resident
# waves per SIMD of the instantiation that ran, asked from the library (bm_trace_waves_per_simd)
LIMITER = {
    "config2": "VALU issue at the occupancy the registers allow ({waves} waves per SIMD): the vector pipes are busy for most of the launch with ~21 of 64 lanes active per "
               "instruction (a wave's lanes are in different states; a pass costs the same with 15 or 64 of them); the scene (110 MiB) stays in L2 / Infinity Cache",
    "config3": "VALU issue ({waves} waves per SIMD, ~23 of 64 lanes), with 0.4 G single-sector reads reaching the fabric per launch (XCD-aware hand-out)",
    "config5": "VALU issue (79 G wave instructions at ~20 of 64 lanes) together with the fabric's request rate for single 64-byte sectors (every read of the walk is one "
               "sector; the GPU sustains ~48 G such requests/s); occupancy ({waves} waves per SIMD) is what hides the walk's dependent loads",
}
PROFILE_ROUNDS = ("r06", "r05", "r04", "r03")  # profiles/<round>_pmc_summary_<workload>.json is where roofline.traffic comes from (newest first)
TUNING_VARIABLES = ("BM_REFILL_MIN", "BM_XCD_HANDOUT", "BM_HELPERS", "BM_TRACE_BLOCKS_PER_CU")  # what bm_tuning_overrides reports: the library's A/B knobs
MULTI_FRAMES_PER_LAUNCH = 5  # N > 1: steps per launch and per exchange (the last batch's gather is not hidden by a next batch: keep it short)


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


def batches(first, count, per_launch):
    """[(first step, steps)] of `count` consecutive steps cut into launches of at most `per_launch` (as even as possible)."""
    if count <= 0:
        return []
    n = -(-count // max(1, per_launch))
    base, extra = divmod(count, n)
    out, at = [], first
    for i in range(n):
        size = base + (1 if i < extra else 0)
        out.append((at, size))
        at += size
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="config2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--decomposition", choices=["rows", "samples"], default="rows",
                    help="N > 1: rows = interleaved row bands + RCCL gather (north-star); samples = full frame per rank, one sum-reduction at the end")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1: strong = the fixed job (frame at --multi-gpu-spp samples) over N ranks; weak = N x spp samples in total")
    ap.add_argument("--multi-gpu-spp", type=int, default=8, help="samples per pixel of the strong-scaled job (N > 1)")
    ap.add_argument("--frames-per-launch", type=int, default=0,
                    help="consecutive steps issued as ONE launch of the persistent kernel (bm_render_frames, the frame ring): every step keeps its own "
                         "constants, ticket counters and rays; waves walk from a used-up frame to the next by themselves.  0 = default: all timed steps "
                         "(at most 256 per launch) for --gpus 1, %d per launch and per exchange for --gpus N > 1, 1 for streaming workloads (bricks are "
                         "serviced between launches).  1 = one launch per step, what rounds 1-5 timed" % MULTI_FRAMES_PER_LAUNCH)
    ap.add_argument("--spinup-ms", type=float, default=60.0,
                    help="untimed frames of the workload's shape rendered into a scratch buffer for this many ms of wall time before the warm-up steps: the GPU's "
                         "clocks take ~10 ms of load to come back up after the (CPU) world build; 0 = none")
    ap.add_argument("--verify", action="store_true",
                    help="N > 1: after the timed region rank 0 renders every step unsharded and compares it with the gathered / reduced frame")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the untimed extra measurements of the N = 1 line (one launch per frame, the 4-spp north-star shape, the shard predictions): "
                         "used when the run is profiled, so that rocprofv3's per-kernel averages cover the timed launches")
    ap.add_argument("--streaming-mode", choices=["overlapped", "blocking"], default="overlapped",
                    help="streaming workloads: overlapped = two request rings, the host never waits for the GPU; blocking = the reference's order")
    ap.add_argument("--schedule", choices=["fused", "wavefront"], default="fused",
                    help="fused = one persistent kernel tracing complete paths (the product's default); wavefront = the "
                         "reference's own queue schedule, one segment of every path in flight per step (N = 1 only)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` with no launcher: start the N ranks ourselves (one process per GPU; rank 0 prints the line)
        raise SystemExit(self_launch(args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist

    import brickmap_amd as bm

    # the library's tuning knobs are read from the environment: a stray variable on a lease would silently change what is timed.
    # Echoed into the line (config.env_overrides); BM_BENCH_STRICT=1 refuses to produce a number under any of them.
    overrides = bm.tuning_overrides()
    if overrides and os.environ.get("BM_BENCH_STRICT") == "1":
        raise SystemExit(f"bench.py: BM_BENCH_STRICT=1 and tuning overrides are set: {overrides}")

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
    # BM_BENCH_FORCE_DIST=1 (the 1-rank RCCL dry run of tests/test_gpu_dist.py): take the multi-GPU code path -- process group
    # on the "nccl" backend, row-band shard, (chunk, sample) work items, FrameGatherer / FrameReducer collectives on device
    # tensors, barrier, max-over-ranks reduction -- with a single rank, so that everything except the second rank has run
    multi = world > 1 or os.environ.get("BM_BENCH_FORCE_DIST") == "1"
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    W, H, spp, max_bounces, n_super, streaming = workload(args.workload)
    G = 128 * n_super
    segments = max_bounces + 1
    by_rows = multi and args.decomposition == "rows"
    # samples per pixel of one whole-job step, and of this rank's launch
    if not multi:
        spp_total = spp
    elif args.scaling == "strong":
        spp_total = max(args.multi_gpu_spp, 1) if args.workload == "config2" else spp * 8  # configs 4 / 5 name their totals (2 x 8, 4 x 8)
    else:
        spp_total = spp * world
    if multi and not by_rows and spp_total % world:
        raise SystemExit(f"--decomposition samples needs spp_total ({spp_total}) divisible by the number of ranks")
    spp_rank = spp_total if (by_rows or not multi) else spp_total // world
    band = bm.dist.DEFAULT_BAND_ROWS

    # ---- scene replica on this GPU (world build is CPU plumbing and is not timed)
    t0 = time.time()
    scene = bm.Scene(G, G, device=local_rank)
    if streaming:
        scene.set_queue_capacity(1 << 20)
    scene.generate()
    if streaming:
        scene.reset_residency()
        scene.set_streaming_mode(args.streaming_mode == "overlapped")
    else:
        scene.preload_all()
    build_s = time.time() - t0
    cam = bm.Camera(position=(G / 2, G / 8, 0.8 * G), horizontal_angle=0.8, vertical_angle=-0.5).update()
    if by_rows:
        state = bm.State(W, H, device=local_rank, band_rows=band, shard_rank=rank, shard_count=world)
    else:
        state = bm.State(W, H, device=local_rank)
    accum = state.blit_buffer
    if args.schedule == "wavefront":
        if world != 1:
            raise SystemExit("--schedule wavefront does not shard (replicas only): run it with --gpus 1")
        return bench_wavefront(args, bm, torch, np, scene, cam, accum, W, H, max_bounces, n_super, G, streaming, build_s)

    # ---- how the steps are issued: the frame ring (module docstring).  Streaming workloads service their brick requests between
    # launches (bm_scene_process_load_queue): one step per launch there.
    # The ring pays where a frame's end is a visible part of it: it gives back ~0.2 ms per frame, and the multi-frame instantiations spill a
    # few scalars more than the one-frame kernel -- a win at 1080p up to 8 spp (1.00 -> 0.77 ms, 3.24 -> 2.98, 6.22 -> 6.1), nothing on the
    # 8K / 4K workloads (config 5: 96.9 ms per frame in a ring of five against 96.1 as single launches).
    ring_pays = W * state.local_rows * spp_rank <= (1 << 24)
    if streaming:
        per_launch = 1
    elif args.frames_per_launch > 0:
        per_launch = min(args.frames_per_launch, 256)
    elif not ring_pays:
        per_launch = 1
    elif by_rows:
        per_launch = min(MULTI_FRAMES_PER_LAUNCH, max(args.steps, 1))
    else:
        per_launch = min(max(args.steps, 1), 256)
    item_flag = bm.BM_FLAG_SAMPLE_ITEMS if multi else 0  # (chunk, sample) work items keep 1/N-of-the-pixels shards fed
    if not by_rows and bm.frame_plan(bm.FrameParams(W, H, spp=spp_rank, max_bounces=max_bounces, flags=item_flag))["ordered"]:
        per_launch = 1  # (BM_HELPERS=0 runs: ordered frames write pixels back with plain stores -- the frames of a launch cannot share the buffer)

    def params(step, flags=0):
        if by_rows:  # rank r owns the bands b with b % N == r and traces every sample of the step for them
            return bm.FrameParams(W, H, spp=spp_rank, sample_base=step * spp_total, max_bounces=max_bounces, flags=flags | item_flag,
                                  band_rows=band, shard_rank=rank, shard_count=world)
        # sample shards: rank r owns samples [step*spp_total + r*spp_rank, ... + spp_rank) of every pixel
        return bm.FrameParams(W, H, spp=spp_rank, sample_base=step * spp_total + rank * spp_rank, max_bounces=max_bounces, flags=flags | item_flag)

    # rows: every step of a launch goes into its own packed buffer of ONE allocation (slot = position in the launch), and the batch is
    # gathered as one message per peer while the next batch is traced (one gather in flight); the buffers keep accumulating from
    # batch to batch.  samples: ONE sum-reduction after the last step (the per-rank buffers are additive).
    batch_buf = torch.zeros((per_launch,) + tuple(accum.shape), dtype=torch.float32, device=dev) if by_rows else None
    gatherer = bm.dist.FrameGatherer(H, W, band_rows=band, device=dev, force_collective=True, frames=per_launch) if by_rows else None
    reducer = bm.dist.FrameReducer(H, W, device=dev, force_collective=True) if (multi and not by_rows) else None

    exchange_error = None
    if multi and not share_gpu and os.environ.get("BM_DIST_TORCH", "0") != "1":
        # the exchange of an RCCL group is the C-ABI's (bm_comm_create + bm_comm_selftest ran inside the constructor above, on every
        # rank).  If it could not be made, say so LOUDLY, with RCCL's words -- on stderr and in the line (`ranks.exchange_error`) -- and
        # go on with torch.distributed's gather over the same RCCL, so that a multi-GPU run still yields its number;
        # BM_BENCH_STRICT=1 makes it the run's failure instead.
        ex = gatherer if gatherer is not None else reducer
        if getattr(ex, "comm", None) is None:
            exchange_error = (f"rank {rank}/{world}: the C-ABI RCCL exchange failed its start-up self-test (bm_comm_create / bm_comm_selftest): "
                              f"{bm.dist.last_comm_error or 'no error text'}")
            print("bench.py: " + exchange_error + " -- falling back to torch.distributed's exchange", file=sys.stderr, flush=True)
            if os.environ.get("BM_BENCH_STRICT") == "1":
                raise SystemExit("bench.py " + exchange_error)

    gathered = {}  # --verify: the last gathered batch (rank 0): slot k holds everything step-slot k accumulated

    def keep(batch):
        if args.verify and batch is not None:
            gathered["batch"] = batch.clone() if batch.dim() == 4 else batch.clone().unsqueeze(0)  # (a one-frame gatherer hands out [H, W, 4])

    def issue(first, count):
        """`count` consecutive steps starting at step `first`: one launch (streaming: one launch + request servicing)"""
        ps = [params(first + i) for i in range(count)]
        if by_rows:
            scene.render_frames(cam, ps, [batch_buf[k] for k in range(count)])
            if streaming:
                scene.process_load_queue()
            keep(gatherer.finish(wait=args.verify))  # the previous batch is complete on rank 0 (only --verify looks at it here)
            gatherer.start(batch_buf)                # snapshot + asynchronous gather of this batch's packed bands
        elif count == 1:
            scene.render(cam, ps[0], accum)
            if streaming:
                scene.process_load_queue()
        else:
            scene.render_frames(cam, ps, accum)

    def reach_steady_state(frame_params, buffer):
        """render + service until nothing is requested any more.  Overlapped servicing reports, at call i, what frame i-1 asked for, and
        a brick asked for in frame k is resident from frame k+2 on (it may then uncover others): three quiet calls in a row."""
        quiet, need = 0, (3 if args.streaming_mode == "overlapped" else 1)
        for i in range(256):
            scene.render(cam, frame_params, buffer)
            quiet = quiet + 1 if scene.process_load_queue() == 0 else 0
            if quiet >= need:
                return i + 1
        raise SystemExit("bench.py: streaming did not reach a steady state")

    if streaming:  # reach streaming steady state before anything is timed
        reach_steady_state(params(0), accum)
        accum.zero_()

    # ---- power state.  The world build is seconds of CPU work with an idle GPU, and the clocks take ~10 ms of load to come back up:
    # with 3 warm-up steps (2.4 ms) the first timed launch ran 3-4 % slow (20 steps: 0.794 ms per step after 3 warm-up steps, 0.767 after
    # 20, 0.765 after 60).  Untimed frames of the workload's own shape into a scratch buffer, for --spinup-ms of wall time, before the W
    # warm-up steps: what is timed is the kernel, not the governor.  (Nothing here touches the timed region or its buffers.)
    spun_ms = 0.0
    launch_frames_log = []  # frames of every trace_paths launch of this process up to the end of the timed region, in order (resident workloads)
    if args.spinup_ms > 0 and not streaming:
        spin_buf = torch.zeros_like(accum)
        t_spin = time.perf_counter()
        k = 0
        while (time.perf_counter() - t_spin) * 1e3 < args.spinup_ms:
            n_spin = min(per_launch, 8)
            ps = [params(1 << 20 | (k + i)) for i in range(n_spin)]
            if n_spin > 1:
                scene.render_frames(cam, ps, spin_buf)
            else:
                scene.render(cam, ps[0], spin_buf)
            torch.cuda.synchronize()
            k += n_spin
            launch_frames_log.append(n_spin)
        spun_ms = (time.perf_counter() - t_spin) * 1e3
        del spin_buf

    for first, count in batches(0, args.warmup, per_launch):
        issue(first, count)
    if gatherer is not None:
        keep(gatherer.finish())
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    timed = batches(args.warmup, args.steps, per_launch)
    t_start = time.perf_counter()
    for first, count in timed:
        issue(first, count)
    if gatherer is not None:
        keep(gatherer.finish())  # the last batch's gather is inside the timed region
    if reducer is not None:
        reducer.start(accum)
        reduced = reducer.finish()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    launch_ms = scene.render_times(len(timed))  # HIP events on the launch stream, one pair per launch
    launch_frames_log += [c for _, c in batches(0, args.warmup, per_launch)] + [c for _, c in timed]
    launch_ms_log = scene.render_times(len(launch_frames_log)) if (not streaming and len(launch_frames_log) <= 256) else None
    kernel_ms_per_step = float(np.sum(launch_ms)) / args.steps
    ranks_info = None
    if multi:
        # every rank's own clock and device, for the line: an imbalance (or two ranks on one GPU) shows up in SCALE_rNN.json
        ex = gatherer if gatherer is not None else reducer
        comm_world = ex.comm.info()[1] if getattr(ex, "comm", None) is not None else None
        mine = {"rank": rank, "device": torch.cuda.get_device_name(local_rank), "device_index": local_rank,
                "pci_bus_id": getattr(torch.cuda.get_device_properties(local_rank), "pci_bus_id", None),
                "ms_per_step": round(elapsed / args.steps * 1e3, 4), "kernel_ms_per_step": round(kernel_ms_per_step, 4)}
        every = [None] * world
        dist.all_gather_object(every, mine)
        ranks_info = {"process_group_world": dist.get_world_size(), "process_group_backend": dist.get_backend(),
                      "exchange_error": exchange_error,  # None: the C-ABI exchange passed its self-test on this rank (a failure on any rank makes all fall back)
                      "communicator_world": comm_world,  # bm_comm_info of the C-ABI communicator the frames travelled through (None: torch.distributed's exchange)
                      "ms_per_step": [e["ms_per_step"] for e in every], "kernel_ms_per_step": [e["kernel_ms_per_step"] for e in every],
                      "devices": [f'{e["device"]} #{e["device_index"]}' + (f' {e["pci_bus_id"]}' if e["pci_bus_id"] is not None else "") for e in every]}
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share_gpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- --verify: the frames rank 0 holds now must be the frames of one GPU rendering every sample of every step
    verified = None
    if args.verify and multi and rank == 0:
        want = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
        if streaming and by_rows:  # this rank has streamed in what ITS rows see: reach the whole frame's steady state before rendering it
            reach_steady_state(bm.FrameParams(W, H, spp=1, max_bounces=max_bounces), want)
            want.zero_()
        for i in range(args.warmup + args.steps):
            scene.render(cam, bm.FrameParams(W, H, spp=spp_total, sample_base=i * spp_total, max_bounces=max_bounces), want)
        torch.cuda.synchronize()
        have = (gathered["batch"].sum(dim=0) if by_rows else reduced).to(dev)  # (the batch's slots together hold every step)
        err = float((have - want).abs().max() / want.abs().max())
        verified = {"frames": args.warmup + args.steps, "max_rel_err": err}
        if streaming:
            # A streaming scene never stops asking: every step's bounce rays go new ways, and a brick that is not resident yet is
            # traced as solid (voxel.cuh:228-245).  The ranks rendered their steps with the residency THEY had, rank 0 renders the
            # reference frame now, with more: pixels whose paths met such a brick differ, legitimately.  What the exchange must not do
            # is lose or duplicate a path -- alpha counts terminated paths, exactly -- or damage the other pixels.
            if not torch.equal(have[..., 3], want[..., 3]):
                raise SystemExit(f"--verify: the {world}-rank frame holds other terminated-path counts than the single-GPU frame")
            differs = ((have[..., :3] - want[..., :3]).abs() > 1e-5 * want[..., :3].abs().max()).any(dim=-1)
            verified.update(terminated_paths_exact=True, pixels_differing=int(differs.sum().item()), fraction_differing=round(float(differs.float().mean().item()), 6),
                            note="streaming: pixels whose paths met a brick that was resident in one render and not yet in the other differ; terminated-path counts are exact")
            if verified["fraction_differing"] > 0.02:
                raise SystemExit(f"--verify: {verified['fraction_differing']:.2%} of the pixels of the {world}-rank frame differ from the single-GPU frame")
        elif not err < 1e-5:  # bit-identical paths; only the order of the float additions differs
            raise SystemExit(f"--verify: the {world}-rank frame differs from the single-GPU frame (relative error {err:g})")

    # ---- algorithmic bytes of exactly the timed steps, from the instrumented kernel variant (not timed; same launches, same scheduling)
    scene.counters_reset()
    scratch = torch.zeros_like(accum)
    for first, count in timed:
        ps = [params(first + i, flags=bm.BM_FLAG_COUNTERS) for i in range(count)]
        if count == 1:
            scene.render(cam, ps[0], scratch)
        else:
            scene.render_frames(cam, ps, scratch)
    cnt = scene.counters()
    local_rows = state.local_rows
    alg_bytes = 4 * cnt["index_loads"] + 64 * cnt["brick_tests"] + 16 * W * local_rows * args.steps
    kernel_s = float(np.sum(launch_ms)) * 1e-3  # all timed launches
    achieved_gbs = alg_bytes / kernel_s / 1e9
    actual_rays = cnt["extend_rays"] + cnt["shadow_rays"]
    bytes_per_step = alg_bytes / args.steps

    nominal_rays_per_step = W * H * spp_total * segments
    value = nominal_rays_per_step * args.steps / elapsed / 1e6
    extras = not args.no_extras

    def measure(render_one, frames, reps=2):
        """wall seconds per frame of `frames` frames issued by render_one(i), best of `reps` (after one untimed pass)"""
        best = None
        for rep in range(reps + 1):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(frames):
                render_one(rep * frames + i)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / frames
            if rep > 0:
                best = dt if best is None else min(best, dt)
        return best

    # ---- one launch per frame on the same stream (what rounds 1-5 reported as the headline), for the record: N = 1 / resident only
    single = None
    if not multi and not streaming and extras and per_launch > 1:
        try:  # an extra: whatever goes wrong here must not cost the headline measurement above
            s1 = measure(lambda i: scene.render(cam, params(args.warmup + i), scratch), args.steps)
            k1 = float(np.mean(scene.render_times(args.steps))) * 1e-3
            single = {"ms_per_step": round(s1 * 1e3, 4), "value": round(nominal_rays_per_step / s1 / 1e6, 3), "kernel_ms_avg": round(k1 * 1e3, 4),
                      "roofline_frac": round(bytes_per_step / k1 / 1e9 / HBM_PEAK_GBS, 5),
                      "note": "the same steps as one launch each (bm_render_frame) on the same stream: the end of every frame -- the latency of its last paths -- "
                              "is an idle GPU; rounds 1-5 reported this as `value`"}
        except Exception as e:  # noqa: BLE001
            torch.cuda.synchronize()
            single = {"error": repr(e)}

    # ---- the north-star target shape (BASELINE.json: "1080p, 4 spp, 4-bounce"), N = 1 / config 2 only, reported next to the
    # headline, never as `value`: the same frame at 4 samples per pixel with (chunk, sample) work items -- one launch per frame, and
    # consecutive frames as one launch (as many as the headline's steps, at most 20)
    target4 = None
    if not multi and args.workload == "config2" and not streaming and extras:
        n4 = 5
        p4 = lambda i, flags=0: bm.FrameParams(W, H, spp=4, sample_base=1000 + 4 * i, max_bounces=max_bounces, flags=flags | bm.BM_FLAG_SAMPLE_ITEMS)
        s4 = measure(lambda i: scene.render(cam, p4(i), scratch), n4)
        k4 = float(np.mean(scene.render_times(n4))) * 1e-3
        scene.counters_reset()
        scene.render(cam, p4(2, flags=bm.BM_FLAG_COUNTERS), scratch)
        c4 = scene.counters()
        b4 = 4 * c4["index_loads"] + 64 * c4["brick_tests"] + 16 * W * H
        nr = max(n4, min(args.steps, 20))  # (as many frames per launch as the headline's steps)
        r4 = measure(lambda i: scene.render_frames(cam, [p4(nr * i + k) for k in range(nr)], scratch), 1) / nr
        target4 = {"workload": f"{W}x{H}, 4 spp, {segments} segments/path, (chunk, sample) work items", "ms_per_step": round(s4 * 1e3, 4),
                   "Mrays_s": round(W * H * 4 * segments / s4 / 1e6, 1), "kernel_ms_avg": round(k4 * 1e3, 4),
                   "roofline_frac": round(b4 / k4 / 1e9 / HBM_PEAK_GBS, 5),
                   "frame_ring": {"frames_per_launch": nr, "ms_per_step": round(r4 * 1e3, 4), "Mrays_s": round(W * H * 4 * segments / r4 / 1e6, 1),
                                  "roofline_frac": round(b4 / r4 / 1e9 / HBM_PEAK_GBS, 5)}}

    # ---- the denominator of a scaling efficiency: the N > 1 lines strong-scale ONE fixed job (the frame at spp_total samples,
    # (chunk, sample) work items); rank 0 renders exactly that job unsharded on its GPU, untimed, so that value(N) can be
    # compared with the single-GPU time of ITS OWN job and not with the N = 1 line's 1-spp frame.  Measured the way the N > 1 ranks
    # issue their steps (MULTI_FRAMES_PER_LAUNCH steps per launch) and as one launch per step.  The N = 1 line carries the same
    # measurement for its workload's multi-GPU job (config 2: the default 8-spp job).
    same_job = None
    job_spp = spp_total if multi else (max(args.multi_gpu_spp, 1) if args.workload == "config2" else spp * 8)
    nj = MULTI_FRAMES_PER_LAUNCH

    def steps_per_launch(rows, samples):
        """how a rank issues a job of `rows` x W pixels at `samples` spp: the frame ring where it pays (see ring_pays above)"""
        return nj if W * rows * samples <= (1 << 24) else 1

    if rank == 0 and not streaming and extras:
        try:
            whole = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
            pj = lambda i: bm.FrameParams(W, H, spp=job_spp, sample_base=5000 + i * job_spp, max_bounces=max_bounces, flags=bm.BM_FLAG_SAMPLE_ITEMS)
            reps = 2 if W * H * job_spp < 100e6 else 1
            s_one = measure(lambda i: scene.render(cam, pj(i), whole), 3, reps=reps)
            k_one = float(np.mean(scene.render_times(3)))
            nw = steps_per_launch(H, job_spp)
            s_ring = measure(lambda i: scene.render_frames(cam, [pj(nw * i + k) for k in range(nw)], whole), 1, reps=reps) / nw if nw > 1 else s_one
            same_job = {"workload": f"{W}x{H}, {job_spp} spp, {segments} segments/path, (chunk, sample) work items, unsharded on one GPU",
                        "ms_per_step": round(s_ring * 1e3, 4), "Mrays_s": round(W * H * job_spp * segments / s_ring / 1e6, 1), "frames_per_launch": nw,
                        "one_frame_per_launch": {"ms_per_step": round(s_one * 1e3, 4), "kernel_ms_avg": round(k_one, 4)}}
            del whole
        except Exception as e:  # noqa: BLE001 -- an extra: must not cost the headline measurement
            torch.cuda.synchronize()
            same_job = {"error": repr(e)}

    # ---- what the N > 1 lines should come out at, measured on THIS GPU (N = 1 line only, untimed extra): every rank's 1/N shard of
    # the job (interleaved row bands, (chunk, sample) items; the slowest rank counts), consecutive steps on ONE stream, issued as the
    # ranks issue them (MULTI_FRAMES_PER_LAUNCH steps per launch).  predicted_speedup = unsharded job time / shard time, both issued
    # the same way: the kernels alone, before the exchange (4.1 MB per peer per step at N = 8 for the 1080p job, one message per
    # batch, overlapped with the next batch).  The first real SCALE run is checked against it.  config 2: N = 2 / 4 / 8; the big
    # workloads: N = 8 (what BASELINE.json configs 4 / 5 name).
    shard_pred = None
    if not multi and not streaming and extras and same_job is not None and "ms_per_step" in same_job:
        try:
            shard_pred = {"job": same_job["workload"], "frames_per_launch": steps_per_launch(state.local_rows // 8, job_spp), "unsharded_ms_per_step": same_job["ms_per_step"],
                          "unsharded_ms_per_step_one_frame_per_launch": same_job["one_frame_per_launch"]["ms_per_step"],
                          "shard_ms_per_step": {}, "predicted_speedup": {}, "shard_ms_per_step_one_frame_per_launch": {}, "predicted_speedup_one_frame_per_launch": {},
                          "per_rank_ms_per_step": {}, "band_rows": band,
                          "note": "every rank's shard of the job timed on this GPU, one after the other, on ONE stream, no exchange; shard_* = the SLOWEST rank.  "
                                  "value(N) / (the unsharded job's rate on one GPU) of a --gpus N run should come out near predicted_speedup[N]"}
            for n_ranks in ((2, 4, 8) if args.workload == "config2" else (8,)):
                ring_ms, one_ms = [], []
                for r in range(n_ranks):
                    st = bm.State(W, H, device=local_rank, band_rows=band, shard_rank=r, shard_count=n_ranks)
                    ps = lambda i: bm.FrameParams(W, H, spp=job_spp, sample_base=7000 + i * job_spp, max_bounces=max_bounces, flags=bm.BM_FLAG_SAMPLE_ITEMS,
                                                  band_rows=band, shard_rank=r, shard_count=n_ranks)
                    reps = 2 if W * H * job_spp < 100e6 else 1
                    one_ms.append(measure(lambda i: scene.render(cam, ps(i), st.blit_buffer), 4, reps=reps) * 1e3)
                    ns = steps_per_launch(st.local_rows, job_spp)
                    ring_ms.append(measure(lambda i: scene.render_frames(cam, [ps(ns * i + k) for k in range(ns)], st.blit_buffer), 1, reps=reps) / ns * 1e3 if ns > 1 else one_ms[-1])
                    del st
                key = str(n_ranks)
                shard_pred["per_rank_ms_per_step"][key] = [round(t, 4) for t in ring_ms]
                shard_pred["shard_ms_per_step"][key] = round(max(ring_ms), 4)
                shard_pred["predicted_speedup"][key] = round(same_job["ms_per_step"] / max(ring_ms), 3)
                shard_pred["shard_ms_per_step_one_frame_per_launch"][key] = round(max(one_ms), 4)
                shard_pred["predicted_speedup_one_frame_per_launch"][key] = round(same_job["one_frame_per_launch"]["ms_per_step"] / max(one_ms), 3)
        except Exception as e:  # noqa: BLE001 -- an extra: must not cost the headline measurement
            torch.cuda.synchronize()
            shard_pred = {"error": repr(e)}

    if multi:
        dist.barrier()  # rank 0's untimed extras are done: every rank leaves the process group together
    if rank != 0:
        if multi:
            dist.destroy_process_group()
        return

    # the instantiation that ran and what it keeps resident, from the library's own plan of the timed frames
    plan = bm.frame_plan(params(args.warmup))
    ring = (2 if not by_rows or per_launch > 1 else 1) if (per_launch > 1 and any(c > 1 for _, c in timed)) else 0  # (bench.py's launches are uniform: one view, stepping sample_base / buffers)
    waves = bm.trace_waves_per_simd(instrumented=False, xcd_handout=bool(plan["xcd_handout"]), helpers=bool(plan["helpers"]), device=local_rank)
    kernel_name = "bm::trace_paths<false, %s, %s, %d>" % ("true" if plan["xcd_handout"] else "false", "true" if plan["helpers"] else "false", ring)
    out = {
        "metric": "Mrays/sec (primary x spp x bounces) at 1080p 4-bounce",
        "value": round(value, 3),
        "unit": "Mrays/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": args.scaling if multi else "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic (SimplexNoise terrain built on the CPU by the product generator; canonical xorshift RNG streams)",
        "config": {
            "workload": f"BASELINE {args.workload}: {W}x{H}, {spp_total} spp per step, "
                        f"{segments} segments/path, {n_super}^3 superchunks ({G}^3 voxels), "
                        + (f"brick streaming at steady state ({args.streaming_mode})" if streaming else "all bricks pre-loaded")
                        + ("" if world == 1 else f"; the fixed {spp_total}-spp job strong-scaled over {world} ranks" if args.scaling == "strong"
                           else f"; weak scaling, {spp_total // world} spp per rank"),
            "width": W, "height": H, "spp_per_step": spp_total, "segments": segments, "world_voxels": G,
            "step_issue": ("one launch per step (bm_render_frame) + bm_scene_process_load_queue" if per_launch == 1 else
                           f"frame ring: {[c for _, c in timed]} consecutive steps per launch of the persistent kernel (bm_render_frames) on one stream; every step is a "
                           "complete frame with its own constants, ticket counters and rays -- waves walk from a used-up frame to the next by themselves"),
            "frames_per_launch": per_launch,
            "spinup_ms": round(spun_ms, 1),  # untimed frames before the warm-up steps (GPU clocks after the idle world build), see --spinup-ms
            "sharding": ("single GPU" if world == 1 else
                         f"{world} x interleaved {band}-row bands, every rank all {spp_rank} samples of its rows ((chunk, sample) work items) "
                         f"+ RCCL gather of the packed bands to rank 0, one message per peer per batch of {per_launch} steps, one gather in flight" if by_rows else
                         f"{world} x sample shards (every rank the full frame, {spp_rank} of the {spp_total} samples) + one RCCL sum-reduce to rank 0 after the last step"),
            "exchange": (None if not multi else ("C-ABI bm_gather_frames / bm_reduce_frame over RCCL (csrc/comm.hip)" if getattr(gatherer or reducer, "comm", None) is not None
                                                     else "torch.distributed gather / reduce")),
            "frame_mode": ("ordered sums (helper lanes off)" if not plan["helpers"] else
                           "production default: shadow rays on helper lanes, radiance added with float atomics (like the reference's connect, kernel.cu:341-343); "
                           "multi-sample frames as (chunk, sample) work items; BM_FLAG_ORDERED frames -- what the parity suite compares bit for bit -- trace the same rays"),
            "frame_plan": plan,
            "env_overrides": overrides,  # the library's tuning knobs found in the environment ({}: the product's own rules)
            "camera": {"position": list(cam.position), "angles": [0.8, -0.5]},
            "world_build_s": round(build_s, 2),
        },
        "rays": {"nominal_per_step": nominal_rays_per_step, "actual_per_step_rank0": actual_rays / args.steps,
                 "actual_Mrays_s_rank0_kernel": round(actual_rays / kernel_s / 1e6, 2)},
        "roofline": {
            "bound": "hbm",
            # what `achieved` is: SURVEY.md 8(d)'s ALGORITHMIC bytes -- the traffic of the REFERENCE's walk over the same rays
            # (4 B per cell it would load + 64 B per brick test + 16 B per pixel) -- of the timed launches, divided by their
            # duration (HIP events on the launch stream).  The kernel itself reads far less (one cube-field byte per STOP of the walk,
            # index words only at candidates: see `traffic`, from the PMC counters); `limiter` names what actually bounds it.
            "achieved_is": "reference-equivalent (algorithmic) bytes of the timed launches / their duration",
            **limiter_of(args.workload, waves),
            "achieved": round(achieved_gbs, 2),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved_gbs / HBM_PEAK_GBS, 5),
            **counter_figures(args.workload, kernel_ms_per_step * 1e-3, args.steps / len(timed)),
            "kernel": kernel_name,
            "waves_per_simd": waves,
            "launches": len(timed), "frames_per_launch": [c for _, c in timed],
            "kernel_ms_avg": round(float(np.mean(launch_ms)), 4),   # average duration of one timed LAUNCH of `kernel` (what rocprofv3 --stats reports for it)
            "kernel_ms_per_step": round(kernel_ms_per_step, 4),
            # every launch of `kernel` this process made up to the end of the timed region (spin-up, warm-up, timed), so that the
            # per-kernel average of `rocprofv3 --stats` over a --no-extras run can be re-derived: it averages launches of different frame
            # counts; sum(frames) x algorithmic_bytes_per_step / sum(ms) is the rate over all of them, the last `launches` entries are the timed ones
            "launch_log": ({"frames": launch_frames_log, "ms": [round(float(t), 4) for t in launch_ms_log]} if launch_ms_log is not None else None),
            "algorithmic_bytes_per_launch": alg_bytes / len(timed),
            "algorithmic_bytes_per_step": bytes_per_step,
            "bytes_per_actual_ray": round(alg_bytes / max(actual_rays, 1), 1),
            "counts_per_step": {k: v / args.steps for k, v in cnt.items()},
        },
    }
    if single is not None:
        out["one_frame_per_launch"] = single
    if target4 is not None:
        out["north_star_4spp"] = target4
    if ranks_info is not None:
        out["ranks"] = ranks_info
    if verified is not None:
        out["verified_against_single_gpu"] = verified
    if same_job is not None:
        out["same_job_single_gpu" if multi else "multi_gpu_job_on_one_gpu"] = same_job
    if shard_pred is not None:
        out["multi_gpu_prediction"] = shard_pred
    if not multi and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(W, H, max_bounces, G, cam)
    print(json.dumps(out), flush=True)
    if multi:
        dist.destroy_process_group()


def self_launch(n_gpus):
    """Re-run this command line under torch.distributed.run: one rank per GPU on this node, static rendezvous on 127.0.0.1 (the
    container's hostname may not resolve), a free port, HSA_ENABLE_IPC_MODE_LEGACY=0 for RCCL's dmabuf IPC.  The children's
    stdout / stderr pass through, so rank 0's JSON line is this process's; returns the launcher's exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    env["BM_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


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


def host_cpu():
    """(model name, physical cores, logical CPUs) of this host, from /proc/cpuinfo."""
    model, cores = "unknown", set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None:
                cores.add((phys, core))
                phys = core = None
        if phys is not None:
            cores.add((phys, core))
    except OSError:
        pass
    logical = os.cpu_count() or 1
    return model, (len(cores) or logical), logical


def cpu_quota():
    """CPUs this process may actually use: the cgroup CPU quota (cpu.max / cfs_quota) if there is one, else the affinity mask."""
    allowed = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            allowed = min(allowed, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                allowed = min(allowed, max(1, q // per))
        except (OSError, ValueError):
            pass
    return allowed


def cpu_baseline(W, H, max_bounces, G, cam):
    """The oracle's scalar C port of the same per-pixel path (oracle/oracle.c orc_render), timed on the host: one thread
    per USABLE core (physical cores, capped by the cgroup CPU quota; tile-granular dynamic scheduling), plus a 1-thread
    figure.
    Sample: whole frames of the same workload -- one calibration frame, then enough samples per pixel for roughly 3 s of
    wall time on all cores; 1 thread: every 8th 8-row band of one frame (an eighth of the pixels, spread over the image)."""
    import oracle
    model, physical, logical = host_cpu()
    quota = cpu_quota()  # the GPU boxes run in a cgroup with a CPU quota far below the core count: more threads only get throttled
    threads = max(1, min(physical, quota, 256))
    world = oracle.World(G, G, threads=logical)
    world.reset_device(True)
    ocam = oracle.make_camera(cam.position, cam.direction)
    _, _, _, t1 = world.render(ocam, oracle.make_frame(W, H, spp=1, max_bounces=max_bounces), want_dbg=False, threads=threads)
    n = max(1, min(64, int(3.0 / max(t1, 1e-3))))
    _, _, cnt, secs = world.render(ocam, oracle.make_frame(W, H, spp=n, max_bounces=max_bounces, sample_base=1), want_dbg=False, threads=threads)
    nominal = W * H * n * (max_bounces + 1)
    strip = oracle.make_frame(W, H, spp=1, max_bounces=max_bounces, band_rows=8, shard_rank=0, shard_count=8)
    strip_rows = sum(1 for y in range(H) if (y // 8) % 8 == 0)
    _, _, cnt1, secs1 = world.render(ocam, strip, want_dbg=False, threads=1)
    nominal1 = W * strip_rows * (max_bounces + 1)
    return {
        "value": round(nominal / secs / 1e6, 4),
        "unit": "Mrays/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{n} samples/pixel of the same {W}x{H} frame ({nominal} nominal rays, "
                  f"{cnt['extend_rays'] + cnt['shadow_rays']} actual) in {secs:.2f} s on {threads} threads "
                  f"(one per usable core: {physical} physical cores, cgroup CPU quota {quota})",
        "cpu_model": model, "physical_cores": physical, "logical_cpus": logical, "cpu_quota": quota,
        "one_thread": {"value": round(nominal1 / secs1 / 1e6, 4), "unit": "Mrays/s",
                       "sample": f"every 8th 8-row band of one frame ({nominal1} nominal rays, {cnt1['extend_rays'] + cnt1['shadow_rays']} actual) in {secs1:.2f} s"},
    }


def counter_figures(workload, kernel_s_per_step, steps_per_launch=1.0):
    """pmc_traffic() plus the north-star's own measure: `frac_by_counters` = fabric-side bytes per step (committed counter passes)
    / THIS run's kernel time per step / 8 TB/s -- next to `frac`, which prices the reference-equivalent bytes.  `traffic` is per
    launch, like `achieved`'s bytes (traffic_per_step x the steps of a timed launch)."""
    t = pmc_traffic(workload)
    per_step = t.pop("traffic_per_step")
    t["traffic"] = int(per_step * steps_per_launch) if per_step else None
    t["traffic_per_step"] = per_step
    t["frac_by_counters"] = round(per_step / kernel_s_per_step / 1e9 / HBM_PEAK_GBS, 5) if per_step else None
    t["traffic_age"] = traffic_age(workload)
    return t


def pmc_summary_path(workload):
    for rnd in PROFILE_ROUNDS:
        path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_summary_{workload}.json")
        if os.path.exists(path):
            return path
    return None


def git_head():
    """short commit of the tree bench.py runs from: git where there is a repository, else the note a gpurun snapshot carries (scratch/HEAD)"""
    try:
        r = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True, timeout=10)
        if r.returncode == 0 and r.stdout.strip():
            return r.stdout.strip()
    except (OSError, subprocess.SubprocessError):
        pass
    try:
        return open(os.path.join(ROOT, "scratch", "HEAD")).read().strip() or None
    except OSError:
        return None


def traffic_age(workload):
    """How fresh the counter figures are: the commit the PMC summary was collected at (recorded in the file by
    tools/profile_workload.py) against the commit this run is from -- a stale counter figure is visible in the line."""
    path = pmc_summary_path(workload)
    if path is None:
        return None
    try:
        collected = json.load(open(path)).get("collected_at_commit")
    except (OSError, ValueError):
        collected = None
    if not collected:  # summaries of earlier rounds do not say: the commit that put the file there
        try:
            r = subprocess.run(["git", "log", "-1", "--format=%h", "--", path], cwd=ROOT, capture_output=True, text=True, timeout=10)
            collected = r.stdout.strip() or None if r.returncode == 0 else None
        except (OSError, subprocess.SubprocessError):
            collected = None
    head = git_head()
    behind = None
    if collected and head:
        try:
            r = subprocess.run(["git", "rev-list", "--count", f"{collected}..{head}"], cwd=ROOT, capture_output=True, text=True, timeout=10)
            behind = int(r.stdout.strip()) if r.returncode == 0 and r.stdout.strip() else None
        except (OSError, subprocess.SubprocessError, ValueError):
            behind = None
    return {"summary": f"profiles/{os.path.basename(path)}", "collected_at_commit": collected, "head": head, "commits_behind_head": behind}


def limiter_of(workload, waves_per_simd=None):
    """{"limiter": what bounds the kernel on this workload, "limiter_source": the committed counter passes the statement rests on}.
    The statement is about the DEFAULT build and schedule as profiled in `limiter_source` (not re-derived by this run), with the
    occupancy filled in from the library (bm_trace_waves_per_simd of the instantiation that ran); it is omitted when no PMC summary
    of the workload is committed (config 1, config 4: config 3's kernel on another frame size)."""
    path = pmc_summary_path(workload)
    if workload not in LIMITER or path is None:
        return {"limiter": None, "limiter_source": None}
    return {"limiter": LIMITER[workload].format(waves=waves_per_simd if waves_per_simd else "?"),
            "limiter_source": f"profiles/{os.path.basename(path)} + DESIGN.md 4 (default build, default schedule)"}


def pmc_traffic(workload):
    """{"traffic_per_step": fabric-side bytes per step (frame), "valu_lanes": lanes active per VALU instruction, "valu_insts": wave-level VALU
    instructions per step, "traffic_source": where they come from}.  The counters are NOT collected by this run (rocprofv3 PMC
    needs its own passes, tools/profile_workload.py): the figures are read from the committed summary of the same workload and
    kernel -- per launch there, divided by the summary's frames per launch -- and the source is named in the line.  FETCH_SIZE and
    WRITE_SIZE are KiB.  On gfx950 FETCH_SIZE = read requests x 64 B: the guide's x2 applies to full-line coalesced streams only;
    every read of this kernel is a single 64-byte sector request, for which FETCH_SIZE is exact (profiles/r03_fetch_calibration.txt,
    measured with tools/ubench/fetch_calib.hip on 1-byte / 4-byte / 64-byte gathers) -- factor 1.0.  Infinity-Cache hits are included."""
    none = {"traffic_per_step": None, "valu_lanes": None, "valu_insts": None, "traffic_source": None}
    path = pmc_summary_path(workload)
    if path is None:
        return none
    try:
        with open(path) as f:
            d = json.load(f)
        if d.get("workload") != workload:
            return none
        frames = float(d.get("frames_per_launch", 1) or 1)
        lanes = d["SQ_THREAD_CYCLES_VALU"] / d["SQ_ACTIVE_INST_VALU"] if d.get("SQ_ACTIVE_INST_VALU") else None
        return {"traffic_per_step": int((1.0 * d["FETCH_SIZE_KiB"] + d["WRITE_SIZE_KiB"]) * 1024 / frames),
                "valu_lanes": round(lanes, 2) if lanes else None,
                "valu_insts": int(d["SQ_INSTS_VALU"] / frames) if d.get("SQ_INSTS_VALU") else None,
                "traffic_source": f"profiles/{os.path.basename(path)} (separate rocprofv3 --pmc passes of this workload, not this run)"}
    except (OSError, KeyError, ValueError):
        return none


if __name__ == "__main__":
    main()
