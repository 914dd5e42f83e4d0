"""Parity of the HIP path (through the C-ABI, brickmap_amd -> libbrickmap_hip.so) with the CPU oracle.

Bar: hit records (distance bits, normal, brick id, voxel id, per-path hashes over every extend and
shadow ray) BIT-EXACT; traversal counters exact; radiance within 1e-4 relative (the sky model calls
expf/powf/acosf of the platform, the only non-bit-exact ingredient).
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

RTOL = 1e-4  # north-star tolerance on per-pixel radiance


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def assert_radiance(got, want):
    scale = np.maximum(np.abs(want), 1e-6)
    err = np.abs(got - want) / scale
    assert float(err.max()) <= RTOL, f"max relative radiance error {err.max():.3e}"


def gpu_render(bm, torch, scene, cam, params, accum=None, want_dbg=True, also_plain=True):
    """Render through the C-ABI; returns the frame with ORDERED sums (BM_FLAG_ORDERED: reproducible bit for bit) and its hit records.
    With want_dbg the instrumented instantiation trace_paths<true> runs (hit records for the oracle comparison); the SAME frame is
    then rendered again
      (a) with the production instantiation trace_paths<false> (no hit records, no counters, twice the occupancy), ordered: a
          bit-identical accumulator;
      (b) as production frames run by default -- shadow rays on helper lanes, added with float atomics (trace_paths<false, *, true>,
          the kernel bench.py times): identical terminated-path counts, radiance equal up to summation order (2e-5);
      (c) where the caller checks traversal counters (BM_FLAG_COUNTERS on a resident scene with clean counters): with helper lanes AND
          counters, instrumented, no hit records -- the counters must equal the ordered frame's, i.e. the helpers walked the same rays;
      (d) as a production frame WITH hit records of its own (BM_FLAG_RAY_DIGEST: trace_paths<true, *, true>, helper lanes, (chunk,
          sample) items where the library would choose them): same accumulator as (b) up to summation order, and an order-independent
          per-pixel digest of every ray's hit -- kept in gpu_render.last_digest for the caller to compare with the oracle's
          (assert_ray_digest): the instantiation family bench.py times, compared with the oracle on HITS, bit for bit.
    So every case that checks the instrumented kernel against the oracle also pins the benchmarked one."""
    import copy
    rows = bm.local_rows(params)
    if accum is None:
        accum = torch.zeros((rows, params.width, 4), dtype=torch.float32, device="cuda:0")
    before = accum.clone() if also_plain else None
    dbg = torch.zeros((rows, params.width, 8), dtype=torch.int32, device="cuda:0") if want_dbg else None
    helper_counts = helper_acc = None
    if want_dbg and (params.flags & bm.BM_FLAG_COUNTERS) and not (params.flags & bm.BM_FLAG_ORDERED):
        # the caller compares the scene's counters with the oracle's after this call: render the frame once with helper lanes AND
        # counters first (instrumented instantiation, no hit records), keep its counts, and hand the caller a clean slate.  Only when
        # the counters are clean and the scene is fully resident (an extra frame of a streaming scene changes what is requested).
        info = scene.info()
        if not any(scene.counters().values()) and info["resident_bricks"] == info["total_bricks"]:
            scratch = accum.clone()
            scene.render(cam, params, scratch)
            torch.cuda.synchronize()
            helper_counts, helper_acc = scene.counters(), scratch.cpu().numpy()
            scene.counters_reset()
    ordered = copy.copy(params)
    ordered.flags = params.flags | bm.BM_FLAG_ORDERED  # the frame the caller gets: reproducible sums, whichever instantiation renders it
    scene.render(cam, ordered, accum, debug=dbg)
    torch.cuda.synchronize()
    a = accum.cpu().numpy()
    if also_plain:
        # (a) ordered sums (BM_FLAG_ORDERED: one lane accumulates a pixel's events in path order): the same bits as the instrumented kernel
        if want_dbg:
            plain = copy.copy(params)
            plain.flags = (params.flags & ~bm.BM_FLAG_COUNTERS) | bm.BM_FLAG_ORDERED
            acc2 = before.clone()
            scene.render(cam, plain, acc2)
            torch.cuda.synchronize()
            b = acc2.cpu().numpy()
            if params.flags & bm.BM_FLAG_SAMPLE_ITEMS:
                np.testing.assert_allclose(b, a, rtol=2e-5, atol=1e-7, err_msg="trace_paths<false> differs from trace_paths<true>")
            else:
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "trace_paths<false> differs from trace_paths<true>"
        # (b) the DEFAULT of production frames -- shadow rays on helper lanes, added with float atomics (trace.hip HELP) -- traces the same
        # rays: terminated-path counts identical, radiance equal up to summation order
        if not (params.flags & bm.BM_FLAG_ORDERED):
            hp = copy.copy(params)
            hp.flags = params.flags & ~(bm.BM_FLAG_COUNTERS | bm.BM_FLAG_ORDERED)
            acc3 = before.clone()
            scene.render(cam, hp, acc3)
            torch.cuda.synchronize()
            h = acc3.cpu().numpy()
            assert np.array_equal(h[..., 3], a[..., 3]), "helper lanes: terminated-path counts differ"
            np.testing.assert_allclose(h[..., :3], a[..., :3], rtol=2e-5, atol=1e-7, err_msg="helper lanes: radiance differs")
    if helper_counts is not None:
        # (c) the instrumented kernel WITH helper lanes (counters on, no hit records) walked exactly what this frame walked
        assert scene.counters() == helper_counts, "helper lanes: traversal counters differ from the ordered frame's"
        assert np.array_equal(helper_acc[..., 3], a[..., 3])
        np.testing.assert_allclose(helper_acc[..., :3], a[..., :3], rtol=2e-5, atol=1e-7, err_msg="helper lanes (instrumented): radiance differs")
    gpu_render.last_digest = None
    if want_dbg and not (params.flags & (bm.BM_FLAG_ORDERED | bm.BM_FLAG_PRIMARY_ONLY)):
        info = scene.info()
        if info["resident_bricks"] == info["total_bricks"]:  # (an extra frame of a streaming scene changes what is requested)
            dp = copy.copy(params)
            dp.flags = (params.flags & ~bm.BM_FLAG_COUNTERS) | bm.BM_FLAG_RAY_DIGEST
            acc_d = before.clone() if before is not None else torch.zeros_like(accum)
            dig = torch.zeros((rows, params.width, 8), dtype=torch.int32, device="cuda:0")
            scene.render(cam, dp, acc_d, debug=dig)
            torch.cuda.synchronize()
            if before is not None:
                g = acc_d.cpu().numpy()
                assert np.array_equal(g[..., 3], a[..., 3]), "ray-digest frame: terminated-path counts differ"
                np.testing.assert_allclose(g[..., :3], a[..., :3], rtol=2e-5, atol=1e-7, err_msg="ray-digest frame: radiance differs")
            gpu_render.last_digest = dig.cpu().numpy().view(np.uint32)
            # ray counts and cells visited are sums either way: they must equal the ordered frame's records
            d_ord = dbg.cpu().numpy().view(np.uint32)
            assert np.array_equal(gpu_render.last_digest[..., 6:], d_ord[..., 6:]) and np.array_equal(gpu_render.last_digest[..., :4], d_ord[..., :4])
    return a, (dbg.cpu().numpy().view(np.uint32) if want_dbg else None)


gpu_render.last_digest = None


def assert_ray_digest(world, rows=None):
    """The production-plan frame of the last gpu_render call (helper lanes, float atomics: the timed instantiation family) traced the
    oracle's rays: per pixel the keyed sums over all its extend / shadow rays of (hit, distance bits, normal, level, brick id, voxel
    id) / (occluded, occluder), the ray counts and the cells visited equal those of the oracle's last render, bit for bit."""
    got, want = gpu_render.last_digest, world.last_ray_digest
    assert got is not None and want is not None
    if rows is not None:
        got, want = got[rows], want[rows]
    assert np.array_equal(got, want), f"{np.count_nonzero((got != want).any(-1))} pixels whose ray digest differs from the oracle's"


def cameras(bm, orc, grid, pos=None, h=0.8, v=-0.5, direction=None):
    pos = pos or (grid / 2, grid / 8, 0.8 * grid)
    cam = bm.Camera(position=pos, horizontal_angle=h, vertical_angle=v).update()
    if direction is not None:
        cam.direction = tuple(float(x) for x in direction)
    return cam, orc.make_camera(cam.position, cam.direction)


@pytest.fixture(scope="module")
def scene256(bm, torch_cuda):
    s = bm.Scene(256, 256, device=0).generate()
    s.preload_all()
    yield s
    s.close()


def test_library_loaded_is_the_hip_extension(bm, torch_cuda):
    from brickmap_amd import _lib
    maps = open("/proc/self/maps").read()
    assert _lib.LIB_PATH in maps, "the in-tree HIP extension is not the code that is running"


def test_sincos_bit_exact(bm, orc, torch_cuda):
    from brickmap_amd import _lib
    x = np.concatenate([np.linspace(-8, 8, 200001), np.random.default_rng(1).random(100000) * 2 * np.pi]).astype(np.float32)
    s, c = np.zeros_like(x), np.zeros_like(x)
    _lib.check(_lib.load().bm_debug_sincos(0, x.size, x.ctypes.data, s.ctypes.data, c.ctypes.data))
    ws, wc = orc.sincos(x)
    assert np.array_equal(s.view(np.uint32), ws.view(np.uint32)) and np.array_equal(c.view(np.uint32), wc.view(np.uint32))


def test_sky_model_matches(bm, orc, torch_cuda):
    from brickmap_amd import _lib
    rng = np.random.default_rng(2)
    v = rng.normal(size=(4096, 3)).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True).astype(np.float32)
    for sun in [(0.05, 0.1), (0.3, 0.4), (0.7, 0.05)]:
        out = [np.zeros_like(v) for _ in range(3)]
        sp = np.float32(sun)
        _lib.check(_lib.load().bm_debug_sky(0, sp.ctypes.data, len(v), v.ctypes.data, *[o.ctypes.data for o in out]))
        want = [np.stack([orc.sky_probe(d, sun=sun)[k] for d in v]) for k in ("sun", "sky", "sunsky")]
        for g, w in zip(out, want):
            finite = np.isfinite(w)
            assert np.array_equal(np.isfinite(g), finite)
            np.testing.assert_allclose(g[finite], w[finite], rtol=2e-5, atol=1e-12)


def test_world_on_device_matches_oracle(bm, orc, torch_cuda, scene256, world256):
    info = scene256.info()
    assert info["total_bricks"] == world256.total_bricks() and info["resident_bricks"] == info["total_bricks"]
    assert info["index_bytes"] == world256.nsc * 16384 and info["pool_bytes"] == 64 * info["total_bricks"] <= info["brick_bytes"]
    assert info["cube_field_bytes"] == 8 * 34 * 34 * 64  # 8 planes x (cells_h + 2) slices x (cells + 2) rows, rows padded to a power of two
    for sc in range(world256.nsc):
        idx, bricks = scene256.host_supercell(sc)  # host side: the reference's words and brick order
        want_idx, want_bricks = world256.sc_indices(sc), world256.sc_bricks(sc)
        assert np.array_equal(idx, want_idx) and np.array_equal(bricks, want_bricks)
        # device side, "all bricks pre-loaded": the reference's host words (slot | loaded | lod, Scene.cpp:104) and the
        # supercell's bricks in host order -- pools are the full host brick vectors
        dev = scene256.device_indices(sc)
        assert np.array_equal(dev, want_idx)
        nz = np.flatnonzero(dev)
        for local in nz[:: max(1, len(nz) // 40)]:
            assert np.array_equal(scene256.device_brick(sc, int(dev[local] & 0xFFF)), want_bricks[want_idx[local] & 0xFFF])


def test_config1_primary_rays_golden(bm, orc, torch_cuda):
    """BASELINE config 1 against the committed fixture: 256x256 primary-ray DDA into one superchunk."""
    g = np.load(os.path.join(GOLDEN, "config1_primary.npz"))
    scene = bm.Scene(int(g["grid"]), int(g["grid"]), device=0).generate().preload_all()
    cam = bm.Camera(position=tuple(float(x) for x in g["cam_pos"]), direction=tuple(float(x) for x in g["cam_dir"]))
    p = bm.FrameParams(int(g["width"]), int(g["height"]), spp=1, max_bounces=0, flags=bm.BM_FLAG_PRIMARY_ONLY | bm.BM_FLAG_COUNTERS)
    scene.counters_reset()
    acc, dbg = gpu_render(bm, torch_cuda, scene, cam, p)
    assert np.array_equal(dbg[..., :4], g["hits"])
    assert_radiance(acc, g["accum"])
    cnt = scene.counters()
    assert [cnt[k] for k in orc.COUNTER_NAMES] == list(g["counters"])
    scene.close()


def test_path4_golden_and_counters(bm, orc, torch_cuda, scene256):
    g = np.load(os.path.join(GOLDEN, "path4_small.npz"))
    cam = bm.Camera(position=tuple(float(x) for x in g["cam_pos"]), direction=tuple(float(x) for x in g["cam_dir"]))
    p = bm.FrameParams(int(g["width"]), int(g["height"]), spp=int(g["spp"]), max_bounces=int(g["max_bounces"]), flags=bm.BM_FLAG_COUNTERS)
    scene256.counters_reset()
    acc, dbg = gpu_render(bm, torch_cuda, scene256, cam, p)
    assert np.array_equal(dbg, g["dbg"])
    assert_radiance(acc, g["accum"])
    cnt = scene256.counters()
    assert [cnt[k] for k in orc.COUNTER_NAMES] == list(g["counters"])


CASES = [
    # (name, width, height, spp, max_bounces, camera kwargs)
    ("inside_down", 128, 96, 1, 3, dict()),
    ("ragged_multi_spp", 75, 41, 3, 3, dict()),
    ("eight_segments", 64, 48, 2, 7, dict()),
    ("outside_world", 96, 64, 1, 3, dict(pos=(-300.0, -200.0, 500.0), h=0.9, v=-0.4)),
    ("above_looking_down", 64, 64, 1, 3, dict(pos=(128.0, 128.0, 700.0), h=0.3, v=-1.5)),
    ("all_sky", 48, 32, 1, 3, dict(v=1.2)),
    ("axis_aligned", 64, 48, 1, 3, dict(pos=(10.5, 100.5, 140.5), direction=(1.0, 0.0, 0.0))),
    ("grazing_boundary", 64, 48, 1, 3, dict(pos=(0.0, 0.0, 255.999), h=0.785, v=-0.3)),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_paths_match_oracle(case, bm, orc, torch_cuda, scene256, world256):
    _, W, H, spp, mb, ck = case
    cam, ocam = cameras(bm, orc, 256, **ck)
    world256.reset_device(True)
    p = bm.FrameParams(W, H, spp=spp, max_bounces=mb, flags=bm.BM_FLAG_COUNTERS)
    scene256.counters_reset()
    acc, dbg = gpu_render(bm, torch_cuda, scene256, cam, p)
    oacc, odbg, ocnt, _ = world256.render(ocam, orc.make_frame(W, H, spp=spp, max_bounces=mb))
    assert np.array_equal(dbg, odbg), f"{np.count_nonzero((dbg != odbg).any(-1))} pixels with different hit records"
    assert_ray_digest(world256)  # ... and the helper-lane frame's hits, ray by ray
    assert_radiance(acc, oacc)
    assert scene256.counters() == ocnt
    assert np.all(acc[..., 3] == spp)  # every path terminates exactly once


def test_lens_and_focal_distance(bm, orc, torch_cuda, scene256, world256):
    cam, _ = cameras(bm, orc, 256)
    cam.lensRadius, cam.focalDistance = 0.75, 2.5
    ocam = orc.make_camera(cam.position, cam.direction, focal_distance=2.5, lens_radius=0.75)
    p = bm.FrameParams(64, 48, spp=2, max_bounces=3)
    acc, dbg = gpu_render(bm, torch_cuda, scene256, cam, p)
    oacc, odbg, _, _ = world256.render(ocam, orc.make_frame(64, 48, spp=2, max_bounces=3))
    assert np.array_equal(dbg, odbg)
    assert_ray_digest(world256)
    assert_radiance(acc, oacc)


def test_sun_position_change(bm, orc, torch_cuda, scene256, world256):
    cam, ocam = cameras(bm, orc, 256)
    sun = (0.31, 0.22)
    p = bm.FrameParams(64, 48, spp=1, max_bounces=3, sun_position=sun)
    acc, dbg = gpu_render(bm, torch_cuda, scene256, cam, p)
    oacc, odbg, _, _ = world256.render(ocam, orc.make_frame(64, 48, spp=1, max_bounces=3, sun=sun))
    assert np.array_equal(dbg, odbg)
    assert_ray_digest(world256)
    assert_radiance(acc, oacc)


def test_lod_levels(bm, orc, torch_cuda):
    """LoD thresholds shrunk so that all three levels (8^3 voxels, 2^3 byte, solid brick) occur in one frame."""
    scene = bm.Scene(256, 256, device=0)
    scene.set_lod(400, 60)
    scene.generate().preload_all()
    w = orc.World(256, 256)
    w.set_lod(400, 60)
    w.reset_device(True)
    cam, ocam = cameras(bm, orc, 256)
    p = bm.FrameParams(96, 64, spp=1, max_bounces=3, flags=bm.BM_FLAG_COUNTERS)
    scene.counters_reset()
    acc, dbg = gpu_render(bm, torch_cuda, scene, cam, p)
    oacc, odbg, ocnt, _ = w.render(ocam, orc.make_frame(96, 64, spp=1, max_bounces=3))
    levels = set(((odbg[..., 1] >> 12) & 0xF)[odbg[..., 1] != 0].tolist())
    assert {0, 1, 2} <= levels, f"test frame does not exercise all LoD levels: {levels}"
    assert np.array_equal(dbg, odbg)
    assert_ray_digest(w)
    assert_radiance(acc, oacc)
    assert scene.counters() == ocnt and ocnt["byte_tests"] > 0
    scene.close()


def test_non_cubic_world(bm, orc, torch_cuda):
    scene = bm.Scene(384, 128, device=0).generate().preload_all()
    w = orc.World(384, 128)
    w.reset_device(True)
    cam = bm.Camera(position=(200.0, 40.0, 110.0), horizontal_angle=0.6, vertical_angle=-0.35).update()
    ocam = orc.make_camera(cam.position, cam.direction)
    acc, dbg = gpu_render(bm, torch_cuda, scene, cam, bm.FrameParams(80, 60, spp=1, max_bounces=3))
    oacc, odbg, _, _ = w.render(ocam, orc.make_frame(80, 60, spp=1, max_bounces=3))
    assert np.array_equal(dbg, odbg)
    assert_radiance(acc, oacc)
    scene.close()


@pytest.mark.parametrize("dims,pos,angles", [
    ((128, 7936), (40.0, 40.0, 5900.0), (0.4, -1.5)),      # a very tall world (992 bricks): down the shaft
    ((128, 7936), (-900.0, 64.0, 3500.0), (1.5, 0.05)),    # camera outside, looking across the column below the terrain top
    ((8192, 128), (8000.0, 8100.0, 120.0), (3.9, -0.12)),  # widest world (1024 bricks): the far corner, looking back
])
def test_extreme_world_dimensions(dims, pos, angles, bm, orc, torch_cuda):
    """The packed brick cell (11 + 11 + 10 bits, biased) at the ends of its range, the bordered block grid at the far faces."""
    G, GH = dims
    scene = bm.Scene(G, GH, device=0).generate().preload_all()
    w = orc.World(G, GH, threads=os.cpu_count() or 1)
    w.reset_device(True)
    cam = bm.Camera(position=pos, horizontal_angle=angles[0], vertical_angle=angles[1]).update()
    ocam = orc.make_camera(cam.position, cam.direction)
    acc, dbg = gpu_render(bm, torch_cuda, scene, cam, bm.FrameParams(96, 64, spp=1, max_bounces=3))
    oacc, odbg, _, _ = w.render(ocam, orc.make_frame(96, 64, spp=1, max_bounces=3), threads=os.cpu_count() or 1)
    assert np.array_equal(dbg, odbg)
    assert (dbg[..., 1] != 0).mean() > 0.02  # the view does hit the world
    assert_radiance(acc, oacc)
    scene.close()


def test_streaming_first_frame_and_steady_state(bm, orc, torch_cuda):
    """Reference initial residency: nothing loaded.  Frame 1 treats every unloaded brick as solid and
    requests it (deterministic image, deterministic request SET); at steady state the image equals
    the all-resident image."""
    G, W, H = 256, 96, 64
    scene = bm.Scene(G, G, device=0)
    scene.set_queue_capacity(1 << 16)
    scene.generate()
    w = orc.World(G, G)
    w.set_queue_cap(1 << 16)
    w.reset_device(False)
    cam, ocam = cameras(bm, orc, G)
    p = bm.FrameParams(W, H, spp=1, max_bounces=3, flags=bm.BM_FLAG_COUNTERS)
    scene.counters_reset()
    acc, dbg = gpu_render(bm, torch_cuda, scene, cam, p)
    oacc, odbg, ocnt, _ = w.render(ocam, orc.make_frame(W, H, spp=1, max_bounces=3))
    assert np.array_equal(dbg, odbg)
    assert_radiance(acc, oacc)
    assert scene.counters() == ocnt and ocnt["requests"] > 0 and ocnt["brick_tests"] == 0
    # requested-bit sets agree supercell by supercell
    for sc in range(w.nsc):
        assert np.array_equal(scene.device_indices(sc), w.sc_dev_indices(sc))
    n_gpu, n_cpu = scene.process_load_queue(), w.process_load_queue()
    w.upload()
    assert n_gpu == n_cpu == ocnt["requests"]
    assert scene.info()["resident_bricks"] == n_gpu
    for _ in range(64):
        gpu_render(bm, torch_cuda, scene, cam, p, want_dbg=False)
        if scene.process_load_queue() == 0:
            break
    else:
        pytest.fail("streaming did not reach a steady state")
    acc_s, dbg_s = gpu_render(bm, torch_cuda, scene, cam, bm.FrameParams(W, H, spp=1, max_bounces=3))
    scene.preload_all()
    acc_r, dbg_r = gpu_render(bm, torch_cuda, scene, cam, bm.FrameParams(W, H, spp=1, max_bounces=3))
    assert np.array_equal(dbg_s, dbg_r) and np.array_equal(acc_s, acc_r)
    scene.dump("/tmp/bm_dump.txt")
    assert len(open("/tmp/bm_dump.txt").read().split()) == w.nsc
    scene.close()


def test_helper_lanes_request_the_same_bricks(bm, orc, torch_cuda):
    """Streaming with production frames: shadow rays traced by helper lanes raise brick requests like any other ray (voxel.cuh:228-245).
    From empty residency, production frames (helper lanes, 2 spp as (chunk, sample) items) and ordered frames request the same SET of
    bricks frame by frame -- the counts serviced per frame are equal, the resident sets end up identical, and the first frame's requests
    are the oracle's -- although the ORDER in which a frame's lanes reach the ring differs."""
    G, W, H = 256, 112, 80
    torch = torch_cuda
    cam, ocam = cameras(bm, orc, G)
    w = orc.World(G, G)
    w.set_queue_cap(1 << 16)
    w.reset_device(False)
    _, _, ocnt, _ = w.render(ocam, orc.make_frame(W, H, spp=2, max_bounces=3))
    scenes, counts = [], []
    for flags in (0, bm.BM_FLAG_ORDERED):
        sc = bm.Scene(G, G, device=0)
        sc.set_queue_capacity(1 << 16)
        sc.generate()
        acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
        per_frame = []
        for _ in range(64):
            sc.render(cam, bm.FrameParams(W, H, spp=2, max_bounces=3, flags=flags), acc)
            n = sc.process_load_queue()
            per_frame.append(n)
            if n == 0:
                break
        else:
            pytest.fail("no streaming steady state")
        scenes.append(sc)
        counts.append(per_frame)
    assert counts[0] == counts[1] and counts[0][0] == ocnt["requests"] > 0
    a, b = scenes
    assert a.info()["resident_bricks"] == b.info()["resident_bricks"] == sum(counts[0])
    for sci in range(w.nsc):
        ia, ib = a.device_indices(sci), b.device_indices(sci)
        # the same bricks are resident (loaded bit), whatever slot the request order gave them
        assert np.array_equal(ia >> 31, ib >> 31) and np.array_equal(ia == 0, ib == 0)
    # ... and the steady-state frames agree: ordered bit for bit with each other, the production frame to summation order
    pa, _ = gpu_render(bm, torch, a, cam, bm.FrameParams(W, H, spp=2, max_bounces=3), want_dbg=False)
    pb, _ = gpu_render(bm, torch, b, cam, bm.FrameParams(W, H, spp=2, max_bounces=3), want_dbg=False)
    assert np.array_equal(pa.view(np.uint32), pb.view(np.uint32))
    a.close(); b.close()


def test_request_ring_overflow(bm, orc, torch_cuda):
    """A frame that wants more bricks than the ring holds gets exactly `capacity` serviced; the losers'
    request bits are cleared again (voxel.cuh:234-240) so that they can ask again next frame."""
    scene = bm.Scene(256, 256, device=0)
    assert scene.info()["queue_capacity"] == 1024  # reference ring size, variables.h:35
    scene.set_queue_capacity(64)
    scene.generate()
    cam, _ = cameras(bm, orc, 256)
    p = bm.FrameParams(256, 192, spp=1, max_bounces=3)
    gpu_render(bm, torch_cuda, scene, cam, p, want_dbg=False)
    requested = sum(int(np.count_nonzero(scene.device_indices(sc) & bm.BRICK_REQUESTED_BIT)) for sc in range(8))
    assert requested == 64
    assert scene.process_load_queue() == 64
    requested = sum(int(np.count_nonzero(scene.device_indices(sc) & bm.BRICK_REQUESTED_BIT)) for sc in range(8))
    loaded = sum(int(np.count_nonzero(scene.device_indices(sc) & bm.BRICK_LOADED_BIT)) for sc in range(8))
    assert loaded == 64 and requested == 0
    total = 64
    for _ in range(200):
        gpu_render(bm, torch_cuda, scene, cam, p, want_dbg=False)
        n = scene.process_load_queue()
        total += n
        if n == 0:
            break
    assert n == 0 and total == scene.info()["resident_bricks"] > 64
    scene.close()


def test_request_ring_overflow_in_overlapped_mode(bm, orc, torch_cuda):
    """The overlapped mode's snapshot kernel copies min(count, capacity) entries of a ring whose counter ran past its capacity
    (the device counter keeps counting, kernel.cu:409 / Scene.cpp:203 clamp it): every call services at most `capacity` bricks,
    none twice, and the scene still converges to the resident image."""
    G, W, H = 256, 96, 64
    scene = bm.Scene(G, G, device=0)
    scene.set_queue_capacity(64)
    scene.generate()
    scene.set_streaming_mode(True)
    cam, _ = cameras(bm, orc, G)
    p = bm.FrameParams(W, H, spp=1, max_bounces=3)
    total, batches, idle = 0, [], 0
    for _ in range(400):
        gpu_render(bm, torch_cuda, scene, cam, p, want_dbg=False)
        n = scene.process_load_queue()
        assert 0 <= n <= 64
        total += n
        batches.append(n)
        idle = idle + 1 if n == 0 else 0
        if idle >= 3:  # (a request is serviced two calls after the frame that raised it)
            break
    else:
        pytest.fail("overlapped streaming with a 64-entry ring did not reach a steady state")
    assert max(batches) == 64 and total == scene.info()["resident_bricks"] > 256
    acc_s, dbg_s = gpu_render(bm, torch_cuda, scene, cam, p)
    ref = bm.Scene(G, G, device=0).generate().preload_all()
    acc_r, dbg_r = gpu_render(bm, torch_cuda, ref, cam, p)
    assert np.array_equal(dbg_s, dbg_r) and np.array_equal(acc_s, acc_r)
    ref.close()
    scene.close()


def test_sharded_equals_unsharded_and_accumulation(bm, orc, torch_cuda, scene256):
    torch = torch_cuda
    cam, _ = cameras(bm, orc, 256)
    W, H = 100, 70
    full, dfull = gpu_render(bm, torch, scene256, cam, bm.FrameParams(W, H, spp=2, max_bounces=3))
    out = np.zeros_like(full)
    for r in range(3):
        p = bm.FrameParams(W, H, spp=2, max_bounces=3, band_rows=16, shard_rank=r, shard_count=3)
        acc, _ = gpu_render(bm, torch, scene256, cam, p)
        out[bm.dist.shard_rows(H, 16, r, 3)] = acc
    assert np.array_equal(out, full)
    # two launches of 1 spp == one launch of 2 spp (same per-pixel accumulation order)
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    for s in range(2):
        scene256.render(cam, bm.FrameParams(W, H, spp=1, sample_base=s, max_bounces=3, flags=bm.BM_FLAG_ORDERED), acc)
    torch.cuda.synchronize()
    assert np.array_equal(acc.cpu().numpy(), full)


@pytest.mark.parametrize("W,H,spp,mb", [(1, 1, 1, 3), (3, 5, 3, 0), (17, 1, 2, 7), (130, 33, 1, 2)])
def test_edge_frame_shapes(W, H, spp, mb, bm, orc, torch_cuda, scene256, world256):
    """Frames smaller than a chunk / a wave, ragged against the 4x4 chunks and 16x16 tiles, no bounces, long paths; spp = 0."""
    cam, ocam = cameras(bm, orc, 256)
    acc, dbg = gpu_render(bm, torch_cuda, scene256, cam, bm.FrameParams(W, H, spp=spp, max_bounces=mb))
    oacc, odbg, _, _ = world256.render(ocam, orc.make_frame(W, H, spp=spp, max_bounces=mb))
    assert np.array_equal(dbg, odbg)
    assert_radiance(acc, oacc)
    none, _ = gpu_render(bm, torch_cuda, scene256, cam, bm.FrameParams(W, H, spp=0, max_bounces=mb), want_dbg=False)
    assert not none.any()  # no samples: the accumulation buffer is untouched


def test_sample_sharded_frames_sum_to_the_single_render(bm, orc, torch_cuda, scene256):
    """The throughput decomposition bench.py uses at N > 1: rank r renders the FULL frame with samples
    [r*spp, (r+1)*spp); the sum of the N frames is the N*spp render up to floating-point association."""
    torch = torch_cuda
    cam, _ = cameras(bm, orc, 256)
    W, H, spp, N = 96, 64, 2, 4
    whole, dbg = gpu_render(bm, torch, scene256, cam, bm.FrameParams(W, H, spp=spp * N, max_bounces=3))
    total = np.zeros_like(whole)
    for r in range(N):
        part, _ = gpu_render(bm, torch, scene256, cam, bm.FrameParams(W, H, spp=spp, sample_base=r * spp, max_bounces=3), want_dbg=False)
        total += part
    assert np.array_equal(total[..., 3], whole[..., 3]) and whole[..., 3].min() == spp * N
    assert_radiance(total, whole)


def test_launch_kernels_mirror_and_resolve(bm, orc, torch_cuda, scene256, world256):
    """The reference-shaped call sequence: State + launch_kernels per frame, then the resolve ("blit")."""
    torch = torch_cuda
    from brickmap_amd.host import _LaunchStatics
    st = _LaunchStatics()
    state = bm.State(64, 48, device=0)
    cam, ocam = cameras(bm, orc, 256)
    for _ in range(3):
        assert bm.launch_kernels(state, state.blit_buffer, scene256, cam, spp=1, statics=st) == 0
    torch.cuda.synchronize()
    oacc, _, _, _ = world256.render(ocam, orc.make_frame(64, 48, spp=3, max_bounces=3), want_dbg=False)
    assert_radiance(state.blit_buffer.cpu().numpy(), oacc)
    out = scene256.resolve(state.blit_buffer)
    torch.cuda.synchronize()
    want = np.zeros_like(oacc)
    orc.lib().orc_resolve(oacc.ctypes.data, want.ctypes.data, 64 * 48)
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=2e-4)
    # moving the camera resets the accumulation (kernel.cu:387-403)
    cam2 = bm.Camera(position=(130.0, 40.0, 200.0), horizontal_angle=0.7, vertical_angle=-0.5).update()
    bm.launch_kernels(state, state.blit_buffer, scene256, cam2, spp=1, statics=st)
    torch.cuda.synchronize()
    assert float(state.blit_buffer[..., 3].max()) == 1.0


def test_full_size_config2_properties(bm, orc, torch_cuda):
    """BASELINE config 2 at full size (1920x1080, 4 segments, 8^3 superchunks): size-independent
    properties plus an exact comparison with the oracle on every 60th row."""
    torch = torch_cuda
    G, W, H = 1024, 1920, 1080
    scene = bm.Scene(G, G, device=0).generate().preload_all()
    cam, ocam = cameras(bm, orc, G)
    p = bm.FrameParams(W, H, spp=1, max_bounces=3, flags=bm.BM_FLAG_COUNTERS)
    scene.counters_reset()
    acc, dbg = gpu_render(bm, torch, scene, cam, p)
    digest_full = gpu_render.last_digest
    cnt = scene.counters()
    assert np.all(acc[..., 3] == 1.0) and np.all(np.isfinite(acc))
    assert cnt["paths"] == W * H and cnt["paths"] <= cnt["extend_rays"] <= 4 * cnt["paths"] and cnt["shadow_rays"] <= cnt["extend_rays"]
    assert int((dbg[..., 6] & 0xFFFF).sum()) == cnt["extend_rays"] and int((dbg[..., 6] >> 16).sum()) == cnt["shadow_rays"]
    assert int(dbg[..., 7].astype(np.uint64).sum()) == cnt["index_loads"]
    # sharded render is bit-identical to the unsharded one at full size
    out = np.zeros_like(acc)
    for r in range(2):
        a, _ = gpu_render(bm, torch, scene, cam, bm.FrameParams(W, H, spp=1, max_bounces=3, band_rows=16, shard_rank=r, shard_count=2), want_dbg=False)
        out[bm.dist.shard_rows(H, 16, r, 2)] = a
    assert np.array_equal(out, acc)
    # oracle on a 1/60 sample of the rows
    w = orc.World(G, G)
    w.reset_device(True)
    oacc, odbg, _, _ = w.render(ocam, orc.make_frame(W, H, spp=1, max_bounces=3, band_rows=1, shard_rank=7, shard_count=60),
                                threads=os.cpu_count() or 1)
    rows = bm.dist.shard_rows(H, 1, 7, 60)
    assert np.array_equal(dbg[rows], odbg[rows])
    assert_radiance(acc[rows], oacc[rows])
    # the frame bench.py times -- trace_paths<*, false, true> on the full-size frame -- against the oracle on hits: the digest of the
    # production-plan render (the first gpu_render call above) on the same 18 rows
    assert np.array_equal(digest_full[rows], w.last_ray_digest[rows]), "config 2 at full size: the helper-lane frame's ray digest differs from the oracle's"
    assert int((digest_full[..., 6] & 0xFFFF).sum()) == cnt["extend_rays"] and int(digest_full[..., 7].astype(np.uint64).sum()) == cnt["index_loads"]
    # ... and as bench.py ISSUES its steps: consecutive frames of one view as ONE uniform frame-ring launch (trace_paths<*, false, true, 2>,
    # lanes of several frames in one wave), here three frames with the ray digest of the whole launch in one buffer = the oracle's digest
    # of the 3-sample frame, on the same 18 rows
    acc3 = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    dig3 = torch.zeros((H, W, 8), dtype=torch.int32, device="cuda:0")
    scene.render_frames(cam, [bm.FrameParams(W, H, spp=1, sample_base=k, max_bounces=3, flags=bm.BM_FLAG_RAY_DIGEST) for k in range(3)], acc3, debugs=[dig3] * 3)
    torch.cuda.synchronize()
    oacc3, _, _, _ = w.render(ocam, orc.make_frame(W, H, spp=3, max_bounces=3, band_rows=1, shard_rank=7, shard_count=60), threads=os.cpu_count() or 1)
    assert np.array_equal(dig3.cpu().numpy().view(np.uint32)[rows], w.last_ray_digest[rows]), "config 2 at full size, frame ring: ray digest differs from the oracle's"
    a3 = acc3.cpu().numpy()
    assert np.all(a3[..., 3] == 3.0)
    assert_radiance(a3[rows], oacc3[rows])
    scene.close()


def test_errors_are_reported_not_fatal(bm, torch_cuda):
    import ctypes as C
    from brickmap_amd import _lib
    L = _lib.load()
    h = C.c_void_p()
    assert L.bm_scene_create(0, 100, 128, C.byref(h)) == 10001 and b"multiples of 128" in L.bm_last_error_string()
    # a ray's cell is one 32-bit offset into the padded cube field, candidates take 24-bit products of brick coordinates: larger worlds
    # are refused, not mis-traced
    assert L.bm_scene_create(0, 8192 + 128, 128, C.byref(h)) == 10001 and b"world too large" in L.bm_last_error_string()
    assert L.bm_scene_create(0, 1024, 8192 + 128, C.byref(h)) == 10001 and b"world too large" in L.bm_last_error_string()
    assert L.bm_scene_create(0, 8192, 8192, C.byref(h)) == 10001 and b"4 GiB" in L.bm_last_error_string()
    s = bm.Scene(128, 128, device=0)
    with pytest.raises(bm.BrickmapError):
        s.preload_all()  # not generated yet
    s.close()


def test_cpp_headless_example_streams_and_renders(bm, torch_cuda, tmp_path):
    """The C++ mirror (include/brickmap.hpp) driven like the reference's main loop: generate, then
    launch_kernels + process_load_queue per frame with on-demand brick streaming, resolve, write a PPM."""
    import subprocess
    from conftest import ROOT
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples")])
    out = tmp_path / "frame.ppm"
    r = subprocess.run([os.path.join(ROOT, "examples", "headless_main"), "256", "256", "160", "96", "24", str(out)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "bricks resident" in r.stdout
    data = out.read_bytes()
    assert data.startswith(b"P6\n160 96\n255\n") and len(data) == len(b"P6\n160 96\n255\n") + 160 * 96 * 3
    px = np.frombuffer(data[len(b"P6\n160 96\n255\n"):], np.uint8)
    assert px.max() > 0 and len(np.unique(px)) > 16  # an actual image, not a constant
    # launch_frames (the frame ring through the C++ mirror): 24 frames as ONE launch on the resident world give the image of 24
    # launch_kernels calls (here: of the streamed run once everything it sees is resident -- same paths, 8-bit pixels)
    ring = tmp_path / "ring.ppm"
    r = subprocess.run([os.path.join(ROOT, "examples", "headless_main"), "256", "256", "160", "96", "24", str(ring), "ring"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    rp = np.frombuffer(ring.read_bytes()[len(b"P6\n160 96\n255\n"):], np.uint8)
    assert rp.shape == px.shape and rp.max() > 0
    # the streamed run's first frames treated unloaded bricks as solid: most pixels agree closely, all are an image of the same scene
    assert np.mean(np.abs(rp.astype(np.int32) - px.astype(np.int32)) <= 8) > 0.9


def test_cpp_multi_gpu_example_with_one_rank(bm, torch_cuda, tmp_path):
    """The C++ multi-GPU loop of INTEGRATION.md (Shard + Comm + gather_frame of include/brickmap.hpp, RCCL behind the C-ABI)
    with one rank: communicator from a file-carried id, sharded State, bm_gather_frame every frame, barrier, resolve."""
    import subprocess
    from conftest import ROOT
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples")])
    out = tmp_path / "mg.ppm"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([os.path.join(ROOT, "examples", "multi_gpu_main"), "0", "1", str(tmp_path / "id"), "256", "256", "160", "96", "12", "2", str(out)],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "1 ranks, 12 frames of 2 spp" in r.stdout
    data = out.read_bytes()
    px = np.frombuffer(data[len(b"P6\n160 96\n255\n"):], np.uint8)
    assert len(px) == 160 * 96 * 3 and px.max() > 0 and len(np.unique(px)) > 16


def test_scheduler_statistics(bm, orc, torch_cuda, scene256):
    cam, _ = cameras(bm, orc, 256)
    scene256.counters_reset()
    gpu_render(bm, torch_cuda, scene256, cam, bm.FrameParams(128, 96, spp=2, max_bounces=3, flags=bm.BM_FLAG_COUNTERS), want_dbg=False)
    c, s = scene256.counters(), scene256.sched_stats()
    # a visited cell is a ray start, a single move, or one of the (at most 3 * 254) cells crossed by a jump
    assert s["step_lanes"] + 762 * s["jump_lanes"] + c["extend_rays"] + c["shadow_rays"] >= c["index_loads"]
    assert s["step_lanes"] + s["jump_lanes"] + c["extend_rays"] + c["shadow_rays"] < c["index_loads"]  # jumps do skip cells
    assert s["candidate_lanes"] >= c["brick_tests"] and s["shade_lanes"] >= c["extend_rays"] and s["connect_lanes"] == c["shadow_rays"]
    assert s["connect_runs"] <= s["shade_runs"]  # connect is part of the shade pass
    assert 0 <= s["step_lanes"] <= 64 * s["step_runs"] and 0 < s["jump_lanes"] <= 64 * s["jump_runs"] and s["waves"] > 0


def test_randomised_views_match_oracle(bm, orc, torch_cuda, scene256, world256):
    """Seeded sweep over camera position (inside, on the boundary, outside), angles, sun, lens and sample offsets."""
    rng = np.random.default_rng(20250614)
    world256.reset_device(True)
    for trial in range(24):
        pos = tuple(float(v) for v in rng.uniform(-80, 336, size=3))
        if trial % 4 == 0:  # exactly on a face / corner of the world box
            pos = (0.0, float(rng.uniform(0, 256)), 256.0)
        h, v = float(rng.uniform(-3.2, 3.2)), float(rng.uniform(-1.5, 1.5))
        sun = (float(rng.uniform(0, 1)), float(rng.uniform(0.02, 0.45)))
        lens = float(rng.choice([0.0, 0.0, 0.4]))
        W, H = int(rng.integers(17, 90)), int(rng.integers(9, 60))
        spp, mb, sb = int(rng.integers(1, 4)), int(rng.integers(0, 6)), int(rng.integers(0, 50))
        cam = bm.Camera(position=pos, horizontal_angle=h, vertical_angle=v, lensRadius=lens, focalDistance=float(rng.uniform(0.5, 3))).update()
        ocam = orc.make_camera(cam.position, cam.direction, focal_distance=cam.focalDistance, lens_radius=lens)
        p = bm.FrameParams(W, H, spp=spp, sample_base=sb, max_bounces=mb, sun_position=sun)
        acc, dbg = gpu_render(bm, torch_cuda, scene256, cam, p)
        oacc, odbg, _, _ = world256.render(ocam, orc.make_frame(W, H, spp=spp, max_bounces=mb, sample_base=sb, sun=sun))
        assert np.array_equal(dbg, odbg), f"trial {trial}: pos={pos} h={h} v={v}"
        assert_ray_digest(world256)
        finite = np.isfinite(oacc)
        assert np.array_equal(np.isfinite(acc), finite), f"trial {trial}"
        assert_radiance(np.where(finite, acc, 0), np.where(finite, oacc, 0))


def test_config3_like_streaming_lod_at_scale(bm, orc, torch_cuda):
    """BASELINE config 3 geometry at a reduced frame: 16^3 superchunks (2048^3 voxels), 8 segments, reference LoD
    thresholds, bricks streamed in on demand to a steady state; every 40th row compared with the oracle."""
    G, W, H = 2048, 960, 540
    scene = bm.Scene(G, G, device=0)
    scene.set_queue_capacity(1 << 20)
    scene.generate()
    cam, ocam = cameras(bm, orc, G)
    p = bm.FrameParams(W, H, spp=2, max_bounces=7)
    torch = torch_cuda
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    for _ in range(64):
        scene.render(cam, p, acc)
        if scene.process_load_queue() == 0:
            break
    else:
        pytest.fail("no streaming steady state")
    info = scene.info()
    assert 0 < info["resident_bricks"] < info["total_bricks"]  # only what the rays touched was ever uploaded
    acc, dbg = gpu_render(bm, torch, scene, cam, p)
    w = orc.World(G, G)
    w.reset_device(True)
    oacc, odbg, _, _ = w.render(ocam, orc.make_frame(W, H, spp=2, max_bounces=7, band_rows=1, shard_rank=11, shard_count=40), threads=os.cpu_count() or 1)
    rows = bm.dist.shard_rows(H, 1, 11, 40)
    assert np.array_equal(dbg[rows], odbg[rows])
    assert_radiance(acc[rows], oacc[rows])
    # ---- BASELINE config 4 at FULL size on this one GPU: 3840x2160, 16 spp, 8 segments, the same streamed world, as the
    # 8 row-band shards bench.py --gpus 8 hands out ((chunk, sample) work items, float-atomic adds); the assembled frame is
    # the frame of one launch tracing everything, and four of its rows are the oracle's
    full_size_sharded_job(bm, orc, torch, scene, cam, ocam, w, 3840, 2160, spp=16, shards=8, oracle_row_groups=540)
    scene.close()


def test_config3_own_shape_4k_streamed(bm, orc, torch_cuda):
    """BASELINE config 3 in ITS OWN shape -- 3840x2160, 4 spp, 8 segments, pixel work items, the 2048^3 world streamed in on demand
    to its steady state (voxel.cuh:228-245, Scene.cpp:200-252) -- i.e. the frame `bench.py --workload config3` times, rendered by the
    instantiation it runs there: 32 400 tiles take the XCD-aware hand-out (trace_paths<*, true, *>), production frames use helper
    lanes and, having four samples per pixel, (chunk, sample) work items; the ordered frames trace pixel items.  gpu_render renders it instrumented + ordered (hit records), production + ordered (same bits), production + helper
    lanes (same counts, radiance to 2e-5) and with the K-slot schedule; every 540th row (4 rows x 3840 pixels x 4 samples) is the
    oracle's, hit records bit for bit."""
    G, W, H, spp = 2048, 3840, 2160, 4
    torch = torch_cuda
    scene = bm.Scene(G, G, device=0)
    scene.set_queue_capacity(1 << 20)
    scene.generate()
    scene.reset_residency()
    cam, ocam = cameras(bm, orc, G)
    p = bm.FrameParams(W, H, spp=spp, max_bounces=7)
    assert ((W + 15) // 16) * ((H + 15) // 16) >= 32000  # scene.cpp frame_constants: this frame takes the XCD-aware hand-out
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    for _ in range(64):
        scene.render(cam, p, acc)
        if scene.process_load_queue() == 0:
            break
    else:
        pytest.fail("no streaming steady state")
    info = scene.info()
    assert 0 < info["resident_bricks"] < info["total_bricks"]
    del acc
    acc, dbg = gpu_render(bm, torch, scene, cam, p)
    assert scene.process_load_queue() == 0                      # steady state: the frames above asked for nothing new
    assert np.all(acc[..., 3] == spp)                           # every path of every pixel ended exactly once
    w = orc.World(G, G)
    w.reset_device(True)
    oacc, odbg, _, _ = w.render(ocam, orc.make_frame(W, H, spp=spp, max_bounces=7, band_rows=1, shard_rank=337, shard_count=540), threads=os.cpu_count() or 1)
    rows = bm.dist.shard_rows(H, 1, 337, 540)
    assert len(rows) == 4
    assert np.array_equal(dbg[rows], odbg[rows])
    assert_radiance(acc[rows], oacc[rows])
    scene.close()


def full_size_sharded_job(bm, orc, torch, scene, cam, ocam, oracle_world, W, H, spp, shards, oracle_row_groups, streamed=True):
    """One multi-GPU job of BASELINE (configs 4 / 5) run shard by shard on this GPU; see the callers."""
    band = bm.dist.DEFAULT_BAND_ROWS
    whole = bm.FrameParams(W, H, spp=spp, max_bounces=7)
    full = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    for _ in range(64 if streamed else 1):  # streaming steady state for THIS frame (more samples touch more bricks)
        full.zero_()
        scene.render(cam, whole, full)
        if not streamed or scene.process_load_queue() == 0:
            break
    else:
        pytest.fail("no streaming steady state at full size")
    dbg = torch.zeros((H, W, 8), dtype=torch.int32, device="cuda:0")
    scene.render(cam, whole, torch.zeros_like(full), debug=dbg)  # hit records of the same frame (instrumented instantiation)
    assembled = torch.zeros_like(full)
    digest = torch.zeros_like(dbg)
    for r in range(shards):
        p = bm.FrameParams(W, H, spp=spp, max_bounces=7, band_rows=band, shard_rank=r, shard_count=shards, flags=bm.BM_FLAG_SAMPLE_ITEMS)
        packed = torch.zeros((bm.local_rows(p), W, 4), dtype=torch.float32, device="cuda:0")
        scene.render(cam, p, packed)
        rows = torch.as_tensor(bm.dist.shard_rows(H, band, r, shards), device="cuda:0", dtype=torch.long)
        assembled.index_copy_(0, rows, packed)
        # the same shard once more with the instrumented instantiation: per-pixel digest of the (chunk, sample) items
        packed_dbg = torch.zeros((bm.local_rows(p), W, 8), dtype=torch.int32, device="cuda:0")
        scene.render(cam, p, torch.zeros_like(packed), debug=packed_dbg)
        digest.index_copy_(0, rows, packed_dbg)
    torch.cuda.synchronize()
    # ray counts and cells visited per pixel are sums in both modes: the shards' items traced exactly the rays of the whole frame
    assert torch.equal(digest[..., 6:], dbg[..., 6:]) and torch.equal(digest[..., :4], dbg[..., :4])
    if streamed:
        assert scene.process_load_queue() == 0  # the shards asked for nothing the whole frame had not
    a, f = assembled.cpu().numpy(), full.cpu().numpy()
    assert np.array_equal(a[..., 3], f[..., 3]) and float(f[..., 3].min()) == spp  # every path of every pixel ended, once
    assert np.allclose(a[..., :3], f[..., :3], rtol=2e-5, atol=1e-7)  # same paths, the samples of a pixel added in another order
    oacc, odbg, _, _ = oracle_world.render(ocam, orc.make_frame(W, H, spp=spp, max_bounces=7, band_rows=1, shard_rank=oracle_row_groups // 3,
                                                                 shard_count=oracle_row_groups), threads=os.cpu_count() or 1)
    rows = bm.dist.shard_rows(H, 1, oracle_row_groups // 3, oracle_row_groups)
    rows_dev = torch.as_tensor(rows, device="cuda:0", dtype=torch.long)
    assert np.array_equal(dbg[rows_dev].cpu().numpy().view(np.uint32), odbg[rows])
    assert_radiance(f[rows], oacc[rows])
    assert_radiance(a[rows], oacc[rows])
    # ... and the per-sample path hashes of the shards' work items sum to the oracle's on those rows: the mode the
    # multi-GPU job runs in is pinned on hits, bit for bit, at full size
    want = oracle_sample_digest(orc, oracle_world, ocam, W, H, spp, max_bounces=7, band_rows=1, shard_rank=oracle_row_groups // 3, shard_count=oracle_row_groups)
    assert np.array_equal(digest[rows_dev].cpu().numpy().view(np.uint32), want[rows])


def test_config5_like_lod_world_at_scale(bm, orc, torch_cuda):
    """BASELINE config 5 geometry: 32^3 superchunks (4096^3 voxels, 512 MiB index grid, ~4.3 GiB of bricks), reference
    LoD thresholds (all three levels occur), 8 segments, at a reduced frame; every 64th row compared with the oracle."""
    G, W, H = 4096, 1024, 576
    scene = bm.Scene(G, G, device=0)
    scene.set_queue_capacity(1 << 20)
    scene.generate().preload_all()
    info = scene.info()
    assert info["index_bytes"] == 512 * 1024 * 1024 and info["supercells"] == 32768
    cam, ocam = cameras(bm, orc, G)
    p = bm.FrameParams(W, H, spp=1, max_bounces=7, flags=bm.BM_FLAG_COUNTERS)
    scene.counters_reset()
    acc, dbg = gpu_render(bm, torch_cuda, scene, cam, p)
    cnt = scene.counters()
    assert cnt["byte_tests"] > 0 and cnt["brick_tests"] > 0
    levels = set(((dbg[..., 1] >> 12) & 0xF)[dbg[..., 1] != 0].tolist())
    assert {1, 2} <= levels, levels
    w = orc.World(G, G, threads=os.cpu_count() or 1)
    assert w.total_bricks() == info["total_bricks"]
    w.reset_device(True)
    oacc, odbg, _, _ = w.render(ocam, orc.make_frame(W, H, spp=1, max_bounces=7, band_rows=1, shard_rank=5, shard_count=64), threads=os.cpu_count() or 1)
    rows = bm.dist.shard_rows(H, 1, 5, 64)
    assert np.array_equal(dbg[rows], odbg[rows])
    assert_radiance(acc[rows], oacc[rows])
    # ---- the same view STREAMED: pools start at 16 bricks and double (Scene.cpp:231-251), the arena grows with
    # residency -- LoD keeps most of the world's ~4.3 GiB of bricks out of it -- and the steady-state image is the
    # resident one bit for bit
    assert info["pool_bytes"] == 64 * info["total_bricks"] <= info["brick_bytes"]
    scene.reset_residency()
    plain = bm.FrameParams(W, H, spp=1, max_bounces=7)
    for _ in range(200):
        gpu_render(bm, torch_cuda, scene, cam, plain, want_dbg=False)
        if scene.process_load_queue() == 0:
            break
    else:
        pytest.fail("streaming did not reach a steady state")
    acc_s, dbg_s = gpu_render(bm, torch_cuda, scene, cam, bm.FrameParams(W, H, spp=1, max_bounces=7))
    assert np.array_equal(dbg_s, dbg) and np.array_equal(acc_s, acc)
    streamed = scene.info()
    assert 0 < streamed["resident_bricks"] < info["total_bricks"] // 8
    assert 64 * streamed["resident_bricks"] <= streamed["pool_bytes"] <= streamed["brick_bytes"] < 64 * info["total_bricks"] // 4
    # ---- BASELINE config 5 at FULL size on this one GPU: 7680x4320, 32 spp, 8 segments (8.5 G nominal rays), streamed, as
    # the 8 row-band shards of the 8-GPU job; assembled frame == one launch tracing everything; two rows are the oracle's
    full_size_sharded_job(bm, orc, torch_cuda, scene, cam, ocam, w, 7680, 4320, spp=32, shards=8, oracle_row_groups=2160)
    scene.close()


def test_pool_growth_keeps_bricks_intact(bm, orc, torch_cuda):
    """Streaming with a tiny request ring: pools grow 16 -> 32 -> ... one batch at a time (moves between arena regions,
    freed regions reused); at steady state every loaded word points at exactly its brick."""
    G = 256
    scene = bm.Scene(G, G, device=0)
    scene.set_queue_capacity(48)
    scene.generate()
    w = orc.World(G, G)
    cam, _ = cameras(bm, orc, G)
    p = bm.FrameParams(200, 150, spp=1, max_bounces=3)
    for _ in range(2000):
        gpu_render(bm, torch_cuda, scene, cam, p, want_dbg=False)
        if scene.process_load_queue() == 0:
            break
    else:
        pytest.fail("streaming did not reach a steady state")
    info = scene.info()
    assert info["resident_bricks"] > 500 and info["pool_bytes"] >= 64 * info["resident_bricks"]
    assert max(int((scene.device_indices(sc) & bm.BRICK_LOADED_BIT != 0).sum()) for sc in range(w.nsc)) > 64  # some pool doubled at least twice
    for sc in range(w.nsc):
        dev, host_idx, host_bricks = scene.device_indices(sc), w.sc_indices(sc), w.sc_bricks(sc)
        loaded = np.flatnonzero(dev & bm.BRICK_LOADED_BIT)
        slots = dev[loaded] & 0xFFF
        assert sorted(slots.tolist()) == list(range(len(loaded)))  # request order: dense, each slot once
        for local in loaded[:: max(1, len(loaded) // 60)]:
            assert np.array_equal(scene.device_brick(sc, int(dev[local] & 0xFFF)), host_bricks[host_idx[local] & 0xFFF])
    scene.close()


def test_generate_supercell_is_refused_on_a_live_scene(bm, orc, torch_cuda):
    """Scene::generate_supercell rebuilds a HOST supercell and resets its slot counter; once pools hold bricks in request
    order that would hand out slots twice, so the call is refused after generate() (and the stream goes on unharmed)."""
    G = 256
    scene = bm.Scene(G, G, device=0)
    scene.generate_supercell(0, 0, 0)  # before generate(): allowed (host only, the reference's use)
    scene.set_queue_capacity(128)
    scene.generate()
    cam, ocam = cameras(bm, orc, G)
    p = bm.FrameParams(96, 64, spp=1, max_bounces=3)
    for _ in range(3):
        gpu_render(bm, torch_cuda, scene, cam, p, want_dbg=False)
        scene.process_load_queue()
    with pytest.raises(bm.BrickmapError):
        scene.generate_supercell(0, 0, 0)
    for _ in range(400):
        gpu_render(bm, torch_cuda, scene, cam, p, want_dbg=False)
        if scene.process_load_queue() == 0:
            break
    else:
        pytest.fail("no streaming steady state")
    acc, dbg = gpu_render(bm, torch_cuda, scene, cam, p)
    w = orc.World(G, G)
    w.reset_device(True)
    oacc, odbg, _, _ = w.render(ocam, orc.make_frame(96, 64, spp=1, max_bounces=3))
    assert np.array_equal(dbg, odbg)
    assert_radiance(acc, oacc)
    scene.close()


def test_streaming_scene_with_frames_on_several_streams(bm, orc, torch_cuda):
    """Frames of a STREAMING scene issued round-robin on three streams, both servicing modes: every stream is ordered behind
    the uploads it has not seen and process_load_queue behind the frames of all of them, so the scene streams in as it does
    on one stream -- every brick uploaded exactly once, nothing left half-requested, steady state == resident image."""
    torch = torch_cuda
    G, W, H = 256, 160, 120
    cam, _ = cameras(bm, orc, G)
    resident = bm.Scene(G, G, device=0).generate().preload_all()
    want_acc, want_dbg = gpu_render(bm, torch, resident, cam, bm.FrameParams(W, H, spp=1, sample_base=0, max_bounces=3))
    resident.close()
    for overlapped in (False, True):
        scene = bm.Scene(G, G, device=0)
        scene.set_queue_capacity(512)
        scene.generate()
        scene.set_streaming_mode(overlapped)
        streams = [torch.cuda.Stream() for _ in range(3)]
        bufs = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0") for _ in range(3)]
        torch.cuda.synchronize()
        total, idle = 0, 0
        for k in range(1200):
            j = k % 3
            scene.render(cam, bm.FrameParams(W, H, spp=1, sample_base=k, max_bounces=3), bufs[j], stream=streams[j].cuda_stream)
            n = scene.process_load_queue()
            total += n
            idle = idle + 1 if n == 0 else 0
            if idle >= 6:
                break
        assert idle >= 6
        torch.cuda.synchronize()
        info = scene.info()
        assert total == info["resident_bricks"] and info["failed"] == 0
        nsc = info["supercells"]
        loaded = sum(int(np.count_nonzero(scene.device_indices(sc) & bm.BRICK_LOADED_BIT)) for sc in range(nsc))
        requested = sum(int(np.count_nonzero(scene.device_indices(sc) & bm.BRICK_REQUESTED_BIT)) for sc in range(nsc))
        assert loaded == total and requested == 0
        acc, dbg = gpu_render(bm, torch, scene, cam, bm.FrameParams(W, H, spp=1, sample_base=0, max_bounces=3))
        assert np.array_equal(dbg, want_dbg) and np.array_equal(acc, want_acc)
        scene.close()


def test_arena_grows_without_stopping_the_world(bm, orc, torch_cuda):
    """The brick arena is a reserved address range that physical chunks are mapped into: the config-5 world streams in from
    nothing in overlapped mode with frames in flight, the arena grows several times, and none of the growths copies the
    arena or synchronises the device (arena_copy_growths == 0); the steady-state image is the oracle's on sampled rows."""
    G, W, H = 2048, 1024, 576
    scene = bm.Scene(G, G, device=0)
    scene.set_queue_capacity(1 << 18)
    scene.generate()
    scene.set_streaming_mode(True)
    info0 = scene.info()
    assert info0["arena_virtual"] == 1, "this device supports hipMemAddressReserve/hipMemMap: the arena must use it"
    cam, ocam = cameras(bm, orc, G)
    acc = torch_cuda.zeros((H, W, 4), dtype=torch_cuda.float32, device="cuda:0")
    idle = 0
    for k in range(400):
        scene.render(cam, bm.FrameParams(W, H, spp=2, max_bounces=7), acc)  # the same paths every frame: the request set converges
        idle = idle + 1 if scene.process_load_queue() == 0 else 0
        if idle >= 3:
            break
    assert idle >= 3
    info = scene.info()
    assert info["arena_growths"] >= 2 and info["arena_copy_growths"] == 0 and info["failed"] == 0
    assert info["brick_bytes"] > info0["brick_bytes"] and 64 * info["resident_bricks"] <= info["pool_bytes"] <= info["brick_bytes"]
    p = bm.FrameParams(W, H, spp=1, max_bounces=7)
    got, dbg = gpu_render(bm, torch_cuda, scene, cam, p)
    w = orc.World(G, G, threads=os.cpu_count() or 1)
    w.reset_device(True)
    oacc, odbg, _, _ = w.render(ocam, orc.make_frame(W, H, spp=1, max_bounces=7, band_rows=1, shard_rank=7, shard_count=48), threads=os.cpu_count() or 1)
    rows = bm.dist.shard_rows(H, 1, 7, 48)
    assert np.array_equal(dbg[rows], odbg[rows])
    assert_radiance(got[rows], oacc[rows])
    scene.close()


def test_arena_fallback_without_virtual_memory(bm, torch_cuda):
    """Devices without hipMemAddressReserve / hipMemMap reallocate + copy the arena behind a device synchronisation; the path is
    kept alive here by switching virtual memory management off (BM_ARENA_VMM=0) in a child process: the 2048^3 world streams in,
    the arena grows by copying, and the steady-state frame equals the same frame of a scene with the mapped arena."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
import brickmap_amd as bm
G, W, H = 2048, 640, 360
s = bm.Scene(G, G, device=0); s.set_queue_capacity(1 << 18); s.generate()
cam = bm.Camera(position=(G / 2, G / 8, 0.8 * G), horizontal_angle=0.8, vertical_angle=-0.5).update()
acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
p = bm.FrameParams(W, H, spp=2, max_bounces=7)
idle = 0
for k in range(400):
    s.render(cam, p, acc)
    idle = idle + 1 if s.process_load_queue() == 0 else 0
    if idle >= 2: break
assert idle >= 2
i = s.info()
p.flags |= bm.BM_FLAG_ORDERED  # (the frame whose sum is compared: reproducible sums)
out = torch.zeros_like(acc); s.render(cam, p, out); torch.cuda.synchronize()
print("RESULT", i["arena_virtual"], i["arena_growths"], i["arena_copy_growths"], i["resident_bricks"], float(out.double().sum().item()))
""" % root
    results = []
    for vmm in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=dict(os.environ, BM_ARENA_VMM=vmm), cwd=root)
        assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
        results.append([l for l in r.stdout.splitlines() if l.startswith("RESULT")][-1].split()[1:])
    (v0, g0, c0, n0, sum0), (v1, g1, c1, n1, sum1) = results
    assert v0 == "0" and int(g0) >= 1 and int(c0) == int(g0)  # fallback: every growth copied
    assert v1 == "1" and int(g1) >= 1 and int(c1) == 0        # mapped arena: none did
    assert n0 == n1 and sum0 == sum1                            # same residency, same frame


def test_overlapped_streaming_reaches_the_resident_image(bm, orc, torch_cuda):
    """Overlapped request servicing (two rings, no host wait): same steady state as the all-resident scene, every
    brick uploaded exactly once, requests land two calls after they were raised."""
    G, W, H = 256, 160, 120
    scene = bm.Scene(G, G, device=0)
    scene.set_queue_capacity(256)  # several frames of back-pressure on the ring
    scene.generate()
    scene.set_streaming_mode(True)
    cam, _ = cameras(bm, orc, G)
    p = bm.FrameParams(W, H, spp=1, max_bounces=3)
    torch = torch_cuda
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    scene.render(cam, p, acc)
    assert scene.process_load_queue() == 0          # call 1 only starts the copy-out of frame 1's ring
    scene.render(cam, p, acc)
    assert scene.process_load_queue() == 256        # call 2 services frame 1's (full) ring
    total, idle = 256, 0
    for _ in range(400):
        scene.render(cam, p, acc)
        n = scene.process_load_queue()
        total += n
        idle = idle + 1 if n == 0 else 0
        if idle >= 3:
            break
    assert idle >= 3
    info = scene.info()
    assert total == info["resident_bricks"] <= info["total_bricks"]
    loaded = sum(int(np.count_nonzero(scene.device_indices(sc) & bm.BRICK_LOADED_BIT)) for sc in range(8))
    requested = sum(int(np.count_nonzero(scene.device_indices(sc) & bm.BRICK_REQUESTED_BIT)) for sc in range(8))
    assert loaded == total and requested == 0       # exactly once each, nothing left half-requested
    a_stream, d_stream = gpu_render(bm, torch, scene, cam, p)
    scene.process_load_queue()
    scene.process_load_queue()
    scene.set_streaming_mode(False)
    scene.preload_all()
    a_res, d_res = gpu_render(bm, torch, scene, cam, p)
    assert np.array_equal(d_stream, d_res) and np.array_equal(a_stream, a_res)
    scene.close()


def test_overlapped_streaming_matches_oracle_frame_by_frame(bm, orc, torch_cuda):
    """The two-ring mode against the oracle's model of it (orc_process_load_queue_overlapped): a request raised in frame
    k is resident from frame k+2 on.  Hit records, request / residency flags and upload counts agree frame by frame
    while the scene streams in (ring large enough that nothing overflows: the request SET is then deterministic)."""
    G, W, H = 256, 96, 64
    scene = bm.Scene(G, G, device=0)
    scene.set_queue_capacity(1 << 16)
    scene.generate()
    scene.set_streaming_mode(True)
    w = orc.World(G, G)
    w.set_queue_cap(1 << 16)
    w.reset_device(False)
    cam, ocam = cameras(bm, orc, G)
    uploads = []
    for k in range(7):
        p = bm.FrameParams(W, H, spp=1, sample_base=k, max_bounces=3)
        acc, dbg = gpu_render(bm, torch_cuda, scene, cam, p)
        oacc, odbg, _, _ = w.render(ocam, orc.make_frame(W, H, spp=1, sample_base=k, max_bounces=3))
        assert np.array_equal(dbg, odbg), f"frame {k}"
        assert_radiance(acc, oacc)
        for sc in range(w.nsc):  # same requested / unloaded / loaded flags and LoD bytes (slots follow the request order, which differs)
            assert np.array_equal(scene.device_indices(sc) & ~np.uint32(0xFFF), w.sc_dev_indices(sc) & ~np.uint32(0xFFF)), f"frame {k} supercell {sc}"
        n_gpu, n_cpu = scene.process_load_queue(), w.process_load_queue_overlapped()
        assert n_gpu == n_cpu, f"call {k}"
        uploads.append(n_gpu)
    assert uploads[0] == 0 and uploads[1] > 0  # call 1 only copies out frame 1's ring, call 2 uploads it
    for _ in range(2):  # drain what the last two frames asked for
        n_gpu, n_cpu = scene.process_load_queue(), w.process_load_queue_overlapped()
        assert n_gpu == n_cpu
        uploads.append(n_gpu)
    assert sum(uploads) == scene.info()["resident_bricks"]
    scene.close()


def test_reference_flythrough_views_on_native_world(bm, orc, torch_cuda):
    """The reference's native world (4096 x 4096 x 512) from two of its fly-through viewpoints (one inside, one far
    outside the world box, performance_measure.h:4-25); every 54th row compared with the oracle."""
    scene = bm.Scene(4096, 512, device=0).generate().preload_all()
    assert scene.info()["total_bricks"] == 8663747  # SURVEY.md [probe]
    w = orc.World(4096, 512, threads=os.cpu_count() or 1)
    w.reset_device(True)
    W, H = 960, 540
    for view in (0, 4):
        cam = bm.flythrough_camera(view)
        ocam = orc.make_camera(cam.position, cam.direction)
        assert np.array_equal(np.float32(cam.direction), orc.camera_direction(*bm.FLYTHROUGH_VIEWS[view][1]))
        acc, dbg = gpu_render(bm, torch_cuda, scene, cam, bm.FrameParams(W, H, spp=1, max_bounces=3))
        oacc, odbg, _, _ = w.render(ocam, orc.make_frame(W, H, spp=1, max_bounces=3, band_rows=1, shard_rank=3, shard_count=54), threads=os.cpu_count() or 1)
        rows = bm.dist.shard_rows(H, 1, 3, 54)
        assert np.array_equal(dbg[rows], odbg[rows]), f"view {view}"
        assert_radiance(acc[rows], oacc[rows])
        assert np.count_nonzero(dbg[..., 1]) > 0  # the world is in view
    scene.close()


def test_sample_items_mode_matches_pixel_items(bm, orc, torch_cuda, scene256):
    """BM_FLAG_SAMPLE_ITEMS only changes how work is dealt out ((4x4 chunk, sample) items, float atomics): every sample is
    the same path, so path counts are exact and radiance differs by summation order only."""
    cam, _ = cameras(bm, orc, 256)
    for kw in (dict(spp=5, sample_base=3), dict(spp=3, band_rows=16, shard_rank=1, shard_count=2), dict(spp=1)):
        p = bm.FrameParams(150, 90, max_bounces=3, **kw)
        want, _ = gpu_render(bm, torch_cuda, scene256, cam, p, want_dbg=False)
        pi = bm.FrameParams(150, 90, max_bounces=3, flags=bm.BM_FLAG_SAMPLE_ITEMS, **kw)
        got, _ = gpu_render(bm, torch_cuda, scene256, cam, pi, want_dbg=False)
        assert np.array_equal(got[..., 3], want[..., 3])  # terminated paths per pixel: small integers, exact in any order
        np.testing.assert_allclose(got[..., :3], want[..., :3], rtol=2e-6, atol=1e-9)
    # accumulates onto existing content like the default mode
    acc = torch_cuda.full((90, 150, 4), 2.0, dtype=torch_cuda.float32, device="cuda:0")
    got, _ = gpu_render(bm, torch_cuda, scene256, cam, bm.FrameParams(150, 90, spp=2, flags=bm.BM_FLAG_SAMPLE_ITEMS), accum=acc, want_dbg=False)
    want, _ = gpu_render(bm, torch_cuda, scene256, cam, bm.FrameParams(150, 90, spp=2), accum=torch_cuda.full((90, 150, 4), 2.0, dtype=torch_cuda.float32, device="cuda:0"), want_dbg=False)
    np.testing.assert_allclose(got, want, rtol=2e-6)


def oracle_sample_digest(orc, world, ocam, W, H, spp, sample_base=0, max_bounces=3, **shard):
    """What debug_dev holds under BM_FLAG_SAMPLE_ITEMS, from the oracle: words 0-3 the first-hit record of the first sample,
    words 4-7 the sums (mod 2^32) over the samples of the per-sample path hashes / ray counts / cell counts -- every sample
    rendered on its own, so each hash chain covers exactly one path."""
    total = None
    for k in range(spp):
        _, d, _, _ = world.render(ocam, orc.make_frame(W, H, spp=1, sample_base=sample_base + k, max_bounces=max_bounces, **shard), threads=os.cpu_count() or 1)
        d = d.astype(np.uint32)
        if total is None:
            total = d.copy()
        else:
            total[..., 4:] += d[..., 4:]
    return total


def test_sample_items_digest_matches_oracle(bm, orc, torch_cuda, scene256, world256):
    """The work-item mode every multi-GPU shard uses, pinned on GEOMETRY: per pixel the sum over its samples of the per-path
    hit hashes (distance bits, normal, level, brick id, voxel id of every segment; occlusion + occluder of every shadow ray),
    of the ray counts and of the cells visited equals the oracle's, bit for bit; radiance within 1e-4."""
    torch = torch_cuda
    world256.reset_device(True)
    for (W, H, kw) in ((150, 90, dict(spp=5, sample_base=3)), (96, 64, dict(spp=3, band_rows=16, shard_rank=1, shard_count=2)), (64, 48, dict(spp=1))):
        for pos, h, v in (((128, 32, 204.8), 0.8, -0.5), ((40.5, 200.25, 150.0), 2.3, -0.2)):
            cam, ocam = cameras(bm, orc, 256, pos=pos, h=h, v=v)
            p = bm.FrameParams(W, H, max_bounces=3, flags=bm.BM_FLAG_SAMPLE_ITEMS, **kw)
            rows = bm.local_rows(p)
            acc = torch.zeros((rows, W, 4), dtype=torch.float32, device="cuda:0")
            dbg = torch.zeros((rows, W, 8), dtype=torch.int32, device="cuda:0")
            scene256.render(cam, p, acc, debug=dbg)
            plain = torch.zeros_like(acc)
            scene256.render(cam, p, plain)  # the production instantiation in the same mode
            torch.cuda.synchronize()
            shard = {k: kw[k] for k in ("band_rows", "shard_rank", "shard_count") if k in kw}
            want = oracle_sample_digest(orc, world256, ocam, W, H, kw["spp"], kw.get("sample_base", 0), **shard)
            mine = bm.dist.shard_rows(H, kw.get("band_rows", H), kw.get("shard_rank", 0), kw.get("shard_count", 1))
            got = dbg.cpu().numpy().view(np.uint32)
            assert np.array_equal(got, want[mine]), "sample-item digest differs from the oracle"
            oacc, _, _, _ = world256.render(ocam, orc.make_frame(W, H, max_bounces=3, **kw), threads=os.cpu_count() or 1)
            a, b = acc.cpu().numpy(), plain.cpu().numpy()
            assert np.array_equal(a[..., 3], b[..., 3]) and np.array_equal(a[..., 3], oacc[mine][..., 3])
            assert_radiance(a, oacc[mine])
            assert_radiance(b, oacc[mine])


def test_bench_multi_rank_path_on_one_gpu(bm, torch_cuda, tmp_path):
    """`bench.py --gpus 2` end to end -- row-band shards with (chunk, sample) work items, steps issued as frame-ring launches with one
    gather per batch to rank 0 (the driver's command line; then two steps per launch, so that the timed region holds a full and a
    short batch; then one launch per step; then the sample decomposition),
    the gathered / reduced frames compared with one GPU rendering everything (--verify),
    max-over-ranks timing, one JSON line -- with both ranks on this GPU and gloo instead of RCCL (BM_BENCH_SHARE_GPU=1).
    The 8-GPU run itself is the driver's; this pins the code path it takes."""
    import json
    import socket
    import subprocess
    import sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from conftest import build_fake_rccl
    fake = build_fake_rccl(tmp_path)
    # the last two runs take the C-ABI exchange (bm_gather_frame / bm_reduce_frame, what an RCCL group uses) over the stand-in transport
    for extra, capi in (([], False), (["--frames-per-launch", "2"], False), (["--frames-per-launch", "1"], False), (["--decomposition", "samples"], False), ([], True),
                        (["--frames-per-launch", "2"], True), (["--decomposition", "samples"], True)):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--workload", "config1", "--verify"] + extra
        env = dict(os.environ, BM_BENCH_SHARE_GPU="1", OMP_NUM_THREADS="1")
        if capi:
            env.update(BM_DIST_CAPI="1", BM_RCCL_LIBRARY=fake)
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(line) == 1, r.stdout[-1500:]
        out = json.loads(line[0])
        assert out["n_gpus"] == 2 and out["steps"] == 3 and out["value"] > 0 and out["scaling"] == "strong"
        assert out["config"]["spp_per_step"] == 8 and "roofline" in out and "cpu_baseline" not in out
        assert out["verified_against_single_gpu"]["frames"] == 5 and out["verified_against_single_gpu"]["max_rel_err"] < 1e-5
        want_f = int(extra[1]) if "--frames-per-launch" in extra else 3  # (default: min(5, steps) per launch and per exchange; samples: all steps)
        assert out["config"]["frames_per_launch"] == want_f and sum(out["roofline"]["frames_per_launch"]) == 3
        assert ("C-ABI" in out["config"]["exchange"]) == capi
        port += 1


def test_long_paths_and_ticket_limits(bm, orc, torch_cuda, scene256, world256):
    """max_bounces = 16 (17 segments per path): the schedule has no limit on the path length and matches the oracle; a frame whose
    tickets would wrap the 32-bit hand-out counters is refused when the caller ASKED for (chunk, sample) items, and rendered with pixel
    items (which carry no spp factor) when the items were only the library's own choice."""
    torch = torch_cuda
    cam, ocam = cameras(bm, orc, 256)
    W, H = 96, 64
    p = bm.FrameParams(W, H, spp=1, max_bounces=16)
    a, dbg = gpu_render(bm, torch, scene256, cam, p)
    oacc, odbg, _, _ = world256.render(ocam, orc.make_frame(W, H, spp=1, max_bounces=16))
    assert np.array_equal(dbg, odbg)
    assert_radiance(a, oacc)
    assert int((dbg[..., 6] & 0xFFFF).max()) > 5  # some paths really are long
    # 32-bit ticket counters: (chunk, sample) items of a 4K frame at 100 000 spp would wrap them
    with pytest.raises(Exception) as e:
        scene256.render(cam, bm.FrameParams(3840, 2160, spp=100000, max_bounces=3, flags=bm.BM_FLAG_SAMPLE_ITEMS), torch.zeros((2160, 3840, 4), dtype=torch.float32, device="cuda:0"))
    assert "ticket" in str(e.value)


def test_bench_starts_its_own_ranks(bm, torch_cuda, tmp_path):
    """`python bench.py --gpus 2 ...` with NO launcher and no WORLD_SIZE in the environment (how the driver starts the N = 1 run, and
    possibly the scaling runs): bench.py re-runs itself under torch.distributed.run, one rank per GPU, and rank 0's single JSON line
    comes out of the parent's stdout -- with the per-rank clocks, the devices and the communicator's own rank count in it.
    (Both ranks share this GPU: gloo group; the second run takes the C-ABI exchange over the stand-in transport.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from conftest import build_fake_rccl
    fake = build_fake_rccl(tmp_path)
    for capi in (False, True):
        env = dict(os.environ, BM_BENCH_SHARE_GPU="1", OMP_NUM_THREADS="1")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "BM_BENCH_FORCE_DIST"):
            env.pop(k, None)
        if capi:
            env.update(BM_DIST_CAPI="1", BM_RCCL_LIBRARY=fake)
        cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "config1", "--verify"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(line) == 1, r.stdout[-1500:]
        out = json.loads(line[0])
        assert out["n_gpus"] == 2 and out["steps"] == 3 and out["value"] > 0
        assert out["verified_against_single_gpu"]["frames"] == 4 and out["verified_against_single_gpu"]["max_rel_err"] < 1e-5
        ranks = out["ranks"]
        assert ranks["process_group_world"] == 2 and len(ranks["ms_per_step"]) == 2 and len(ranks["devices"]) == 2 and all(t > 0 for t in ranks["ms_per_step"])
        assert ranks["communicator_world"] == (2 if capi else None)
        assert abs(max(ranks["ms_per_step"]) - out["ms_per_step"]) < 1e-3  # the line's time is the slowest rank's


def test_frames_overlapping_on_two_streams(bm, orc, torch_cuda, scene256):
    """Consecutive frames issued on two streams (what a host that pipelines frames over streams does) may run at the same time: every launch has its
    own ticket counters and constants, and with BM_FLAG_SAMPLE_ITEMS samples are added atomically, so the buffer ends up
    with the same paths as the frames rendered one after the other."""
    torch = torch_cuda
    cam, _ = cameras(bm, orc, 256)
    W, H, frames = 320, 200, 8
    want = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    for i in range(frames):
        scene256.render(cam, bm.FrameParams(W, H, spp=1, sample_base=i, max_bounces=3), want)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(4)]
    got = torch.zeros_like(want)
    torch.cuda.synchronize()
    for i in range(frames):
        scene256.render(cam, bm.FrameParams(W, H, spp=1, sample_base=i, max_bounces=3, flags=bm.BM_FLAG_SAMPLE_ITEMS), got, stream=streams[i % 4].cuda_stream)
    torch.cuda.synchronize()
    a, b = got.cpu().numpy(), want.cpu().numpy()
    assert np.array_equal(a[..., 3], b[..., 3])  # terminated paths per pixel: exact in any order
    np.testing.assert_allclose(a[..., :3], b[..., :3], rtol=2e-6, atol=1e-9)


@pytest.mark.gpu
def test_refill_threshold_and_handout_order_change_nothing_but_time(bm, torch_cuda, tmp_path):
    """FrameConstants::xcd_handout (256x256-pixel super-tiles dealt to the XCDs' counters: big frames by rule, BM_XCD_HANDOUT forces it) and
    FrameConstants::refill_min (scene.cpp frame_constants: 16 / 8 / 4 idle lanes by the samples per work item, BM_REFILL_MIN
    overrides it per process) decides WHEN a wave takes new pixels, never what a path computes: the same frames -- 1, 2 and 5 samples
    per pixel, pixel items and (chunk, sample) items' hit digests -- bit for bit under the rule and under forced thresholds."""
    code = r'''
import sys, hashlib
sys.path.insert(0, %r)
import numpy as np, torch, brickmap_amd as bm
scene = bm.Scene(256, 256, device=0).generate().preload_all()
cam = bm.Camera(position=(128.0, 40.0, 200.0), horizontal_angle=0.9, vertical_angle=-0.6).update()
h = hashlib.sha256()
for W, H, spp, flags in ((97, 61, 1, 0), (97, 61, 2, 0), (97, 61, 5, 0), (97, 61, 3, bm.BM_FLAG_SAMPLE_ITEMS), (700, 500, 1, 0)):  # (the last: 3 x 2 super-tiles, cut by the frame's edge)
    p = bm.FrameParams(W, H, spp=spp, max_bounces=3, flags=flags)
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    dbg = torch.zeros((H, W, 8), dtype=torch.int32, device="cuda:0")
    scene.render(cam, p, acc, debug=dbg)
    torch.cuda.synchronize()
    h.update(dbg.cpu().numpy().tobytes())
    if not flags:
        h.update(acc.cpu().numpy().tobytes())   # (float atomics: the sum's order is free with sample items)
print("DIGEST", h.hexdigest())
''' % ROOT
    import subprocess
    import sys
    digests = {}
    for v in ("", "1", "4", "16", "33", "64", "xcd", "xcd4"):
        env = dict(os.environ)
        env.pop("BM_REFILL_MIN", None)
        env.pop("BM_XCD_HANDOUT", None)
        if v.startswith("xcd"):  # the XCD-aware hand-out (big frames by rule) forced onto these small ones: another order, the same pixels
            env["BM_XCD_HANDOUT"] = "1"
            if v[3:]:
                env["BM_REFILL_MIN"] = v[3:]
        elif v:
            env["BM_REFILL_MIN"] = v
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        digests[v or "rule"] = [l for l in r.stdout.splitlines() if l.startswith("DIGEST")][0]
    assert len(set(digests.values())) == 1, digests


@pytest.mark.gpu
def test_probed_streams_overlap_frames(bm, torch_cuda, scene256):
    """bm_probe_streams hands out streams that demonstrably run side by side: distinct handles, and consecutive frames on two of them
    (one accumulation buffer each) take clearly less time than the same frames on one -- the next frame's workgroups fill the slots
    the previous one frees while it works its last paths off."""
    import time
    torch = torch_cuda
    handles = bm.probe_streams(3, device=0)
    try:
        assert len(set(handles)) == 3 and all(handles)
        cam = bm.Camera(position=(128.0, 32.0, 205.0), horizontal_angle=0.8, vertical_angle=-0.5).update()
        W, H, n = 960, 540, 40
        bufs = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0") for _ in range(2)]

        def run(streams):
            best = 1e9
            for rep in range(3):
                torch.cuda.synchronize()
                t = time.perf_counter()
                for i in range(n):
                    scene256.render(cam, bm.FrameParams(W, H, spp=1, sample_base=i, max_bounces=3), bufs[i % len(streams)], stream=streams[i % len(streams)])
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t)
            return best
        one, two = run(handles[:1]), run(handles[:2])
        assert two < 0.95 * one, (one, two)
    finally:
        bm.release_streams(handles)
