"""bm_render_frames -- the reference's per-frame loop (main.cpp:117-147: one launch_kernels call per frame) as ONE launch of the
persistent kernel (csrc/trace.hip "FRAME RING").  A frame of a launch must be the frame a single launch renders: ordered frames bit
for bit, hit records against the oracle; production frames the same paths, sums in another order."""
import copy

import numpy as np
import pytest

from test_gpu_parity import assert_radiance, cameras

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


@pytest.fixture(scope="module")
def scene256(bm, torch_cuda):
    s = bm.Scene(256, 256, device=0).generate()
    s.preload_all()
    yield s
    s.close()


def fly(bm, orc, k):
    """camera k of a short fly-through over the 256^3 world (every frame of a launch its own view)"""
    return cameras(bm, orc, 256, pos=(128.0 + 9.0 * k, 32.0 + 5.0 * k, 205.0 - 3.0 * k), h=0.8 + 0.07 * k, v=-0.5 - 0.03 * k)


@pytest.mark.parametrize("spp", [1, 3])
def test_ordered_frames_of_one_launch_are_the_single_launches(spp, bm, orc, torch_cuda, scene256, world256):
    """K frames with their own cameras, suns, sample_base and base_frame: as ONE launch (ordered, a buffer each, hit records for the
    first and the last frame) and as K single launches -- identical accumulator bits and hit records; the first and the last
    frame against the oracle."""
    torch = torch_cuda
    W, H, K, mb = 150, 100, 6, 3
    cams = [fly(bm, orc, k) for k in range(K)]
    suns = [(0.05 + 0.04 * k, 0.1 + 0.03 * k) for k in range(K)]
    params = [bm.FrameParams(W, H, spp=spp, sample_base=7 * k, max_bounces=mb, base_frame=1 + k, sun_position=suns[k]) for k in range(K)]
    accs = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0") for _ in range(K)]
    dbgs = [torch.zeros((H, W, 8), dtype=torch.int32, device="cuda:0") if k in (0, K - 1) else None for k in range(K)]
    scene256.render_frames([c for c, _ in cams], params, accs, debugs=dbgs)
    torch.cuda.synchronize()
    for k in range(K):
        single = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
        sd = torch.zeros((H, W, 8), dtype=torch.int32, device="cuda:0") if dbgs[k] is not None else None
        po = copy.copy(params[k])
        po.flags |= bm.BM_FLAG_ORDERED
        scene256.render(cams[k][0], po, single, debug=sd)
        torch.cuda.synchronize()
        assert torch.equal(single.view(torch.int32), accs[k].view(torch.int32)), f"frame {k} of the launch differs from its single launch"
        if sd is not None:
            assert torch.equal(sd, dbgs[k])
    for k in (0, K - 1):
        world256.reset_device(True)
        oacc, odbg, _, _ = world256.render(cams[k][1], orc.make_frame(W, H, spp=spp, max_bounces=mb, sample_base=7 * k, base_frame=1 + k, sun=suns[k]))
        assert np.array_equal(dbgs[k].cpu().numpy().view(np.uint32), odbg), f"frame {k}: hit records differ from the oracle"
        assert_radiance(accs[k].cpu().numpy(), oacc)


def test_production_frames_of_one_launch_share_a_buffer(bm, orc, torch_cuda, scene256, world256):
    """The bench's shape: K production frames (helper lanes, float atomics) of ONE launch add into ONE buffer, like consecutive frames
    of the reference's accumulation (kernel.cu:319-322,341-343): terminated-path counts exact, radiance = the sum of the K single
    launches up to summation order, and = the oracle's K-sample frame."""
    torch = torch_cuda
    W, H, K = 200, 120, 8
    cam, ocam = cameras(bm, orc, 256)
    for spp in (1, 2):
        params = [bm.FrameParams(W, H, spp=spp, sample_base=spp * k, max_bounces=3) for k in range(K)]
        ring = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
        scene256.render_frames(cam, params, ring)
        want = torch.zeros_like(ring)
        for p in params:
            scene256.render(cam, p, want)
        torch.cuda.synchronize()
        a, b = ring.cpu().numpy(), want.cpu().numpy()
        assert np.array_equal(a[..., 3], b[..., 3]) and np.all(a[..., 3] == K * spp)
        np.testing.assert_allclose(a[..., :3], b[..., :3], rtol=2e-5, atol=1e-7)
        world256.reset_device(True)
        oacc, _, _, _ = world256.render(ocam, orc.make_frame(W, H, spp=K * spp, max_bounces=3), want_dbg=False)
        assert_radiance(a, oacc)


def test_uniform_launch_mixes_frames_inside_a_wave(bm, orc, torch_cuda, scene256, world256):
    """A UNIFORM launch -- one view, sample_base and buffers stepping by constants (a resting camera; a rank's batch in one allocation):
    lanes of consecutive frames share a wave (the frame is folded into the lane's sample index and pixel offset).  Ordered frames into one
    allocation are still the single launches bit for bit; the same frames into scattered buffers (not uniform: waves switch frame by
    frame) give the same bits; sample strides other than spp, (chunk, sample) items and a shard included."""
    torch = torch_cuda
    cam, ocam = cameras(bm, orc, 256)
    for (W, H, K, spp, stride, extra) in ((150, 100, 6, 1, 1, {}), (97, 61, 5, 2, 7, {}), (128, 96, 4, 3, 3, dict(flags=bm.BM_FLAG_SAMPLE_ITEMS)),
                                          (160, 100, 4, 2, 2, dict(band_rows=8, shard_rank=1, shard_count=3, flags=bm.BM_FLAG_SAMPLE_ITEMS))):
        ordered = not extra.get("flags", 0) & bm.BM_FLAG_SAMPLE_ITEMS
        flags = extra.get("flags", 0) | (bm.BM_FLAG_ORDERED if ordered else 0)
        kw = {k: v for k, v in extra.items() if k != "flags"}
        params = [bm.FrameParams(W, H, spp=spp, sample_base=11 + stride * k, max_bounces=3, flags=flags, **kw) for k in range(K)]
        rows = bm.local_rows(params[0])
        batch = torch.zeros((K, rows, W, 4), dtype=torch.float32, device="cuda:0")       # one allocation: uniform
        scene256.render_frames(cam, params, [batch[k] for k in range(K)])
        scattered = [torch.zeros((rows, W, 4), dtype=torch.float32, device="cuda:0") for _ in range(K)]  # separate allocations: not uniform
        pad = [torch.zeros(1000 * (k + 1), device="cuda:0") for k in range(K)]  # (keep the allocator from spacing them evenly)
        scene256.render_frames(cam, params, scattered)
        torch.cuda.synchronize()
        for k in range(K):
            single = torch.zeros((rows, W, 4), dtype=torch.float32, device="cuda:0")
            scene256.render(cam, params[k], single)
            torch.cuda.synchronize()
            if ordered:
                assert torch.equal(batch[k].view(torch.int32), single.view(torch.int32)), f"uniform launch, frame {k}"
                assert torch.equal(scattered[k].view(torch.int32), single.view(torch.int32)), f"frame-by-frame launch, frame {k}"
            else:
                for got in (batch[k], scattered[k]):
                    assert torch.equal(got[..., 3], single[..., 3])
                    np.testing.assert_allclose(got[..., :3].cpu().numpy(), single[..., :3].cpu().numpy(), rtol=2e-5, atol=1e-7)
        del pad
    # the bench's shape against the oracle: K production frames of one view into ONE buffer = the oracle's K-sample frame
    W, H, K = 200, 120, 8
    ring = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    scene256.render_frames(cam, [bm.FrameParams(W, H, spp=1, sample_base=k, max_bounces=3) for k in range(K)], ring)
    torch.cuda.synchronize()
    world256.reset_device(True)
    oacc, _, _, _ = world256.render(ocam, orc.make_frame(W, H, spp=K, max_bounces=3), want_dbg=False)
    assert np.all(ring[..., 3].cpu().numpy() == K)
    assert_radiance(ring.cpu().numpy(), oacc)
    # ... on HITS, bit for bit, and on the traversal counters: the instrumented sibling of the kernel bench.py times
    # (trace_paths<true, *, true, 2>: helper lanes, lanes of several frames in one wave) writes the ray digest of the whole launch into one
    # buffer -- the oracle's digest of the K-sample frame -- and counts the oracle's cells, brick tests and rays
    for spp in (1, 2):
        acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
        dig = torch.zeros((H, W, 8), dtype=torch.int32, device="cuda:0")
        ps = [bm.FrameParams(W, H, spp=spp, sample_base=5 + spp * k, max_bounces=3, flags=bm.BM_FLAG_RAY_DIGEST | bm.BM_FLAG_COUNTERS) for k in range(K)]
        scene256.counters_reset()
        scene256.render_frames(cam, ps, acc, debugs=[dig] * K)
        torch.cuda.synchronize()
        got_cnt = scene256.counters()
        scene256.counters_reset()
        world256.reset_device(True)
        oacc, _, ocnt, _ = world256.render(ocam, orc.make_frame(W, H, spp=K * spp, sample_base=5, max_bounces=3))
        assert np.array_equal(dig.cpu().numpy().view(np.uint32), world256.last_ray_digest), "uniform launch: ray digest differs from the oracle's"
        assert got_cnt == ocnt
        assert_radiance(acc.cpu().numpy(), oacc)
    # one hit-record buffer for frames of DIFFERENT views is refused (its keys and its first-hit record would not be defined)
    with pytest.raises(bm.BrickmapError, match="uniform launch"):
        scene256.render_frames([cam, fly(bm, orc, 3)[0]], ps[:2], acc, debugs=[dig, dig])


def test_counters_of_a_launch_are_the_sum_of_its_frames(bm, orc, torch_cuda, scene256, world256):
    """BM_FLAG_COUNTERS on every frame of a launch: the traversal counters are the oracle's, summed over the frames (the instrumented
    instantiation with helper lanes walks the ring like the plain one)."""
    torch = torch_cuda
    W, H, K = 96, 64, 4
    cams = [fly(bm, orc, k) for k in range(K)]
    params = [bm.FrameParams(W, H, spp=1, sample_base=k, max_bounces=3, flags=bm.BM_FLAG_COUNTERS) for k in range(K)]
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    scene256.counters_reset()
    scene256.render_frames([c for c, _ in cams], params, acc)
    torch.cuda.synchronize()
    got = scene256.counters()
    scene256.counters_reset()
    total = {k: 0 for k in got}
    for k in range(K):
        world256.reset_device(True)
        _, _, ocnt, _ = world256.render(cams[k][1], orc.make_frame(W, H, spp=1, max_bounces=3, sample_base=k), want_dbg=False)
        for name in total:
            total[name] += ocnt[name]
    assert got == total


def test_ring_wraps_and_many_launches_in_flight(bm, torch_cuda, scene256, orc):
    """More frames in flight than the scene's ring of constants / ticket counters holds (1024 entries, 256 launches): entries are reused
    only after the launch that used them has finished -- no frame is lost or traced twice (alpha counts terminated paths)."""
    torch = torch_cuda
    W, H = 48, 32
    cam, _ = cameras(bm, orc, 256)
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    frames = 0
    for launch in range(40):
        n = (100, 256, 1, 37)[launch % 4]
        scene256.render_frames(cam, [bm.FrameParams(W, H, spp=1, sample_base=frames + i, max_bounces=2) for i in range(n)], acc)
        frames += n
    for i in range(300):  # single launches through the same ring
        scene256.render(cam, bm.FrameParams(W, H, spp=1, sample_base=frames, max_bounces=2), acc)
        frames += 1
    torch.cuda.synchronize()
    assert torch.all(acc[..., 3] == frames)
    t = scene256.render_times(256)
    assert len(t) == 256 and np.all(t > 0)


def test_sharded_frames_of_one_launch(bm, orc, torch_cuda, scene256):
    """A rank's shard of K frames as one launch (what bench.py --gpus N issues): packed row bands, (chunk, sample) items -- equal to the
    rows of the unsharded frames."""
    torch = torch_cuda
    W, H, K, N, band = 160, 100, 4, 3, 8
    cam, _ = cameras(bm, orc, 256)
    full = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0") for _ in range(K)]
    for k in range(K):
        scene256.render(cam, bm.FrameParams(W, H, spp=2, sample_base=2 * k, max_bounces=3, flags=bm.BM_FLAG_ORDERED), full[k])
    for r in range(N):
        params = [bm.FrameParams(W, H, spp=2, sample_base=2 * k, max_bounces=3, flags=bm.BM_FLAG_SAMPLE_ITEMS, band_rows=band, shard_rank=r, shard_count=N) for k in range(K)]
        rows = bm.local_rows(params[0])
        packed = [torch.zeros((rows, W, 4), dtype=torch.float32, device="cuda:0") for _ in range(K)]
        scene256.render_frames(cam, params, packed)
        torch.cuda.synchronize()
        ys = [y for y in range(H) if (y // band) % N == r]
        for k in range(K):
            a, b = packed[k].cpu().numpy(), full[k][ys].cpu().numpy()
            assert np.array_equal(a[..., 3], b[..., 3])
            np.testing.assert_allclose(a[..., :3], b[..., :3], rtol=2e-5, atol=1e-7)


def test_streaming_scene_frames_of_one_launch(bm, orc, torch_cuda):
    """Frames of one launch on a scene that streams: none of them sees a brick the launch itself requested (servicing happens between
    launches, bm_scene_process_load_queue), so from empty residency the launch's frames are the oracle's frames rendered back to back
    without servicing, and the request SET of the launch is the oracle's."""
    torch = torch_cuda
    G, W, H, K = 256, 96, 64, 3
    scene = bm.Scene(G, G, device=0)
    scene.set_queue_capacity(1 << 16)
    scene.generate()
    w = orc.World(G, G)
    w.set_queue_cap(1 << 16)
    w.reset_device(False)
    cams = [fly(bm, orc, 2 * k) for k in range(K)]
    params = [bm.FrameParams(W, H, spp=1, sample_base=k, max_bounces=3) for k in range(K)]
    accs = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0") for _ in range(K)]
    dbgs = [torch.zeros((H, W, 8), dtype=torch.int32, device="cuda:0") for _ in range(K)]
    scene.render_frames([c for c, _ in cams], params, accs, debugs=dbgs)
    torch.cuda.synchronize()
    for k in range(K):
        oacc, odbg, _, _ = w.render(cams[k][1], orc.make_frame(W, H, spp=1, max_bounces=3, sample_base=k))
        assert np.array_equal(dbgs[k].cpu().numpy().view(np.uint32), odbg)
        assert_radiance(accs[k].cpu().numpy(), oacc)
    for sc in range(w.nsc):  # requested bits after the launch = after the oracle's K frames
        assert np.array_equal(scene.device_indices(sc), w.sc_dev_indices(sc))
    assert scene.process_load_queue() == w.process_load_queue() > 0
    scene.close()


def test_bad_launches_are_refused(bm, orc, torch_cuda, scene256):
    torch = torch_cuda
    W, H = 64, 48
    cam, _ = cameras(bm, orc, 256)
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    p = bm.FrameParams(W, H, spp=1, max_bounces=3)
    with pytest.raises(bm.BrickmapError, match="1 ... 256"):
        scene256.render_frames(cam, [p] * 257, acc)
    with pytest.raises(bm.BrickmapError, match="1 ... 256"):
        scene256.render_frames(cam, [], [])
    # what shapes the hand-out must agree
    for other in (bm.FrameParams(W, H, spp=2, max_bounces=3), bm.FrameParams(W, H, spp=1, max_bounces=2), bm.FrameParams(W, H, spp=1, max_bounces=3, flags=bm.BM_FLAG_ORDERED)):
        with pytest.raises(bm.BrickmapError, match="must agree"):
            scene256.render_frames(cam, [p, other], [acc, torch.zeros_like(acc)])
    # ordered frames overlap in time and write pixels back with plain stores: one buffer for two of them is refused
    po = bm.FrameParams(W, H, spp=1, max_bounces=3, flags=bm.BM_FLAG_ORDERED)
    with pytest.raises(bm.BrickmapError, match="buffers of their own"):
        scene256.render_frames(cam, [po, po], acc)
    dbg = torch.zeros((H, W, 8), dtype=torch.int32, device="cuda:0")
    with pytest.raises(bm.BrickmapError, match="of their own"):
        scene256.render_frames(cam, [p, p], [acc, torch.zeros_like(acc)], debugs=[dbg, dbg])
    scene256.render_frames(cam, [p, p], acc)  # production frames may share the buffer
    torch.cuda.synchronize()
    assert torch.all(acc[..., 3] == 2)
