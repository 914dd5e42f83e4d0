"""Parity of the HIP wavefront mode (bm_wavefront_*: the reference's own queue schedule, kernel.cu:366-439) with
oracle mode A (the reference's kernels run sequentially), through the C-ABI.  Queues, counters and statistics are
integer / bit-exact; the frame buffer is accumulated with float atomics and agrees within RTOL."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
RTOL = 1e-4


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def assert_radiance(got, want):
    if want.size == 0:
        return
    scale = np.maximum(np.abs(want), 1e-6)
    err = np.abs(got - want) / scale
    assert float(err.max()) <= RTOL, f"max relative radiance error {err.max():.3e}"


def compare_frame(wf, owf, st, ost):
    assert st == ost
    n, m = ost["survivors"], ost["shadow"]
    got = wf.read_queue("work", 0, n).view(np.uint8)
    want = owf.read_queue(0, 0, n)
    assert np.array_equal(got, want), "next-frame work queue differs"
    # shadow queue: geometry and pixel bit-exact; `color` is radiance (sky model: fp32 transcendentals) -> RTOL
    got = wf.read_queue("shadow", 0, m)
    want = owf.read_queue(1, 0, m).view(got.dtype)
    for field in ("origin", "direction", "pixel_index"):
        assert np.array_equal(got[field].view(np.uint32), want[field].view(np.uint32)), f"shadow queue {field} differs"
    assert_radiance(got["color"], want["color"])


@pytest.mark.parametrize("queue_size", [5000, 8192])
def test_wavefront_frames_match_oracle_mode_a(queue_size, bm, orc, torch_cuda):
    """Six consecutive launch_kernels calls (queue smaller / larger than the frame), then a camera move with the
    reference's reset; after every call the survivor queue, the shadow queue and the globals are bit-identical."""
    torch = torch_cuda
    G, W, H = 256, 96, 64
    scene = bm.Scene(G, G, device=0).generate()
    scene.preload_all()
    w = orc.World(G, G)
    w.reset_device(True)
    wf, owf = bm.Wavefront(scene, queue_size), orc.Wavefront(queue_size=queue_size, max_bounces=3)
    p = bm.FrameParams(W, H, max_bounces=3, flags=bm.BM_FLAG_COUNTERS)
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    oacc = np.zeros((H, W, 4), np.float32)
    cam = bm.Camera(position=(G / 2, G / 8, 0.8 * G), horizontal_angle=0.8, vertical_angle=-0.5).update()
    ocam = orc.make_camera(cam.position, cam.direction)
    for _ in range(6):
        wf.frame(cam, p, acc)
        ost = owf.frame(w, ocam, W, H, oacc)
        compare_frame(wf, owf, wf.stats(), ost)
    assert_radiance(acc.cpu().numpy(), oacc)
    # camera change: reset_buffer branch (kernel.cu:387-403)
    cam = bm.Camera(position=(40.0, 200.0, 150.0), horizontal_angle=2.1, vertical_angle=-0.3).update()
    ocam = orc.make_camera(cam.position, cam.direction)
    wf.reset(); owf.reset()
    acc.zero_(); oacc[:] = 0
    for _ in range(5):
        wf.frame(cam, p, acc)
        ost = owf.frame(w, ocam, W, H, oacc)
        compare_frame(wf, owf, wf.stats(), ost)
    assert_radiance(acc.cpu().numpy(), oacc)
    assert wf.counters() == owf.counters()
    ce, cc = wf.counters("extend"), wf.counters("connect")
    assert ce["shadow_rays"] == 0 and cc["extend_rays"] == 0 and ce["extend_rays"] == 11 * queue_size
    t = wf.times()
    assert t["total"] > 0 and t["extend"] > 0
    wf.close(); scene.close()


def test_wavefront_streaming_matches_oracle(bm, orc, torch_cuda):
    """Reference residency (nothing loaded), the main loop of main.cpp:142-146: launch_kernels, process_load_queue,
    swap.  Requests, uploads and the queues agree frame by frame while the scene streams in."""
    torch = torch_cuda
    G, W, H, Q = 256, 96, 64, 6144
    scene = bm.Scene(G, G, device=0)
    scene.set_queue_capacity(1 << 16)
    scene.generate()
    w = orc.World(G, G)
    w.set_queue_cap(1 << 16)
    w.reset_device(False)
    wf, owf = bm.Wavefront(scene, Q), orc.Wavefront(queue_size=Q, max_bounces=3)
    p = bm.FrameParams(W, H, max_bounces=3, flags=bm.BM_FLAG_COUNTERS)
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    oacc = np.zeros((H, W, 4), np.float32)
    cam = bm.Camera(position=(G / 2, G / 8, 0.8 * G), horizontal_angle=0.8, vertical_angle=-0.5).update()
    ocam = orc.make_camera(cam.position, cam.direction)
    serviced = 0
    for _ in range(8):
        wf.frame(cam, p, acc)
        serviced += scene.process_load_queue()
        ost = owf.frame(w, ocam, W, H, oacc)  # includes the oracle's process_load_queue
        compare_frame(wf, owf, wf.stats(), ost)
    assert_radiance(acc.cpu().numpy(), oacc)
    cnt, ocnt = wf.counters(), owf.counters()
    assert cnt == ocnt and ocnt["requests"] > 0 and ocnt["brick_tests"] > 0
    assert serviced == ocnt["requests"] == scene.info()["resident_bricks"]
    wf.close(); scene.close()


def test_wavefront_overlapped_streaming_matches_oracle(bm, orc, torch_cuda):
    """The same loop with the two-ring servicing on both sides (orc_process_load_queue_overlapped): requests become
    resident two calls after they were raised; queues, statistics and uploads agree frame by frame."""
    torch = torch_cuda
    G, W, H, Q = 256, 96, 64, 6144
    scene = bm.Scene(G, G, device=0)
    scene.set_queue_capacity(1 << 16)
    scene.generate()
    scene.set_streaming_mode(True)
    w = orc.World(G, G)
    w.set_queue_cap(1 << 16)
    w.reset_device(False)
    w.set_overlapped(True)
    wf, owf = bm.Wavefront(scene, Q), orc.Wavefront(queue_size=Q, max_bounces=3)
    p = bm.FrameParams(W, H, max_bounces=3, flags=bm.BM_FLAG_COUNTERS)
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    oacc = np.zeros((H, W, 4), np.float32)
    cam = bm.Camera(position=(G / 2, G / 8, 0.8 * G), horizontal_angle=0.8, vertical_angle=-0.5).update()
    ocam = orc.make_camera(cam.position, cam.direction)
    serviced = []
    for _ in range(10):
        wf.frame(cam, p, acc)
        serviced.append(scene.process_load_queue())
        ost = owf.frame(w, ocam, W, H, oacc)  # includes the oracle's overlapped servicing
        compare_frame(wf, owf, wf.stats(), ost)
    assert_radiance(acc.cpu().numpy(), oacc)
    cnt, ocnt = wf.counters(), owf.counters()
    assert cnt == ocnt and ocnt["requests"] > 0 and ocnt["brick_tests"] > 0
    assert serviced[0] == 0 and serviced[1] > 0
    scene.process_load_queue(); scene.process_load_queue()  # drain what the last two frames asked for
    assert ocnt["requests"] == scene.info()["resident_bricks"]
    wf.close(); scene.close()


def test_wavefront_frame1_equals_the_reference_run(bm, torch_cuda):
    """The numbers the reference's own kernels produced for frame 1 at 1080p on its default 4096x4096x512 world
    (SURVEY.md 8c probe; tests/golden/survey_probes.json): survivors, shadow rays, start_position."""
    torch = torch_cuda
    p = json.load(open(os.path.join(GOLDEN, "survey_probes.json")))["wavefront_frame1"]
    scene = bm.Scene(4096, 512, device=0).generate()
    scene.preload_all()
    wf = bm.Wavefront(scene, p["queue_size"])
    cam = bm.Camera(position=tuple(p["camera_position"]), horizontal_angle=p["camera_angles"][0], vertical_angle=p["camera_angles"][1]).update()
    acc = torch.zeros((p["height"], p["width"], 4), dtype=torch.float32, device="cuda:0")
    wf.frame(cam, bm.FrameParams(p["width"], p["height"], max_bounces=3), acc)
    st = wf.stats()
    assert (st["survivors"], st["shadow"], st["start_position"]) == (p["survivors"], p["shadow"], p["start_position"])
    assert st["generated"] == p["queue_size"]
    # every slot produced exactly one of {terminated path, survivor}: alpha sums to the misses of this frame
    alpha = float(acc[..., 3].sum().item())
    assert alpha == p["queue_size"] - st["survivors"]
    # a few more frames: the queue stays full, paths retire, alpha keeps counting terminated paths
    for _ in range(4):
        wf.frame(cam, bm.FrameParams(p["width"], p["height"], max_bounces=3), acc)
    st = wf.stats()
    assert st["frame"] == 6 and 0 < st["survivors"] < p["queue_size"]
    a = acc.cpu().numpy()
    assert np.isfinite(a).all() and a[..., 3].min() >= 0
    wf.close(); scene.close()


def test_wavefront_and_fused_schedules_estimate_the_same_image(bm, torch_cuda):
    """launch_kernels(queues=Wavefront) is the reference's schedule, launch_kernels() the fused per-pixel one: different
    RNG streams, same estimator -- after ~100 paths per pixel the two images agree statistically."""
    torch = torch_cuda
    G, W, H = 256, 64, 48
    scene = bm.Scene(G, G, device=0).generate()
    scene.preload_all()
    cam = bm.Camera(position=(G / 2, G / 8, 0.8 * G), horizontal_angle=0.8, vertical_angle=-0.5).update()
    fused, queued = bm.State(W, H), bm.State(W, H)
    from brickmap_amd.host import _LaunchStatics
    s1, s2 = _LaunchStatics(), _LaunchStatics()
    bm.launch_kernels(fused, fused.blit_buffer, scene, cam, spp=128, statics=s1)
    wf = bm.Wavefront(scene, W * H)
    for _ in range(400):
        assert bm.launch_kernels(queued, queued.blit_buffer, scene, cam, statics=s2, queues=wf) == 0
    a, b = fused.blit_buffer.cpu().numpy(), queued.blit_buffer.cpu().numpy()
    assert a[..., 3].min() == 128 and b[..., 3].min() > 60  # terminated paths per pixel
    ia, ib = a[..., :3] / a[..., 3:], b[..., :3] / b[..., 3:]
    assert abs(ia.mean() - ib.mean()) < 0.02 * ia.mean()
    assert np.corrcoef(ia.ravel(), ib.ravel())[0, 1] > 0.95
    # a camera move resets both the frame buffer and the queues (kernel.cu:387-403)
    cam2 = bm.Camera(position=(40.0, 200.0, 150.0), horizontal_angle=2.1, vertical_angle=-0.3).update()
    bm.launch_kernels(queued, queued.blit_buffer, scene, cam2, statics=s2, queues=wf)
    st = wf.stats()
    assert st["generated"] == W * H  # the whole queue was refilled with primary rays
    assert float(queued.blit_buffer[..., 3].sum().item()) == W * H - st["survivors"]
    wf.close(); scene.close()


def test_cpp_headless_example_wavefront_schedule(bm, torch_cuda, tmp_path):
    """include/brickmap.hpp: launch_kernels(state, blit, gpuScene, queues) in the reference's main loop, streaming."""
    import subprocess
    from conftest import ROOT
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples")])
    out = tmp_path / "frame.ppm"
    r = subprocess.run([os.path.join(ROOT, "examples", "headless_main"), "256", "256", "160", "96", "48", str(out), "wavefront"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    data = out.read_bytes()
    head = b"P6\n160 96\n255\n"
    assert data.startswith(head) and len(data) == len(head) + 160 * 96 * 3
    px = np.frombuffer(data[len(head):], np.uint8)
    assert px.max() > 0 and len(np.unique(px)) > 16


def test_wavefront_full_size_config2_queues_match_oracle(bm, orc, torch_cuda):
    """BASELINE config 2 scale (1080p, the reference's 2 Mi-slot queue, 1024^3 world): three consecutive launch_kernels
    calls; every survivor and shadow record (~1.7 M each per call) is compared with oracle mode A."""
    torch = torch_cuda
    G, W, H, Q = 1024, 1920, 1080, 2 * 1048576
    scene = bm.Scene(G, G, device=0).generate()
    scene.preload_all()
    w = orc.World(G, G, threads=os.cpu_count() or 1)
    w.reset_device(True)
    wf, owf = bm.Wavefront(scene, Q), orc.Wavefront(queue_size=Q, max_bounces=3)
    p = bm.FrameParams(W, H, max_bounces=3)
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    oacc = np.zeros((H, W, 4), np.float32)
    cam = bm.Camera(position=(G / 2, G / 8, 0.8 * G), horizontal_angle=0.8, vertical_angle=-0.5).update()
    ocam = orc.make_camera(cam.position, cam.direction)
    for _ in range(3):
        wf.frame(cam, p, acc)
        ost = owf.frame(w, ocam, W, H, oacc)
        compare_frame(wf, owf, wf.stats(), ost)
    assert_radiance(acc.cpu().numpy(), oacc)
    wf.close(); scene.close()


@pytest.mark.parametrize("W,H,Q,mb", [(1, 1, 1, 3), (7, 3, 5, 0), (33, 17, 257, 1), (64, 48, 4096, 7)])
def test_wavefront_edge_shapes(W, H, Q, mb, bm, orc, torch_cuda):
    """Degenerate frames and queues (one slot, queue not a multiple of the workgroup, no bounces, long paths)."""
    torch = torch_cuda
    G = 128
    scene = bm.Scene(G, G, device=0).generate()
    scene.preload_all()
    w = orc.World(G, G)
    w.reset_device(True)
    wf, owf = bm.Wavefront(scene, Q), orc.Wavefront(queue_size=Q, max_bounces=mb)
    p = bm.FrameParams(W, H, max_bounces=mb)
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    oacc = np.zeros((H, W, 4), np.float32)
    cam = bm.Camera(position=(G / 2, G / 8, 0.8 * G), horizontal_angle=0.8, vertical_angle=-0.5).update()
    ocam = orc.make_camera(cam.position, cam.direction)
    for _ in range(2 * mb + 4):
        wf.frame(cam, p, acc)
        compare_frame(wf, owf, wf.stats(), owf.frame(w, ocam, W, H, oacc))
    assert_radiance(acc.cpu().numpy(), oacc)
    wf.close(); scene.close()


def test_wavefront_with_overlapped_streaming(bm, torch_cuda):
    """The queue schedule on a scene that streams in with the overlapped (two-ring, no host wait) request servicing:
    every brick is uploaded exactly once, and once nothing is requested any more a fresh run of the wavefront gives
    exactly the queues of the all-resident scene."""
    torch = torch_cuda
    G, W, H, Q = 256, 160, 120, 16384
    scene = bm.Scene(G, G, device=0)
    scene.set_queue_capacity(512)
    scene.generate()
    scene.set_streaming_mode(True)
    cam = bm.Camera(position=(G / 2, G / 8, 0.8 * G), horizontal_angle=0.8, vertical_angle=-0.5).update()
    p = bm.FrameParams(W, H, max_bounces=3)
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    wf = bm.Wavefront(scene, Q)
    total, idle = 0, 0
    for _ in range(600):
        wf.frame(cam, p, acc)
        n = scene.process_load_queue()
        total += n
        idle = idle + 1 if n == 0 else 0
        if idle >= 12:  # several path generations without a single request
            break
    assert idle >= 12 and total > 0
    info = scene.info()
    assert total == info["resident_bricks"]
    loaded = sum(int(np.count_nonzero(scene.device_indices(sc) & bm.BRICK_LOADED_BIT)) for sc in range(8))
    requested = sum(int(np.count_nonzero(scene.device_indices(sc) & bm.BRICK_REQUESTED_BIT)) for sc in range(8))
    assert loaded == total and requested == 0

    def run(n):
        w2 = bm.Wavefront(scene, Q)
        a = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
        out = []
        for _ in range(n):
            w2.frame(cam, p, a)
            st = w2.stats()
            out.append((st, w2.read_queue("work", 0, st["survivors"]).tobytes(), w2.read_queue("shadow", 0, st["shadow"])["origin"].tobytes()))
        w2.close()
        return out, a.cpu().numpy()

    # a fresh wavefront restarts the RNG streams, so its paths may find bricks the long run never touched: repeat the
    # same six calls until they request nothing (each repeat sees the bricks the previous one asked for)
    for _ in range(16):
        streamed, a_s = run(6)
        if scene.process_load_queue() + scene.process_load_queue() == 0:
            break
    else:
        pytest.fail("the six-call sequence keeps requesting bricks")
    scene.set_streaming_mode(False)
    scene.preload_all()
    resident, a_r = run(6)
    assert streamed == resident
    assert_radiance(a_s, a_r)
    wf.close(); scene.close()
