"""Randomised parity soak (GPU box): many seeded views / suns / lenses / sample offsets / LoD thresholds on two worlds
(256^3 and the non-cubic 384 x 384 x 128), the HIP path against the oracle: hit records bit-exact, radiance within 1e-4,
and the production instantiation bit-identical to the instrumented one; the production default (helper lanes, float atomics)
equal up to summation order; the helper-lane frame's ray digest (BM_FLAG_RAY_DIGEST) equal to the oracle's; the first frame of a
uniform frame-ring launch equal to the ordered frame.  usage: python tools/soak_parity.py [trials=400] [seed=1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import brickmap_amd as bm, oracle as orc

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
worlds = []
for (gs, gh) in ((256, 256), (384, 128)):
    s = bm.Scene(gs, gh, device=0).generate().preload_all()
    w = orc.World(gs, gh, threads=os.cpu_count() or 1)
    w.reset_device(True)
    worlds.append((gs, gh, s, w))
bad = 0
max_err = 0.0
for t in range(trials):
    gs, gh, scene, world = worlds[t % 2]
    lod8, lod2 = (600000, 100000) if t % 3 == 0 else (int(rng.integers(20, 900)), int(rng.integers(4, 200)))
    if lod2 > lod8:
        lod8, lod2 = lod2, lod8
    scene.set_lod(lod8, lod2); world.set_lod(lod8, lod2)
    pos = (float(rng.uniform(-0.3 * gs, 1.3 * gs)), float(rng.uniform(-0.3 * gs, 1.3 * gs)), float(rng.uniform(-0.3 * gh, 1.6 * gh)))
    if t % 7 == 0:
        pos = (float(rng.choice([0.0, gs])), float(rng.uniform(0, gs)), float(rng.choice([0.0, gh, gh / 2])))
    h, v = float(rng.uniform(-3.2, 3.2)), float(rng.uniform(-1.55, 1.55))
    sun = (float(rng.uniform(0, 1)), float(rng.uniform(0.02, 0.45)))
    lens = float(rng.choice([0.0, 0.0, 0.0, 0.6]))
    W, H = int(rng.integers(16, 120)), int(rng.integers(8, 72))
    spp, mb, sb = int(rng.integers(1, 4)), int(rng.integers(0, 8)), int(rng.integers(0, 100))
    cam = bm.Camera(position=pos, horizontal_angle=h, vertical_angle=v, lensRadius=lens, focalDistance=float(rng.uniform(0.5, 3))).update()
    ocam = orc.make_camera(cam.position, cam.direction, focal_distance=cam.focalDistance, lens_radius=lens)
    p = bm.FrameParams(W, H, spp=spp, sample_base=sb, max_bounces=mb, sun_position=sun)
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    plain = torch.zeros_like(acc)
    dbg = torch.zeros((H, W, 8), dtype=torch.int32, device="cuda:0")
    scene.render(cam, p, acc, debug=dbg)
    p_ord = bm.FrameParams(W, H, spp=spp, sample_base=sb, max_bounces=mb, sun_position=sun, flags=bm.BM_FLAG_ORDERED)
    scene.render(cam, p_ord, plain)                    # production instantiation, ordered sums: the instrumented frame's bits
    prod = torch.zeros_like(acc)
    scene.render(cam, p, prod)                         # production default: helper lanes, (chunk, sample) items for spp >= 2, float atomics
    dig = torch.zeros_like(dbg)
    pd = bm.FrameParams(W, H, spp=spp, sample_base=sb, max_bounces=mb, sun_position=sun, flags=bm.BM_FLAG_RAY_DIGEST)
    scene.render(cam, pd, torch.zeros_like(acc), debug=dig)   # the helper-lane instantiation's own hit records: the order-independent ray digest
    ring = torch.zeros((3,) + tuple(acc.shape), dtype=torch.float32, device="cuda:0")
    pr = [bm.FrameParams(W, H, spp=spp, sample_base=sb + k * spp, max_bounces=mb, sun_position=sun, flags=bm.BM_FLAG_ORDERED) for k in range(3)]
    scene.render_frames(cam, pr, [ring[k] for k in range(3)])  # a uniform frame-ring launch: its first frame is the ordered frame, bit for bit
    rdig, racc = torch.zeros_like(dbg), torch.zeros_like(acc)
    if spp * 3 * (mb + 1) < 65536:
        pg = [bm.FrameParams(W, H, spp=spp, sample_base=sb + k * spp, max_bounces=mb, sun_position=sun, flags=bm.BM_FLAG_RAY_DIGEST) for k in range(3)]
        scene.render_frames(cam, pg, racc, debugs=[rdig] * 3)  # production frames of a uniform launch, ONE digest buffer: the oracle's digest of the 3 x spp frame
    torch.cuda.synchronize()
    oacc, odbg, _, _ = world.render(ocam, orc.make_frame(W, H, spp=spp, max_bounces=mb, sample_base=sb, sun=sun), threads=os.cpu_count() or 1)
    a, b, d = acc.cpu().numpy(), plain.cpu().numpy(), dbg.cpu().numpy().view(np.uint32)
    ok = np.array_equal(d, odbg) and np.array_equal(a.view(np.uint32), b.view(np.uint32))
    c = prod.cpu().numpy()
    both = np.isfinite(a) & np.isfinite(c)
    ok = ok and np.array_equal(np.isfinite(a), np.isfinite(c)) and np.array_equal(np.where(both, c, 0)[..., 3], np.where(both, a, 0)[..., 3]) \
        and np.allclose(np.where(both, c, 0)[..., :3], np.where(both, a, 0)[..., :3], rtol=2e-5, atol=1e-7)
    ok = ok and np.array_equal(dig.cpu().numpy().view(np.uint32), world.last_ray_digest) and np.array_equal(ring[0].cpu().numpy().view(np.uint32), a.view(np.uint32))
    world.render(ocam, orc.make_frame(W, H, spp=3 * spp, max_bounces=mb, sample_base=sb, sun=sun), threads=os.cpu_count() or 1)
    ok = ok and np.array_equal(rdig.cpu().numpy().view(np.uint32), world.last_ray_digest)
    fin = np.isfinite(oacc)
    ok = ok and np.array_equal(np.isfinite(a), fin)
    err = float((np.abs(np.where(fin, a, 0) - np.where(fin, oacc, 0)) / np.maximum(np.abs(np.where(fin, oacc, 0)), 1e-6)).max())
    max_err = max(max_err, err)
    if not ok or err > 1e-4:
        bad += 1
        print(f"MISMATCH trial {t}: world {gs}x{gh} lod {lod8}/{lod2} pos {pos} h {h} v {v} sun {sun} lens {lens} {W}x{H} spp {spp} mb {mb} sb {sb}: records {np.array_equal(d, odbg)} plain==dbg {np.array_equal(a.view(np.uint32), b.view(np.uint32))} err {err:.2e}")
print(f"soak: {trials} trials, {bad} mismatches, max radiance error {max_err:.2e}")
sys.exit(1 if bad else 0)
