"""ctypes doorway to the CPU oracle (oracle/oracle.c) -- TEST INFRASTRUCTURE ONLY.

Allowed importers: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under brickmap_amd/ imports this package (tests/test_layout.py enforces it).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")
REF_LIB_PATH = os.path.join(_HERE, "_ref", "libref_simplex.so")
REFERENCE_ROOT = "/root/reference"


def build(force=False):
    """Compile oracle.c (and, where /root/reference exists, the real-reference noise lib)."""
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    if os.path.isdir(os.path.join(REFERENCE_ROOT, "src")):
        if force or not os.path.exists(REF_LIB_PATH):
            subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


class Camera(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("direction", C.c_float * 3), ("up", C.c_float * 3),
                ("focal_distance", C.c_float), ("lens_radius", C.c_float)]


class Frame(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("spp", C.c_int), ("sample_base", C.c_int),
                ("max_bounces", C.c_int), ("base_frame", C.c_uint), ("primary_only", C.c_int),
                ("band_rows", C.c_int), ("shard_rank", C.c_int), ("shard_count", C.c_int),
                ("sun_x", C.c_float), ("sun_y", C.c_float)]


COUNTER_NAMES = ("index_loads", "brick_tests", "byte_tests", "voxel_steps", "extend_rays",
                 "shadow_rays", "requests", "paths")


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in COUNTER_NAMES]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n in COUNTER_NAMES}


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        vp, f, i, u = C.c_void_p, C.c_float, C.c_int, C.c_uint
        fp, ip, up_ = C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_uint)
        sig = {
            "orc_sincos": (None, [i, vp, vp, vp]),
            "orc_noise2": (f, [f, f]),
            "orc_fractal2": (f, [i, f, f]),
            "orc_fractal2_grid": (None, [i, i, vp, vp, vp]),
            "orc_world_create": (vp, [i, i]),
            "orc_world_destroy": (None, [vp]),
            "orc_world_set_lod": (None, [vp, i, i]),
            "orc_world_set_queue_cap": (None, [vp, i]),
            "orc_column_heights": (None, [vp, i, i, vp]),
            "orc_world_reset_device": (None, [vp, i]),
            "orc_world_generate": (None, [vp, i]),
            "orc_world_nsc": (i, [vp]),
            "orc_world_sc_nbricks": (C.c_uint32, [vp, i]),
            "orc_world_sc_indices": (vp, [vp, i]),
            "orc_world_sc_bricks": (vp, [vp, i]),
            "orc_world_sc_dev_indices": (vp, [vp, i]),
            "orc_world_sc_gpu_count": (i, [vp, i]),
            "orc_world_sc_gpu_index_highest": (i, [vp, i]),
            "orc_world_total_bricks": (C.c_uint64, [vp]),
            "orc_world_queue_count": (C.c_uint32, [vp]),
            "orc_world_total_uploaded": (C.c_uint64, [vp]),
            "orc_world_hash": (C.c_uint64, [vp]),
            "orc_upload": (C.c_uint32, [vp]),
            "orc_world_set_overlapped": (None, [vp, i]),
            "orc_process_load_queue_overlapped": (C.c_uint32, [vp]),
            "orc_process_load_queue": (C.c_uint32, [vp]),
            "orc_intersect_brick": (i, [vp, vp, vp, vp, vp, vp]),
            "orc_intersect_byte": (i, [vp, vp, vp, vp, C.c_uint32, vp]),
            "orc_intersect_voxel": (i, [vp, vp, vp, vp, vp, vp, vp, vp]),
            "orc_rng_stream": (None, [u, i, vp]),
            "orc_rng_floats": (None, [u, i, vp, vp]),
            "orc_stratified": (None, [u, vp, vp]),
            "orc_sky_probe": (None, [f, f, vp, vp, vp, vp, vp]),
            "orc_cone_sample": (None, [f, f, u, vp, vp]),
            "orc_camera_direction": (None, [C.c_double, C.c_double, vp]),
            "orc_set_ray_digest_buffer": (None, [vp]),
            "orc_render": (C.c_double, [vp, C.POINTER(Camera), C.POINTER(Frame), vp, vp, C.POINTER(Counters), i]),
            "orc_wavefront_create": (vp, [u, i]),
            "orc_wavefront_destroy": (None, [vp]),
            "orc_wavefront_reset": (None, [vp]),
            "orc_wavefront_stats": (None, [vp, vp]),
            "orc_wavefront_counters": (None, [vp, C.POINTER(Counters)]),
            "orc_wavefront_read_queue": (C.c_int, [vp, C.c_int, C.c_uint, C.c_uint, vp]),
            "orc_wavefront_frame": (None, [vp, vp, C.POINTER(Camera), i, i, f, f, vp]),
            "orc_resolve": (None, [vp, vp, i]),
            "orc_sizeof_counters": (i, []),
            "orc_sizeof_rayqueue": (i, []),
            "orc_sizeof_shadowqueue": (i, []),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        assert L.orc_sizeof_counters() == C.sizeof(Counters)
        _lib = L
    return _lib


def ref_lib():
    """The real reference SimplexNoise (oracle/_ref), or None when it was never built."""
    if not os.path.exists(REF_LIB_PATH):
        return None
    L = C.CDLL(REF_LIB_PATH)
    L.ref_fractal2.restype = C.c_float
    L.ref_fractal2.argtypes = [C.c_int, C.c_float, C.c_float]
    L.ref_noise2.restype = C.c_float
    L.ref_noise2.argtypes = [C.c_float, C.c_float]
    L.ref_fractal2_grid.restype = None
    L.ref_fractal2_grid.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def make_camera(position, direction, up=(0, 0, 1), focal_distance=1.0, lens_radius=0.0):
    cam = Camera()
    cam.position[:] = [float(v) for v in position]
    cam.direction[:] = [float(v) for v in direction]
    cam.up[:] = [float(v) for v in up]
    cam.focal_distance = focal_distance
    cam.lens_radius = lens_radius
    return cam


def camera_direction(h, v):
    out = np.zeros(3, np.float32)
    lib().orc_camera_direction(h, v, _ptr(out))
    return out


def make_frame(width, height, spp=1, max_bounces=3, sample_base=0, base_frame=1, primary_only=0,
               band_rows=0, shard_rank=0, shard_count=1, sun=(0.05, 0.1), ray_digest=False):
    """ray_digest: words 4-7 of the hit records are the order-independent per-RAY sums of the product's BM_FLAG_RAY_DIGEST frames
    (oracle.c render_pixel) instead of hash chains in path order."""
    return Frame(width, height, spp, sample_base, max_bounces, base_frame, (1 if primary_only else 0) | (2 if ray_digest else 0),
                 band_rows if band_rows > 0 else height, shard_rank, shard_count, sun[0], sun[1])


class World:
    def __init__(self, grid_size, grid_height, threads=None, generate=True):
        self.L = lib()
        self.grid_size, self.grid_height = grid_size, grid_height
        self.h = self.L.orc_world_create(grid_size, grid_height)
        if not self.h:
            raise ValueError("world dims must be positive multiples of 128")
        if generate:
            self.L.orc_world_generate(self.h, threads or os.cpu_count() or 1)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_world_destroy(self.h)
            self.h = None

    @property
    def nsc(self):
        return self.L.orc_world_nsc(self.h)

    def reset_device(self, preload_all):
        self.L.orc_world_reset_device(self.h, int(preload_all))

    def set_lod(self, lod8, lod2):
        self.L.orc_world_set_lod(self.h, lod8, lod2)

    def set_queue_cap(self, cap):
        self.L.orc_world_set_queue_cap(self.h, cap)

    def sc_nbricks(self, sc):
        return int(self.L.orc_world_sc_nbricks(self.h, sc))

    def sc_indices(self, sc):
        p = self.L.orc_world_sc_indices(self.h, sc)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(4096,)).copy()

    def sc_dev_indices(self, sc):
        p = self.L.orc_world_sc_dev_indices(self.h, sc)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(4096,)).copy()

    def sc_bricks(self, sc):
        n = self.sc_nbricks(sc)
        if n == 0:
            return np.zeros((0, 16), np.uint32)
        p = self.L.orc_world_sc_bricks(self.h, sc)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(n, 16)).copy()

    def total_bricks(self):
        return int(self.L.orc_world_total_bricks(self.h))

    def hash(self):
        return int(self.L.orc_world_hash(self.h))

    def column_heights(self, sx, sy):
        out = np.zeros((128, 128), np.float32)
        self.L.orc_column_heights(self.h, sx, sy, _ptr(out))
        return out

    def queue_count(self):
        return int(self.L.orc_world_queue_count(self.h))

    def process_load_queue(self):
        return int(self.L.orc_process_load_queue(self.h))

    def upload(self):
        return int(self.L.orc_upload(self.h))

    def set_overlapped(self, on):
        """Two-ring servicing with the reference's two-frame request -> resident latency (orc_process_load_queue_overlapped);
        Wavefront.frame() then uses it instead of the reference-order upload / process_load_queue pair."""
        self.L.orc_world_set_overlapped(self.h, int(on))

    def process_load_queue_overlapped(self):
        return int(self.L.orc_process_load_queue_overlapped(self.h))

    def intersect_voxel(self, origin, direction, campos, normal=(0, 0, 0), distance=1e20):
        o = np.asarray(origin, np.float32)
        d = np.asarray(direction, np.float32)
        n = np.asarray(normal, np.float32).copy()
        dist = np.asarray([distance], np.float32)
        cp = np.asarray(campos, np.int32)
        out4 = np.zeros(4, np.int32)
        loads = C.c_uint64(0)
        r = self.L.orc_intersect_voxel(self.h, _ptr(o), _ptr(d), _ptr(n), _ptr(dist), _ptr(cp), _ptr(out4), C.byref(loads))
        return dict(ret=r, normal=n, distance=float(dist[0]), hit=int(out4[0]), level=int(out4[1]),
                    brick_id=int(out4[2]), sub_id=int(out4[3]), index_loads=int(loads.value))

    def render(self, cam, frame, accum=None, want_dbg=True, threads=1):
        """Canonical per-pixel render (mode B). Returns (accum[H,W,4], dbg[H,W,8] | None, counters dict, seconds).
        With want_dbg the same pass also fills self.last_ray_digest: the hit records with words 4 / 5 replaced by the order-independent
        per-RAY sums (what the product's BM_FLAG_RAY_DIGEST frames hold; oracle.c render_pixel)."""
        W, H = frame.width, frame.height
        if accum is None:
            accum = np.zeros((H, W, 4), np.float32)
        dbg = np.zeros((H, W, 8), np.uint32) if want_dbg else None
        sums = np.zeros((H, W, 2), np.uint32) if want_dbg else None
        cnt = Counters()
        self.L.orc_set_ray_digest_buffer(_ptr(sums) if sums is not None else None)
        try:
            secs = self.L.orc_render(self.h, C.byref(cam), C.byref(frame), _ptr(accum),
                                     _ptr(dbg) if dbg is not None else None, C.byref(cnt), threads)
        finally:
            self.L.orc_set_ray_digest_buffer(None)
        self.last_ray_digest = None
        if dbg is not None:
            self.last_ray_digest = dbg.copy()
            if not (frame.primary_only & 2):
                self.last_ray_digest[..., 4:6] = sums
        return accum, dbg, cnt.as_dict(), secs


class Wavefront:
    """Mode A: the reference's wavefront schedule run sequentially."""

    def __init__(self, queue_size=2 * 1048576, max_bounces=3):
        self.L = lib()
        self.h = self.L.orc_wavefront_create(queue_size, max_bounces)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_wavefront_destroy(self.h)
            self.h = None

    def reset(self):
        self.L.orc_wavefront_reset(self.h)

    def frame(self, world, cam, W, H, accum, sun=(0.05, 0.1)):
        self.L.orc_wavefront_frame(self.h, world.h, C.byref(cam), W, H, sun[0], sun[1], _ptr(accum))
        out = np.zeros(6, np.uint32)
        self.L.orc_wavefront_stats(self.h, _ptr(out))
        return dict(survivors=int(out[0]), shadow=int(out[1]), start_position=int(out[2]), frame=int(out[3]),
                    generated=int(out[4]), primary_ray_cnt=int(out[5]))

    def read_queue(self, which, first, count):
        """raw bytes of `count` records of the work (which=0, 64 B each) or shadow (which=1, 40 B each) queue"""
        out = np.zeros(count * (64 if which == 0 else 40), np.uint8)
        rc = self.L.orc_wavefront_read_queue(self.h, which, first, count, _ptr(out))
        assert rc == 0
        return out

    def counters(self):
        cnt = Counters()
        self.L.orc_wavefront_counters(self.h, C.byref(cnt))
        return cnt.as_dict()


def sincos(x):
    x = np.ascontiguousarray(x, np.float32)
    s = np.zeros_like(x)
    c = np.zeros_like(x)
    lib().orc_sincos(x.size, _ptr(x), _ptr(s), _ptr(c))
    return s, c


def fractal2_grid(octaves, xs, ys):
    xs = np.ascontiguousarray(xs, np.float32)
    ys = np.ascontiguousarray(ys, np.float32)
    out = np.zeros_like(xs)
    lib().orc_fractal2_grid(octaves, xs.size, _ptr(xs), _ptr(ys), _ptr(out))
    return out


def sky_probe(viewdir, sun=(0.05, 0.1)):
    v = np.asarray(viewdir, np.float32)
    sd, a, b, c = (np.zeros(3, np.float32) for _ in range(4))
    lib().orc_sky_probe(sun[0], sun[1], _ptr(v), _ptr(sd), _ptr(a), _ptr(b), _ptr(c))
    return dict(sun_direction=sd, sun=a, sky=b, sunsky=c)


def cone_sample(seed, sun=(0.05, 0.1)):
    out = np.zeros(3, np.float32)
    after = C.c_uint(0)
    lib().orc_cone_sample(sun[0], sun[1], seed, _ptr(out), C.byref(after))
    return out, int(after.value)


def rng_stream(seed, n):
    out = np.zeros(n, np.uint32)
    lib().orc_rng_stream(seed, n, _ptr(out))
    return out


def rng_floats(seed, n):
    a = np.zeros(n, np.float32)
    b = np.zeros(n, np.float32)
    lib().orc_rng_floats(seed, n, _ptr(a), _ptr(b))
    return a, b
