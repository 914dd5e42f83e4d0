"""Host-side mirror of the reference's Scene / Camera / State / launch_kernels interface.

Same names, argument meaning and call order as the reference's C++ (file:line into
the reference checkout, src/): `Scene` (Scene.h:7-44), `Camera` (camera.h:3-24), `State` (state.h:5-34),
`launch_kernels` (launch.h:6, kernel.cu:366-439).  Everything below the method bodies is the
C-ABI of libbrickmap_hip.so; torch only provides device tensors and streams.
"""
import ctypes as C
import math
import os
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from ._lib import bm_camera, bm_counters, bm_frame_params, bm_scene_info, check


# The reference's fly-through presets (performance_measure.h:4-25): camera position + (horizontal, vertical) angle.
# It lists 9 positions but only 8 angle pairs and indexes both with the same counter (performance_measure.cpp:32-34).
# World: the reference's native 4096 x 4096 x 512 voxels.
FLYTHROUGH_VIEWS = (
    ((512.0, 512.0, 300.0), (-61863.5, -0.501796)),
    ((840.254, 832.446, 1169.88), (-61864.4, -0.429796)),
    ((2227.83, 774.886, 204.955), (-61863.9, 0.0622036)),
    ((3326.19, 2055.72, 44.7995), (-61864.2, -0.981796)),
    ((7134.6, 1262.44, 5531.79), (-61865.2, -0.501796)),
    ((11298.6, 3113.03, 598.019), (-61866.3, -0.141796)),
    ((10921.4, 4774.14, 267.808), (-61859.4, 0.0142036)),
    ((9961.29, 4508.12, 189.59), (-61857.2, -0.261796)),
    # the 9th position has no angle pair of its own: the reference reads test_angles[8] past the end of the vector
    # (performance_measure.cpp:33-34, undefined behaviour); here it is flown with the last defined pair
    ((10835.3, 4160.83, 359.992), (-61857.2, -0.261796)),
)


def flythrough_camera(i):
    """Camera of the reference's i-th fly-through viewpoint."""
    pos, (h, v) = FLYTHROUGH_VIEWS[i % len(FLYTHROUGH_VIEWS)]
    return Camera(position=pos, horizontal_angle=h, vertical_angle=v).update()


def _f32(v):
    return np.asarray(v, dtype=np.float32)


def _normalize_f32(v):
    """glm::normalize in fp32: v * (1 / sqrt((x*x + y*y) + z*z))."""
    v = _f32(v)
    d = np.float32(np.float32(v[0] * v[0] + v[1] * v[1]) + v[2] * v[2])
    return v * (np.float32(1.0) / np.sqrt(d, dtype=np.float32))


@dataclass
class Camera:
    """camera.h:3-24.  `handle_input` (GLFW keys/mouse) is out of scope: there is no window."""
    position: tuple = (512.0, 512.0, 300.0)
    direction: tuple = (1.0, 0.0, 0.0)
    up: tuple = (0.0, 0.0, 1.0)
    focalDistance: float = 1.0
    lensRadius: float = 0.0
    horizontal_angle: float = 0.0
    vertical_angle: float = 0.0

    def update(self):
        """Camera::update (camera.cpp:48-54): direction from the two angles, then normalised."""
        h, v = self.horizontal_angle, self.vertical_angle
        d = _f32([math.cos(v) * math.sin(h), math.cos(v) * math.cos(h), math.sin(v)])
        self.direction = tuple(float(x) for x in _normalize_f32(d))
        return self

    def to_c(self):
        c = bm_camera()
        c.position[:] = [float(x) for x in self.position]
        c.direction[:] = [float(x) for x in self.direction]
        c.up[:] = [float(x) for x in self.up]
        c.focal_distance = float(self.focalDistance)
        c.lens_radius = float(self.lensRadius)
        return c


@dataclass
class FrameParams:
    """Per-launch parameters the reference keeps as constexprs / statics (kernel.cu:13,369; variables.cpp:3)."""
    width: int
    height: int
    spp: int = 1
    sample_base: int = 0
    max_bounces: int = 3
    base_frame: int = 1
    flags: int = 0
    band_rows: int = 0  # 0 = whole image in one band
    shard_rank: int = 0
    shard_count: int = 1
    sun_position: tuple = (0.05, 0.1)

    def to_c(self):
        p = bm_frame_params()
        p.width, p.height, p.spp, p.sample_base = self.width, self.height, self.spp, self.sample_base
        p.max_bounces, p.base_frame, p.flags = self.max_bounces, self.base_frame, self.flags
        p.band_rows = self.band_rows if self.band_rows > 0 else self.height
        p.shard_rank, p.shard_count = self.shard_rank, self.shard_count
        p.sun_position[:] = [float(self.sun_position[0]), float(self.sun_position[1])]
        return p


def local_rows(params: FrameParams) -> int:
    """Rows of the frame owned by this shard (bm_local_rows)."""
    p = params.to_c()
    return int(_lib.load().bm_local_rows(C.byref(p)))


def frame_plan(params: FrameParams, hit_records=False):
    """bm_frame_plan_of: what the library decides for a frame with these parameters (host only): flags after its own choice of work
    items, ordered / helpers / sample_items / xcd_handout / refill_min / instrumented, tiles and local rows."""
    plan = _lib.bm_frame_plan()
    par_c = params.to_c()
    check(_lib.load().bm_frame_plan_of(C.byref(par_c), 1 if hit_records else 0, C.byref(plan)))
    return {name: int(getattr(plan, name)) for name, _ in _lib.bm_frame_plan._fields_}


def trace_waves_per_simd(instrumented=False, xcd_handout=False, helpers=True, device=0):
    """bm_trace_waves_per_simd: resident waves per SIMD of the trace_paths instantiation on `device`."""
    n = C.c_int(0)
    check(_lib.load().bm_trace_waves_per_simd(device, int(bool(instrumented)), int(bool(xcd_handout)), int(bool(helpers)), C.byref(n)))
    return int(n.value)


def tuning_overrides():
    """bm_tuning_overrides: {name: value} of the BM_* tuning variables this process runs under ({} = the product's own rules)."""
    buf = C.create_string_buffer(512)
    check(_lib.load().bm_tuning_overrides(buf, 512))
    return {k: int(v) for k, v in (item.split("=") for item in buf.value.decode().split())}


def host_cube_field(grid_size, grid_height):
    """The octant cube field of the generated world (bm_host_cube_field): uint8 [8, cells_height+2, cells+2, cells+2]."""
    L = _lib.load()
    n = C.c_size_t(0)
    check(L.bm_host_cube_field(grid_size, grid_height, None, 0, C.byref(n)))
    out = np.zeros(n.value, np.uint8)
    check(L.bm_host_cube_field(grid_size, grid_height, out.ctypes.data, out.size, C.byref(n)))
    cx, cz = grid_size // 8 + 2, grid_height // 8 + 2
    return out.reshape(8, cz, cx, cx)


def host_column_heights(grid_size, grid_height, sx, sy):
    """Terrain heights of one supercell column from the product's CPU generator (no device needed)."""
    out = np.zeros((128, 128), np.float32)
    check(_lib.load().bm_host_column_heights(grid_size, grid_height, sx, sy, out.ctypes.data))
    return out


def host_generate_supercell(grid_size, grid_height, sx, sy, sz):
    """(indices[4096], bricks[n,16]) of one supercell from the product's CPU generator (no device needed)."""
    L = _lib.load()
    idx = np.zeros(4096, np.uint32)
    n = C.c_uint32(0)
    bricks = np.zeros((4096, 16), np.uint32)
    check(L.bm_host_generate_supercell(grid_size, grid_height, sx, sy, sz, idx.ctypes.data, C.byref(n), bricks.ctypes.data, 4096))
    return idx, bricks[: n.value].copy()


def probe_streams(count, device=0):
    """bm_probe_streams: `count` HIP streams (raw handles, ints) that demonstrably run side by side on `device` -- HIP maps streams
    onto a few hardware queues, and streams that share one do not overlap.  release_streams() gives them back."""
    L = _lib.load()
    arr = (C.c_void_p * int(count))()
    _lib.check(L.bm_probe_streams(int(device), int(count), arr))
    return [int(h) for h in arr]


def release_streams(handles):
    L = _lib.load()
    arr = (C.c_void_p * len(handles))(*handles)
    L.bm_release_streams(len(handles), arr)


class Scene:
    """Scene (Scene.h:7-44): CPU-built world + its residency on ONE GPU.

    World dimensions are a constructor argument here (the reference's are constexpr, variables.h:7-8);
    the defaults are the reference's 4096 x 4096 x 512 voxels.
    """

    def __init__(self, grid_size=4096, grid_height=512, device=0):
        self._L = _lib.load()
        self.device = device
        h = C.c_void_p()
        check(self._L.bm_scene_create(device, grid_size, grid_height, C.byref(h)))
        self.gpuScene = h  # the reference passes Scene::GPUScene by value; here it is the scene handle
        self.grid_size, self.grid_height = grid_size, grid_height

    def close(self):
        if getattr(self, "gpuScene", None):
            self._L.bm_scene_destroy(self.gpuScene)
            self.gpuScene = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- reference API
    def generate(self, threads=None):
        """Scene::generate (Scene.cpp:118-194)."""
        check(self._L.bm_scene_generate(self.gpuScene, threads or os.cpu_count() or 1))
        return self

    def generate_supercell(self, start_x, start_y, start_z):
        """Scene::generate_supercell (Scene.cpp:44-116), host only."""
        check(self._L.bm_scene_generate_supercell(self.gpuScene, start_x, start_y, start_z))

    def process_load_queue(self):
        """Scene::process_load_queue (Scene.cpp:200-252) + upload (kernel.cu:141-151). Returns bricks serviced."""
        n = C.c_uint32(0)
        check(self._L.bm_scene_process_load_queue(self.gpuScene, C.byref(n)))
        return int(n.value)

    def dump(self, path="dump.txt"):
        """Scene::dump (Scene.cpp:254-258)."""
        check(self._L.bm_scene_dump(self.gpuScene, path.encode()))

    # ---- additions the BASELINE configs need
    def preload_all(self):
        check(self._L.bm_scene_preload_all(self.gpuScene))
        return self

    def reset_residency(self):
        check(self._L.bm_scene_reset_residency(self.gpuScene))
        return self

    def set_lod(self, lod_distance_8x8x8=600000, lod_distance_2x2x2=100000):
        check(self._L.bm_scene_set_lod(self.gpuScene, lod_distance_8x8x8, lod_distance_2x2x2))
        return self

    def set_queue_capacity(self, capacity):
        check(self._L.bm_scene_set_queue_capacity(self.gpuScene, capacity))
        return self

    def set_streaming_mode(self, overlapped):
        """False: reference order (service right after the frame). True: double-buffered rings, no host wait."""
        check(self._L.bm_scene_set_streaming_mode(self.gpuScene, int(bool(overlapped))))
        return self

    def info(self):
        i = bm_scene_info()
        check(self._L.bm_scene_get_info(self.gpuScene, C.byref(i)))
        return {name: int(getattr(i, name)) for name, _ in bm_scene_info._fields_}

    def host_supercell(self, sc):
        idx = np.zeros(4096, np.uint32)
        n = C.c_uint32(0)
        check(self._L.bm_scene_host_supercell(self.gpuScene, sc, idx.ctypes.data, C.byref(n), None, 0))
        bricks = np.zeros((int(n.value), 16), np.uint32)
        if n.value:
            check(self._L.bm_scene_host_supercell(self.gpuScene, sc, None, None, bricks.ctypes.data, n.value))
        return idx, bricks

    def device_indices(self, sc):
        idx = np.zeros(4096, np.uint32)
        check(self._L.bm_scene_device_indices(self.gpuScene, sc, idx.ctypes.data))
        return idx

    def device_brick(self, sc, device_slot):
        out = np.zeros(16, np.uint32)
        check(self._L.bm_scene_device_brick(self.gpuScene, sc, device_slot, out.ctypes.data))
        return out

    def column_heights(self, sx, sy):
        out = np.zeros((128, 128), np.float32)
        check(self._L.bm_scene_column_heights(self.gpuScene, sx, sy, out.ctypes.data))
        return out

    def synchronize(self):
        check(self._L.bm_synchronize(self.gpuScene))

    def last_render_ms(self):
        ms = C.c_float(0)
        check(self._L.bm_last_render_ms(self.gpuScene, C.byref(ms)))
        return float(ms.value)

    def render_times(self, capacity=256):
        """Kernel durations (ms) of the most recent launches, oldest first (hipEvents on the launch stream)."""
        ms = np.zeros(capacity, np.float32)
        n = C.c_int(0)
        check(self._L.bm_render_times(self.gpuScene, ms.ctypes.data, capacity, C.byref(n)))
        return ms[: n.value].copy()

    def counters(self):
        c = bm_counters()
        check(self._L.bm_counters_read(self.gpuScene, C.byref(c)))
        return c.as_dict()

    def sched_stats(self):
        """Per-phase run / active-lane counts of the wave scheduler (instrumented launches only)."""
        st = _lib.bm_sched_stats()
        check(self._L.bm_sched_stats_read(self.gpuScene, C.byref(st)))
        return {n: int(getattr(st, n)) for n in _lib.SCHED_NAMES}

    def sched_detail(self):
        """-DBM_PHASE_TIMING builds: time split of the shade pass and loop lengths of the candidate pass (bm_sched_detail_read)."""
        out = (C.c_uint64 * 8)()
        check(self._L.bm_sched_detail_read(self.gpuScene, out))
        names = ("connect_cycles", "shade_hit_cycles", "sky_cycles", "primary_cycles", "setup_cycles", "brick_passes", "brick_loop_trips", "brick_lane_steps")
        return dict(zip(names, (int(v) for v in out)))

    def counters_reset(self):
        check(self._L.bm_counters_reset(self.gpuScene))

    def render(self, camera: Camera, params: FrameParams, accum, debug=None, stream=None):
        """bm_render_frame: add params.spp paths per pixel into `accum` (torch CUDA float32 [rows, W, 4])."""
        import torch
        rows = local_rows(params)
        assert accum.is_cuda and accum.dtype == torch.float32 and accum.is_contiguous()
        assert accum.numel() == rows * params.width * 4, "accum must be [local_rows, width, 4]"
        dbg_ptr = None
        if debug is not None:
            assert debug.is_cuda and debug.dtype == torch.int32 and debug.is_contiguous() and debug.numel() == rows * params.width * 8
            dbg_ptr = C.c_void_p(debug.data_ptr())
        if stream is None:
            stream = torch.cuda.current_stream(accum.device).cuda_stream
        cam_c, par_c = camera.to_c(), params.to_c()
        check(self._L.bm_render_frame(self.gpuScene, C.byref(cam_c), C.byref(par_c), C.c_void_p(accum.data_ptr()), dbg_ptr,
                                      C.c_void_p(stream)))

    def render_frames(self, cameras, params, accums, debugs=None, stream=None):
        """bm_render_frames: `len(params)` consecutive frames -- the reference's per-frame loop, main.cpp:117-147 -- as ONE launch (the
        frame ring: every wave walks from a used-up frame to the next by itself).  cameras: one Camera for all frames or one per frame;
        accums: one tensor for all frames (production frames add with float atomics) or one per frame; debugs: None or one entry per frame
        (None or a hit-record tensor of its own)."""
        import torch
        n = len(params)
        cams = list(cameras) if isinstance(cameras, (list, tuple)) else [cameras] * n
        accs = list(accums) if isinstance(accums, (list, tuple)) else [accums] * n
        dbgs = list(debugs) if debugs is not None else None
        assert len(cams) == n and len(accs) == n and (dbgs is None or len(dbgs) == n)
        cam_c = (_lib.bm_camera * n)(*[c.to_c() for c in cams])
        par_c = (_lib.bm_frame_params * n)(*[p.to_c() for p in params])
        acc_p = (C.c_void_p * n)()
        dbg_p = (C.c_void_p * n)() if dbgs is not None else None
        for i in range(n):
            rows = local_rows(params[i])
            a = accs[i]
            assert a.is_cuda and a.dtype == torch.float32 and a.is_contiguous() and a.numel() == rows * params[i].width * 4, "accum must be [local_rows, width, 4]"
            acc_p[i] = a.data_ptr()
            if dbgs is not None and dbgs[i] is not None:
                d = dbgs[i]
                assert d.is_cuda and d.dtype == torch.int32 and d.is_contiguous() and d.numel() == rows * params[i].width * 8
                dbg_p[i] = d.data_ptr()
        if stream is None:
            stream = torch.cuda.current_stream(accs[0].device if accs else None).cuda_stream
        check(self._L.bm_render_frames(self.gpuScene, n, cam_c, par_c, acc_p, dbg_p, C.c_void_p(stream)))

    def resolve(self, accum, out=None, stream=None):
        """blit_onto_framebuffer (kernel.cu:348-364) into an offscreen float4 tensor."""
        import torch
        if out is None:
            out = torch.empty_like(accum)
        if stream is None:
            stream = torch.cuda.current_stream(accum.device).cuda_stream
        check(self._L.bm_resolve(self.gpuScene, C.c_void_p(accum.data_ptr()), C.c_void_p(out.data_ptr()), accum.numel() // 4,
                                 C.c_void_p(stream)))
        return out


# numpy views of the queue records (variables.h:43-52 RayQueue, :54-59 ShadowQueue)
RAY_QUEUE_DTYPE = np.dtype([("origin", "<f4", 3), ("direction", "<f4", 3), ("throughput", "<f4", 3), ("normal", "<f4", 3),
                            ("distance", "<f4"), ("identifier", "<i4"), ("bounces", "<i4"), ("pixel_index", "<u4")])
SHADOW_QUEUE_DTYPE = np.dtype([("origin", "<f4", 3), ("direction", "<f4", 3), ("color", "<f4", 3), ("pixel_index", "<u4")])


class Wavefront:
    """The reference's own schedule (kernel.cu:366-439): every `frame()` is one launch_kernels call -- primary_rays
    tops the work queue up to `queue_size` (ray_queue_buffer_size, variables.h:61), extend, shade, connect, swap --
    so a path needs max_bounces + 1 frames.  Holds what the reference keeps in statics / __device__ globals
    (frame counter, start_position, primary_ray_cnt) and in State (the two ray queues and the shadow queue)."""

    def __init__(self, scene: Scene, queue_size=2 * 1048576):
        self._L = _lib.load()
        self.scene, self.queue_size = scene, queue_size
        self.handle = C.c_void_p()
        check(self._L.bm_wavefront_create(scene.gpuScene, queue_size, C.byref(self.handle)))

    def close(self):
        if getattr(self, "handle", None):
            self._L.bm_wavefront_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # pragma: no cover
            pass

    def reset(self):
        """reset_buffer branch of launch_kernels (:397-403); the caller zeroes the accumulation buffer."""
        check(self._L.bm_wavefront_reset(self.handle))

    def frame(self, camera: Camera, params: FrameParams, accum, stream=None):
        import torch
        assert accum.is_cuda and accum.dtype == torch.float32 and accum.is_contiguous()
        assert accum.numel() == params.width * params.height * 4, "accum must be [height, width, 4]"
        if stream is None:
            stream = torch.cuda.current_stream(accum.device).cuda_stream
        cam_c, par_c = camera.to_c(), params.to_c()
        check(self._L.bm_wavefront_frame(self.handle, C.byref(cam_c), C.byref(par_c), C.c_void_p(accum.data_ptr()), C.c_void_p(stream)))

    def stats(self):
        out = np.zeros(6, np.uint32)
        check(self._L.bm_wavefront_stats(self.handle, out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return dict(survivors=int(out[0]), shadow=int(out[1]), start_position=int(out[2]), frame=int(out[3]),
                    generated=int(out[4]), primary_ray_cnt=int(out[5]))

    def read_queue(self, which, first=0, count=None):
        """which = "work" (RayQueue records; the survivors of the last frame lead) or "shadow"."""
        kind = {"work": 0, "shadow": 1}[which]
        dtype = RAY_QUEUE_DTYPE if kind == 0 else SHADOW_QUEUE_DTYPE
        if count is None:
            count = self.queue_size - first
        out = np.zeros(count, dtype)
        check(self._L.bm_wavefront_read_queue(self.handle, kind, first, count, out.ctypes.data_as(C.c_void_p)))
        return out

    def counters(self, which="both"):
        """traversal counters of the BM_FLAG_COUNTERS frames: of the "extend" kernel, the "connect" kernel, or "both" """
        c = bm_counters()
        check(self._L.bm_wavefront_counters_read(self.handle, {"extend": 0, "connect": 1, "both": 2}[which], C.byref(c)))
        return c.as_dict()

    def counters_reset(self):
        check(self._L.bm_wavefront_counters_reset(self.handle))

    def sched_stats(self, which="extend"):
        """wave-scheduler statistics of the BM_FLAG_COUNTERS frames of the "extend" or "connect" kernel"""
        out = (C.c_uint64 * 6)()
        check(self._L.bm_wavefront_sched_stats_read(self.handle, {"extend": 0, "connect": 1}[which], out))
        names = ("step_runs", "step_lanes", "candidate_runs", "candidate_lanes", "refills", "refill_rays")
        return dict(zip(names, (int(v) for v in out)))

    def times(self):
        """hipEvent durations (ms) of the last frame."""
        ms = (C.c_float * 5)()
        check(self._L.bm_wavefront_times(self.handle, ms))
        return dict(total=ms[0], primary=ms[1], extend=ms[2], shade=ms[3], connect=ms[4])


class State:
    """state.h:5-34 minus the wavefront queues and the GL interop: owns the float4 accumulation
    ("blit") buffer of this process' shard of the frame."""

    def __init__(self, screen_width, screen_height, device=0, band_rows=0, shard_rank=0, shard_count=1):
        import torch
        self.screen_width, self.screen_height = screen_width, screen_height
        self.device = torch.device("cuda", device)
        self.band_rows, self.shard_rank, self.shard_count = band_rows, shard_rank, shard_count
        self._alloc()

    def _alloc(self):
        import torch
        rows = local_rows(FrameParams(self.screen_width, self.screen_height, band_rows=self.band_rows,
                                      shard_rank=self.shard_rank, shard_count=self.shard_count))
        self.local_rows = rows
        self.blit_buffer = torch.zeros((rows, self.screen_width, 4), dtype=torch.float32, device=self.device)

    def screen_resize(self, screen_width, screen_height):
        self.screen_width, self.screen_height = screen_width, screen_height
        self._alloc()


@dataclass
class _LaunchStatics:
    """The function-local statics of launch_kernels (kernel.cu:367-382)."""
    first_time: bool = True
    frame: int = 1
    sample_base: int = 0
    last: tuple = field(default_factory=tuple)
    sun_position: tuple = (0.05, 0.1)
    sun_position_changed: bool = True


_statics = _LaunchStatics()


def launch_kernels(state: State, blit_buffer, gpuScene: Scene, camera: Camera, spp=1, max_bounces=3, flags=0,
                   sun_position=None, statics=None, queues: "Wavefront" = None):
    """launch_kernels (launch.h:6, kernel.cu:366-439) for the per-pixel design.

    Differences forced by the redesign (DESIGN.md "Boundary"): no GL surface and no ray queues
    (paths live in registers); one call traces `spp` complete paths per pixel instead of advancing
    every in-flight path by one bounce.  As in the reference, a change of camera position /
    direction / focal distance / lens radius or of the sun resets the accumulation buffer
    (kernel.cu:387-403).  Returns 0 (the reference always returns cudaSuccess, kernel.cu:438).

    With `queues` (a Wavefront: the reference's ray_buffer_work / ray_buffer_next / shadow_queue_buffer) the call is
    the reference's own schedule instead: every path in flight advances by one segment, `spp` is ignored.
    """
    st = statics or _statics
    if sun_position is not None and tuple(sun_position) != st.sun_position:
        st.sun_position = tuple(sun_position)
        st.sun_position_changed = True
    key = (tuple(camera.position), tuple(camera.direction), camera.focalDistance, camera.lensRadius)
    reset_buffer = key != st.last
    if st.sun_position_changed:
        st.sun_position_changed = False
        reset_buffer = True
    if reset_buffer:
        blit_buffer.zero_()
        st.sample_base = 0
    if queues is not None:
        assert state.shard_count == 1, "the queue schedule does not shard"
        if reset_buffer and not st.first_time:
            queues.reset()
        queues.frame(camera, FrameParams(state.screen_width, state.screen_height, max_bounces=max_bounces, flags=flags,
                                         sun_position=st.sun_position), blit_buffer)
        st.frame += 1
        st.first_time = False
        st.last = key
        return 0
    params = FrameParams(state.screen_width, state.screen_height, spp=spp, sample_base=st.sample_base, max_bounces=max_bounces,
                         base_frame=1, flags=flags, band_rows=state.band_rows, shard_rank=state.shard_rank,
                         shard_count=state.shard_count, sun_position=st.sun_position)
    gpuScene.render(camera, params, blit_buffer)
    st.sample_base += spp
    st.frame += 1
    st.first_time = False
    st.last = key
    return 0
