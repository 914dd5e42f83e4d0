"""Loader for libbrickmap_hip.so (the C-ABI in include/brickmap.h).

There is deliberately no CPU fallback: if the HIP library is missing or fails to load, every
entry point of this package raises.  torch is imported first so that the library binds to the
same libamdhip64.so.7 torch already loaded (torch is plumbing here: device memory, streams,
torch.distributed).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbrickmap_hip.so")

BM_FLAG_PRIMARY_ONLY = 1
BM_FLAG_SAMPLE_ITEMS = 4
BM_FLAG_COUNTERS = 2
BM_FLAG_ORDERED = 16
BM_FLAG_RAY_DIGEST = 32
BRICK_INDEX_BITS = 0x00000FFF
BRICK_LOD_BITS = 0x000FF000
BRICK_REQUESTED_BIT = 0x20000000
BRICK_UNLOADED_BIT = 0x40000000
BRICK_LOADED_BIT = 0x80000000


class BrickmapError(RuntimeError):
    pass


class bm_camera(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("direction", C.c_float * 3), ("up", C.c_float * 3),
                ("focal_distance", C.c_float), ("lens_radius", C.c_float)]


class bm_frame_params(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("spp", C.c_int32), ("sample_base", C.c_int32),
                ("max_bounces", C.c_int32), ("base_frame", C.c_uint32), ("flags", C.c_uint32),
                ("band_rows", C.c_int32), ("shard_rank", C.c_int32), ("shard_count", C.c_int32),
                ("sun_position", C.c_float * 2)]


class bm_frame_plan(C.Structure):
    _fields_ = [("flags", C.c_uint32), ("ordered", C.c_int32), ("helpers", C.c_int32), ("sample_items", C.c_int32), ("xcd_handout", C.c_int32),
                ("refill_min", C.c_int32), ("refill_min_in_ring", C.c_int32), ("instrumented", C.c_int32), ("tiles_x", C.c_int32), ("tiles_y", C.c_int32), ("local_rows", C.c_int32)]


class bm_scene_info(C.Structure):
    _fields_ = [("grid_size", C.c_int32), ("grid_height", C.c_int32), ("supergrid_xy", C.c_int32),
                ("supergrid_z", C.c_int32), ("supercells", C.c_int32), ("queue_capacity", C.c_int32),
                ("lod_distance_8x8x8", C.c_int32), ("lod_distance_2x2x2", C.c_int32),
                ("generated", C.c_int32), ("on_device", C.c_int32),
                ("total_bricks", C.c_uint64), ("resident_bricks", C.c_uint64),
                ("index_bytes", C.c_uint64), ("brick_bytes", C.c_uint64),
                ("pool_bytes", C.c_uint64), ("cube_field_bytes", C.c_uint64),
                ("arena_growths", C.c_uint64), ("arena_copy_growths", C.c_uint64), ("arena_virtual", C.c_int32), ("failed", C.c_int32),
                ("stream_batches", C.c_uint64), ("stream_host_ns", C.c_uint64)]


COUNTER_NAMES = ("index_loads", "brick_tests", "byte_tests", "voxel_steps", "extend_rays",
                 "shadow_rays", "requests", "paths")


class bm_counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in COUNTER_NAMES]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n in COUNTER_NAMES}


SCHED_NAMES = ("step_runs", "step_lanes", "candidate_runs", "candidate_lanes", "shade_runs", "shade_lanes", "connect_runs", "connect_lanes",
               "step_cycles", "candidate_cycles", "shade_cycles", "drain_cycles", "total_cycles", "jump_runs", "jump_lanes", "waves")


class bm_sched_stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in SCHED_NAMES]


# every symbol include/brickmap.h declares: name -> (restype, argtypes)
_vp, _i, _u32p = C.c_void_p, C.c_int, C.POINTER(C.c_uint32)
SIGNATURES = {
    "bm_last_error_string": (C.c_char_p, []),
    "bm_device_count": (_i, [C.POINTER(C.c_int)]),
    "bm_device_name": (_i, [_i, C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]),
    "bm_scene_create": (_i, [_i, _i, _i, C.POINTER(_vp)]),
    "bm_scene_destroy": (None, [_vp]),
    "bm_scene_set_lod": (_i, [_vp, _i, _i]),
    "bm_scene_set_queue_capacity": (_i, [_vp, _i]),
    "bm_scene_set_streaming_mode": (_i, [_vp, _i]),
    "bm_scene_generate": (_i, [_vp, _i]),
    "bm_scene_generate_supercell": (_i, [_vp, _i, _i, _i]),
    "bm_scene_preload_all": (_i, [_vp]),
    "bm_scene_reset_residency": (_i, [_vp]),
    "bm_scene_process_load_queue": (_i, [_vp, _u32p]),
    "bm_scene_dump": (_i, [_vp, C.c_char_p]),
    "bm_scene_get_info": (_i, [_vp, C.POINTER(bm_scene_info)]),
    "bm_scene_host_supercell": (_i, [_vp, _i, _vp, _u32p, _vp, C.c_uint32]),
    "bm_scene_device_indices": (_i, [_vp, _i, _vp]),
    "bm_scene_device_brick": (_i, [_vp, _i, C.c_uint32, _vp]),
    "bm_scene_column_heights": (_i, [_vp, _i, _i, _vp]),
    "bm_host_column_heights": (_i, [_i, _i, _i, _i, _vp]),
    "bm_host_generate_supercell": (_i, [_i, _i, _i, _i, _i, _vp, _u32p, _vp, C.c_uint32]),
    "bm_host_cube_field": (_i, [_i, _i, _vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "bm_buffer_alloc": (_i, [_i, C.c_size_t, C.POINTER(_vp)]),
    "bm_buffer_free": (_i, [_i, _vp]),
    "bm_buffer_zero": (_i, [_i, _vp, C.c_size_t, _vp]),
    "bm_buffer_read": (_i, [_i, _vp, _vp, C.c_size_t]),
    "bm_buffer_write": (_i, [_i, _vp, _vp, C.c_size_t]),
    "bm_local_rows": (_i, [C.POINTER(bm_frame_params)]),
    "bm_render_frame": (_i, [_vp, C.POINTER(bm_camera), C.POINTER(bm_frame_params), _vp, _vp, _vp]),
    "bm_render_frames": (_i, [_vp, _i, C.POINTER(bm_camera), C.POINTER(bm_frame_params), C.POINTER(_vp), C.POINTER(_vp), _vp]),
    "bm_frame_plan_of": (_i, [C.POINTER(bm_frame_params), _i, C.POINTER(bm_frame_plan)]),
    "bm_trace_waves_per_simd": (_i, [_i, _i, _i, _i, C.POINTER(C.c_int)]),
    "bm_tuning_overrides": (_i, [C.c_char_p, C.c_size_t]),
    "bm_resolve": (_i, [_vp, _vp, _vp, C.c_int64, _vp]),
    "bm_synchronize": (_i, [_vp]),
    "bm_last_render_ms": (_i, [_vp, C.POINTER(C.c_float)]),
    "bm_render_times": (_i, [_vp, _vp, _i, C.POINTER(C.c_int)]),
    "bm_counters_read": (_i, [_vp, C.POINTER(bm_counters)]),
    "bm_counters_reset": (_i, [_vp]),
    "bm_sched_stats_read": (_i, [_vp, C.POINTER(bm_sched_stats)]),
    "bm_sched_detail_read": (_i, [_vp, _vp]),
    "bm_wavefront_create": (_i, [_vp, C.c_uint32, C.POINTER(_vp)]),
    "bm_wavefront_destroy": (None, [_vp]),
    "bm_wavefront_reset": (_i, [_vp]),
    "bm_wavefront_frame": (_i, [_vp, C.POINTER(bm_camera), C.POINTER(bm_frame_params), _vp, _vp]),
    "bm_wavefront_stats": (_i, [_vp, _u32p]),
    "bm_wavefront_read_queue": (_i, [_vp, _i, C.c_uint32, C.c_uint32, _vp]),
    "bm_wavefront_times": (_i, [_vp, C.POINTER(C.c_float)]),
    "bm_wavefront_counters_read": (_i, [_vp, _i, C.POINTER(bm_counters)]),
    "bm_wavefront_counters_reset": (_i, [_vp]),
    "bm_wavefront_sched_stats_read": (_i, [_vp, _i, C.POINTER(C.c_uint64)]),
    "bm_comm_unique_id": (_i, [_vp]),
    "bm_comm_create": (_i, [_i, _i, _i, _vp, C.POINTER(_vp)]),
    "bm_comm_destroy": (None, [_vp]),
    "bm_comm_info": (_i, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "bm_comm_available": (_i, []),
    "bm_debug_division_magic": (_i, [C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_int)]),
    "bm_gather_frame": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "bm_gather_frames": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "bm_reduce_frame": (_i, [_vp, _vp, _vp, C.c_int64, _i, _vp]),
    "bm_probe_streams": (_i, [_i, _i, C.POINTER(_vp)]),
    "bm_release_streams": (None, [_i, C.POINTER(_vp)]),
    "bm_comm_barrier": (_i, [_vp, _vp]),
    "bm_comm_selftest": (_i, [_vp, _vp]),
    "bm_debug_assemble_frame": (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "bm_debug_sincos": (_i, [_i, _i, _vp, _vp, _vp]),
    "bm_debug_sky": (_i, [_i, _vp, _i, _vp, _vp, _vp, _vp]),
}

_lib = None


def load():
    """dlopen libbrickmap_hip.so and bind every C-ABI symbol.  Raises BrickmapError when the HIP
    extension is missing -- the product path never falls back to a CPU implementation."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BrickmapError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C brickmap_amd/csrc`. There is no CPU fallback.")
    import torch  # noqa: F401  (loads torch's libamdhip64.so.7 first; see module docstring)
    try:
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    except OSError as e:  # pragma: no cover
        raise BrickmapError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError here = header and library out of sync
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(code):
    if code != 0:
        msg = load().bm_last_error_string()
        raise BrickmapError(f"brickmap error {code}: {msg.decode() if msg else ''}")
