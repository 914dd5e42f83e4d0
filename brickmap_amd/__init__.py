"""brickmap_amd -- MI355X-native voxel brickmap path tracer (hot path of stijnherfst/BrickMap).

The package is a thin host-side mirror of the reference's Scene / Camera / State /
launch_kernels interface over the C-ABI of libbrickmap_hip.so (include/brickmap.h); all
rendering happens in hand-written HIP kernels for gfx950 (brickmap_amd/csrc/trace.hip).
"""
from ._lib import (BM_FLAG_COUNTERS, BM_FLAG_ORDERED, BM_FLAG_PRIMARY_ONLY, BM_FLAG_RAY_DIGEST, BM_FLAG_SAMPLE_ITEMS, BRICK_INDEX_BITS, BRICK_LOADED_BIT, BRICK_LOD_BITS,  # noqa: F401
                   BRICK_REQUESTED_BIT, BRICK_UNLOADED_BIT, BrickmapError, load)
from .host import (FLYTHROUGH_VIEWS, RAY_QUEUE_DTYPE, SHADOW_QUEUE_DTYPE, Camera, FrameParams, Scene, State, Wavefront, flythrough_camera, frame_plan, host_column_heights, host_cube_field,  # noqa: F401
                   host_generate_supercell, launch_kernels, local_rows, probe_streams, release_streams, trace_waves_per_simd, tuning_overrides)
from . import dist  # noqa: F401

__all__ = ["Scene", "Camera", "State", "FrameParams", "Wavefront", "launch_kernels", "local_rows", "dist", "load", "BrickmapError"]
