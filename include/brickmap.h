/*
 * brickmap.h -- C-ABI of libbrickmap_hip.so: the MI355X (gfx950) brickmap path tracer.
 *
 * This is the drop-in boundary for ONE hot path of stijnherfst/BrickMap: the per-frame
 * path-trace launch (reference src/launch.h:6, src/kernel.cu:366-439) plus the Scene
 * object that feeds it (reference src/Scene.h:7-44, src/Scene.cpp:29-258).  Plain C types
 * only; every entry point returns 0 on success or a non-zero hipError_t / BM_E* code and
 * leaves a message for bm_last_error_string().  (The reference aborts the process inside
 * its cuda() macro, src/assert_cuda.cpp:3-13; the C++ mirror in brickmap.hpp keeps that
 * behaviour on top of these return codes.)
 *
 * Not thread-safe per scene; one scene per GPU (the reference is single-threaded,
 * src/main.cpp:117-182).  All file:line citations are into the reference's src/.
 */
#ifndef BRICKMAP_H
#define BRICKMAP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BM_API __attribute__((visibility("default")))

/* error codes outside the hipError_t range */
#define BM_EINVAL 10001 /* bad argument                                   */
#define BM_ESTATE 10002 /* call made in the wrong state (e.g. not generated) */

/* index-word layout, variables.h:29-33 */
#define BM_BRICK_INDEX_BITS 0x00000FFFu
#define BM_BRICK_LOD_BITS 0x000FF000u
#define BM_BRICK_REQUESTED_BIT 0x20000000u
#define BM_BRICK_UNLOADED_BIT 0x40000000u
#define BM_BRICK_LOADED_BIT 0x80000000u

/* frame flags */
#define BM_FLAG_PRIMARY_ONLY 1u /* BASELINE config 1: extend the primary ray only          */
#define BM_FLAG_COUNTERS 2u     /* accumulate the traversal counters (instrumented kernel)  */
#define BM_FLAG_SAMPLE_ITEMS 4u /* schedule (4x4 chunk, sample) work items instead of pixels: samples of a pixel run on
                                   different lanes and are summed with float atomics (order not fixed).  debug_dev then
                                   holds an order-independent digest per pixel: words 4-7 are the SUMS (mod 2^32) over the
                                   samples of the per-sample path hashes / ray counts / cell counts (zero the buffer first),
                                   words 0-3 the first-hit record of the launch's first sample.
                                   For shards with few pixels and many samples, e.g. 1/N row bands of a multi-GPU frame; also the
                                   faster mode of any frame with several samples per pixel (1080p at 4 spp: 3.7 ms against 4.4,
                                   4K at 4 spp: 23.0 against 24.0 -- shorter items, shorter tail) when a fixed summation order is not needed */

/* (8u: retired -- the K-slot schedule of rounds 4-5, tools/variants/kslot.patch; unknown bits are refused with BM_EINVAL) */

#define BM_FLAG_ORDERED 16u     /* every pixel's events are accumulated in path order by the one lane that owns it and written back with
                                   one plain store: a frame's sums are reproducible bit for bit.  WITHOUT this flag (the default) a wave
                                   may hand a path's shadow ray to one of its idle lanes (csrc/trace.hip HELP): the same rays are
                                   traced, the unoccluded sun light is added to the pixel with float atomics like the reference's
                                   connect does (kernel.cu:341-343), and radiance is equal up to summation order (~1e-7 relative); and a
                                   frame with several samples per pixel is scheduled as (4x4 chunk, sample) work items, as if
                                   BM_FLAG_SAMPLE_ITEMS were set (1080p at 4 spp: 3.4 ms against 4.0).
                                   Frames that write hit records (debug_dev != NULL, without BM_FLAG_RAY_DIGEST) and primary-only frames are always ordered. */

#define BM_FLAG_RAY_DIGEST 32u  /* with debug_dev: the frame keeps its production plan (helper lanes, (chunk, sample) items -- NOT forced ordered) and
                                   debug_dev holds an order-independent digest per pixel instead of hash chains in path order: words 4 / 5 are the
                                   SUMS (mod 2^32) over the pixel's extend / shadow rays of keyed per-ray hashes -- key = sample << 8 | segment of
                                   the ray's path; extend: hit, distance bits, normal | level, brick id, voxel id; shadow: occluded, occluder --
                                   word 6 the ray counts, word 7 the cells visited, words 0-3 the first-hit record of the launch's first sample;
                                   each added by the lane that traced the ray (zero the buffer first).  What pins the TIMED instantiation's hits
                                   to the oracle bit for bit (oracle.c render_pixel holds the same sums).  spp * segments < 65536 per launch. */

typedef struct bm_scene bm_scene; /* Scene + its GPUScene view (Scene.h:7-44), one GPU */

/* camera.h:3-10 -- only the fields the kernels read (launch_kernels:384-385,416). */
typedef struct bm_camera {
	float position[3];    /* default (512,512,300)                      */
	float direction[3];   /* unit; Camera::update, camera.cpp:48-54      */
	float up[3];          /* default (0,0,1)                             */
	float focal_distance; /* default 1                                   */
	float lens_radius;    /* default 0                                   */
} bm_camera;

/*
 * One launch = `spp` complete paths per pixel of this shard's rows (the reference advances
 * every path one bounce per launch_kernels call; see DESIGN.md "Canonical path").
 * Sample s of pixel p (p = y*width + x, global) draws its random numbers exactly as the
 * reference does for queue slot  slot = p + (sample_base+s)*width*height  in frame
 * `base_frame + bounce` (kernel.cu:165,252).
 * Row sharding: global row y belongs to shard (y / band_rows) % shard_count; a shard's rows
 * are packed in increasing y into its local buffer.  (height, 0, 1) renders everything.
 */
typedef struct bm_frame_params {
	int32_t width, height;
	int32_t spp;
	int32_t sample_base;
	int32_t max_bounces;  /* kernel.cu:13 MAX_BOUNCES = 3 -> at most 4 segments per path */
	uint32_t base_frame;  /* kernel.cu:369 `frame` starts at 1                          */
	uint32_t flags;       /* BM_FLAG_*                                                  */
	int32_t band_rows, shard_rank, shard_count;
	float sun_position[2]; /* variables.cpp:3 default (0.05, 0.1)                        */
} bm_frame_params;

typedef struct bm_scene_info {
	int32_t grid_size, grid_height; /* voxels (variables.h:7-8, runtime here)            */
	int32_t supergrid_xy, supergrid_z, supercells;
	int32_t queue_capacity;         /* variables.h:35 brick_load_queue_size, default 1024 */
	int32_t lod_distance_8x8x8, lod_distance_2x2x2; /* variables.h:24-27                  */
	int32_t generated, on_device;
	uint64_t total_bricks;          /* non-empty bricks on the host                      */
	uint64_t resident_bricks;       /* bricks currently in the device arena              */
	uint64_t index_bytes, brick_bytes; /* device allocations: index grid; brick arena as allocated (grows by residency) */
	uint64_t pool_bytes;            /* part of the arena handed to supercell pools (16-brick pools that double, Scene.cpp:231-251) */
	uint64_t cube_field_bytes;      /* octant cube field of the walk (8 bytes per brick cell) */
	uint64_t arena_growths;         /* times the brick arena grew while bricks were resident (pool growth, Scene.cpp:231-251)      */
	uint64_t arena_copy_growths;    /* ... of which by device synchronisation + reallocation + copy: 0 when the arena is a virtual
	                                   address range that physical chunks are mapped into (arena_virtual)                        */
	int32_t arena_virtual;          /* 1: hipMemAddressReserve / hipMemMap arena (grows without copy or synchronisation)          */
	int32_t failed;                 /* 1: a streaming batch could not be completed; frames are refused until the residency is reset */
	uint64_t stream_batches;        /* upload batches queued since the residency was last reset                                   */
	uint64_t stream_host_ns;        /* host time spent staging them (validate, copy bricks to pinned memory, hand out slots, queue) */
} bm_scene_info;

/* traversal counters (BM_FLAG_COUNTERS); same order as oracle/oracle.c orc_counters */
typedef struct bm_counters {
	uint64_t index_loads, brick_tests, byte_tests, voxel_steps, extend_rays, shadow_rays, requests, paths;
} bm_counters;

/* wave-scheduler statistics of the instrumented kernel (BM_FLAG_COUNTERS): how often each phase of the
 * per-wave scheduler ran and with how many of the 64 lanes active */
typedef struct bm_sched_stats {
	uint64_t step_runs, step_lanes;           /* phase A: brick-grid DDA moves       */
	uint64_t candidate_runs, candidate_lanes; /* phase B: index word + bitmask DDA   */
	uint64_t shade_runs, shade_lanes;         /* phase C: shade / next primary ray   */
	uint64_t connect_runs, connect_lanes;     /* shade passes that also held finished shadow rays (connect), and how many */
	/* shader-clock ticks spent in each phase and in the whole scheduler loop, summed over waves */
	uint64_t step_cycles, candidate_cycles, shade_cycles;
	uint64_t drain_cycles;                    /* from a wave's last (failed) refill to its exit: the end-of-frame drain */
	uint64_t total_cycles;
	uint64_t jump_runs, jump_lanes;           /* phase A, cube jumps (step_* count the single moves) */
	uint64_t waves;
} bm_sched_stats;

/* ---- errors (replaces assert_cuda.h:5 / assert_cuda.cpp:3-13) */
BM_API const char* bm_last_error_string(void);
BM_API int bm_device_count(int* count);
BM_API int bm_device_name(int device, char* buf, size_t buflen, int* compute_units);

/* ---- Scene (Scene.h:7-44) */
/* Scene::Scene (Scene.cpp:29-36): pinned staging + load_stream/kernel_stream on `device`. */
BM_API int bm_scene_create(int device, int grid_size, int grid_height, bm_scene** out);
BM_API void bm_scene_destroy(bm_scene* scene);
/* variables.h:24-27,35 made runtime; call before bm_scene_generate. */
BM_API int bm_scene_set_lod(bm_scene* scene, int lod_distance_8x8x8, int lod_distance_2x2x2);
BM_API int bm_scene_set_queue_capacity(bm_scene* scene, int capacity);
/* 0 (default): bm_scene_process_load_queue waits for the frame and services its requests at once (reference order,
 * main.cpp:142-144).  1: overlapped -- two request rings alternate; the call services the ring copied out by the
 * previous call and starts the asynchronous copy-out of the last frame's ring on the load stream, so the host never
 * waits for the GPU (a brick requested in frame k is resident from frame k+2 on; reference order: from frame k+1 on). */
BM_API int bm_scene_set_streaming_mode(bm_scene* scene, int overlapped);
/* Scene::generate (Scene.cpp:118-194): CPU world build on `threads` host threads, then the
 * device allocations in the reference's initial state (nothing resident: unloaded|lod). */
BM_API int bm_scene_generate(bm_scene* scene, int threads);
/* Scene::generate_supercell (Scene.cpp:44-116): rebuild one supercell on the host.  Only before bm_scene_generate has put
 * the world on the device (BM_ESTATE afterwards: the pools hold bricks in request order and the index words name those
 * slots; the reference never regenerates a supercell of a live scene either). */
BM_API int bm_scene_generate_supercell(bm_scene* scene, int sx, int sy, int sz);
/* BASELINE configs 1-2 "all bricks pre-loaded": device words = host words, arena = every host brick. */
BM_API int bm_scene_preload_all(bm_scene* scene);
/* back to the reference's initial residency (Scene.cpp:157-175) */
BM_API int bm_scene_reset_residency(bm_scene* scene);
/* Scene::process_load_queue (Scene.cpp:200-252) fused with the upload kernel of the next
 * launch_kernels (kernel.cu:141-151,407-414): read the request ring, stage bricks in pinned
 * memory, async H2D on the load stream, scatter into arena + index grid, reset the count.
 * *serviced = number of bricks made resident. */
BM_API int bm_scene_process_load_queue(bm_scene* scene, uint32_t* serviced);
/* Scene::dump (Scene.cpp:254-258): one line per supercell = resident brick count. */
BM_API int bm_scene_dump(bm_scene* scene, const char* path);
BM_API int bm_scene_get_info(bm_scene* scene, bm_scene_info* info);
/* test/inspection doors: host supercell content and the device index block */
BM_API int bm_scene_host_supercell(bm_scene* scene, int supercell, uint32_t* indices4096, uint32_t* brick_count,
                                   uint32_t* bricks, uint32_t brick_capacity);
BM_API int bm_scene_device_indices(bm_scene* scene, int supercell, uint32_t* indices4096);
/* the 64-byte brick stored at `device_slot` (the 12-bit slot of a DEVICE index word) of a supercell's arena region */
BM_API int bm_scene_device_brick(bm_scene* scene, int supercell, uint32_t device_slot, uint32_t* brick16);
BM_API int bm_scene_column_heights(bm_scene* scene, int sx, int sy, float* heights128x128);

/* host-only world-build doors (no device needed): the terrain generator behind Scene::generate */
BM_API int bm_host_column_heights(int grid_size, int grid_height, int sx, int sy, float* heights128x128);
BM_API int bm_host_generate_supercell(int grid_size, int grid_height, int sx, int sy, int sz, uint32_t* indices4096,
                                      uint32_t* brick_count, uint32_t* bricks, uint32_t brick_capacity);

/* test door: the constants with which the walk divides a cube-field offset by the slice pitch (floor(n / divisor) ==
 * (uint64(n) * magic >> 32) >> shift for every n < 2^30; 3 <= divisor < 2^23) */
BM_API int bm_debug_division_magic(uint32_t divisor, uint32_t* magic, int* shift);
/* The octant cube field the GPU walk reads instead of index words while it crosses empty space (no reference
 * counterpart: the reference loads one index word per visited cell, voxel.cuh:192-200).  8 planes of
 * (cells+2)^2 x (cells_height+2) bytes, x fastest, one border cell all round; plane o (bit 0 / 1 / 2 = direction
 * negative in x / y / z), cell c: edge (<= 254) of the largest cube of empty brick cells inside the grid with c as
 * its near corner, 0 = the cell holds a brick, 255 = border.  Builds the world on the host; *bytes = size needed
 * (call with field = NULL to query). */
BM_API int bm_host_cube_field(int grid_size, int grid_height, uint8_t* field, size_t capacity, size_t* bytes);

/* ---- State (state.h:5-34): the accumulation ("blit") buffer lives in device memory the
 * caller owns; these helpers exist for callers without their own allocator. */
BM_API int bm_buffer_alloc(int device, size_t bytes, void** dev_ptr);
BM_API int bm_buffer_free(int device, void* dev_ptr);
BM_API int bm_buffer_zero(int device, void* dev_ptr, size_t bytes, void* hip_stream);
BM_API int bm_buffer_read(int device, void* host_dst, const void* dev_src, size_t bytes);
BM_API int bm_buffer_write(int device, void* dev_dst, const void* host_src, size_t bytes);

/* ---- launch_kernels (launch.h:6, kernel.cu:366-439) */
/* rows of the frame owned by this shard */
BM_API int bm_local_rows(const bm_frame_params* params);
/* Adds `spp` paths per pixel into accum_dev (float4 per pixel, local_rows*width, rgb = sum of
 * radiance, a = number of terminated paths; state.h:22, kernel.cu:301,319-322,341-343).
 * debug_dev: NULL or 8 uint32 per pixel (hit records).  hip_stream: the hipStream_t to
 * launch on, used as given (NULL = the device's default stream, like the reference's <<<>>> launches).
 * Asynchronous with respect to the host.
 * Memory and ordering contract: accum_dev (and debug_dev) must be ordinary coarse-grained device memory
 * (hipMalloc / bm_buffer_alloc / a torch CUDA tensor): the wavefront mode accumulates with hardware float
 * atomics, which are not defined on fine-grained or host-mapped allocations.  A scene is not thread-safe.
 * Frames of one scene that accumulate into the SAME buffer may overlap in time (issued on different streams) only
 * with BM_FLAG_SAMPLE_ITEMS, which adds samples with float atomics; without it a pixel is read when a lane takes it
 * and written back when it is done, so such frames must be ordered (one stream, or events).  Every launch has its
 * own ticket counters and constants (a ring of 1024 frames / 256 launches in flight).  Scenes that stream bricks may have frames on
 * several streams as well: every stream is ordered behind the brick uploads it has not seen, and
 * bm_scene_process_load_queue orders itself behind the frames of all of them.  Width and height are limited to 65535, a
 * shard to 2^32 pixels. */
BM_API int bm_render_frame(bm_scene* scene, const bm_camera* camera, const bm_frame_params* params,
                           float* accum_dev, uint32_t* debug_dev, void* hip_stream);
/* The reference's frame loop -- launch_kernels once per frame, main.cpp:117-147 / kernel.cu:416-420 -- for `count` (1 ... 256)
 * consecutive frames as ONE launch of the persistent kernel (the "frame ring", csrc/trace.hip): frame i is exactly
 * bm_render_frame(scene, &cameras[i], &params[i], accum_dev[i], debug_dev ? debug_dev[i] : NULL), but a wave that finds frame i's
 * ticket counters used up finishes its own paths and starts on frame i+1 by itself, so the end of a frame -- the latency of the
 * paths that started last, a sixth of a 1080p / 1-spp frame -- is covered by the beginning of the next one instead of an idle GPU
 * (1080p / 1 spp: 1.01 ms per frame as single launches, see DESIGN.md 4.6 for the ring).  Camera, sun_position, sample_base,
 * base_frame and the buffers may differ from frame to frame; width, height, spp, max_bounces, flags and the shard must be the
 * same (BM_EINVAL otherwise).  Frames of a launch OVERLAP in time: ordered frames (BM_FLAG_ORDERED, hit records, primary-only)
 * write pixels back with plain stores and need accumulation buffers of their own; production frames add with float atomics
 * and may share one buffer, like consecutive frames of the reference's accumulation.  debug_dev: NULL, or `count` entries, each
 * NULL or a hit-record buffer of its own (BM_FLAG_RAY_DIGEST frames of one view whose sample_base steps by a constant and that share
 * their accumulation buffer may also share ONE hit-record buffer: it then holds the digest of the whole launch, samples counted from
 * the first frame's sample_base -- for sample_base stepping by spp, the digest of one frame of count x spp samples).  Results of ordered
 * frames are bit-identical to `count` single launches.  Bricks
 * requested by any frame of the launch are serviced by the next bm_scene_process_load_queue.  bm_render_times /
 * bm_last_render_ms report the launch as one duration. */
BM_API int bm_render_frames(bm_scene* scene, int count, const bm_camera* cameras, const bm_frame_params* params,
                            float* const* accum_dev, uint32_t* const* debug_dev, void* hip_stream);

/* What the library decides for a frame with these parameters (host only, no device needed): the flags after its own choice of
 * work items, whether the frame is ordered, runs helper lanes, takes the XCD-aware hand-out, and when its waves refill.
 * hit_records: the frame will be given a debug_dev buffer. */
typedef struct bm_frame_plan {
	uint32_t flags;        /* params->flags, plus BM_FLAG_SAMPLE_ITEMS where the library schedules (chunk, sample) items by itself */
	int32_t ordered;       /* 1: one lane accumulates a pixel's events in path order (reproducible sums)                          */
	int32_t helpers;       /* 1: shadow rays on helper lanes, float-atomic adds (csrc/trace.hip HELP)                             */
	int32_t sample_items;  /* 1: (4x4 chunk, sample) work items                                                                  */
	int32_t xcd_handout;   /* 1: 256x256-pixel super-tiles dealt to the eight XCDs' ticket counters (big frames)                 */
	int32_t refill_min;    /* a wave takes new work items once this many of its lanes are idle                                    */
	int32_t refill_min_in_ring; /* ... when the frame is one of several of a bm_render_frames launch (later: its end is covered)     */
	int32_t instrumented;  /* 1: the instrumented instantiation (hit records / BM_FLAG_COUNTERS) runs                             */
	int32_t tiles_x, tiles_y, local_rows;
} bm_frame_plan;
BM_API int bm_frame_plan_of(const bm_frame_params* params, int hit_records, bm_frame_plan* out);
/* resident waves per SIMD of the trace_paths instantiation <instrumented, xcd_handout, helpers> on `device` (what the register
 * budget allows: hipOccupancyMaxActiveBlocksPerMultiprocessor of the 256-thread workgroup = one wave per SIMD each) */
BM_API int bm_trace_waves_per_simd(int device, int instrumented, int xcd_handout, int helpers, int* waves);
/* The tuning overrides this process runs under, as "NAME=value NAME=value" ("" when none is set): BM_REFILL_MIN,
 * BM_XCD_HANDOUT, BM_HELPERS, BM_TRACE_BLOCKS_PER_CU -- A/B knobs read from the environment once; a measurement should echo them. */
BM_API int bm_tuning_overrides(char* buf, size_t buflen);

/* blit_onto_framebuffer (kernel.cu:348-364) into an offscreen float4 buffer: rgb/a, a=1, gamma 1/2.2 */
BM_API int bm_resolve(bm_scene* scene, const float* accum_dev, float* out_dev, int64_t n_pixels, void* hip_stream);
/* cudaDeviceSynchronize of launch_kernels:431 */
BM_API int bm_synchronize(bm_scene* scene);
/* duration of the most recent bm_render_frame kernel, measured with hipEvents on its stream (blocks) */
BM_API int bm_last_render_ms(bm_scene* scene, float* ms);
/* durations (ms) of the most recent (up to 256) bm_render_frame kernels, oldest first; blocks until they finished */
BM_API int bm_render_times(bm_scene* scene, float* ms, int capacity, int* count);
BM_API int bm_counters_read(bm_scene* scene, bm_counters* out);
BM_API int bm_counters_reset(bm_scene* scene);
BM_API int bm_sched_stats_read(bm_scene* scene, bm_sched_stats* out);
/* profiling builds (-DBM_PHASE_TIMING) only, zeros otherwise.  out8 = shader-clock ticks, summed over waves, a shade pass spends
 * in: connect, shade (hit branch), the sky model, pixel hand-back + primary ray, ray set-up; then the candidate passes that
 * walked an 8^3 brick, the sum of their loop lengths (longest walk among the lanes of the pass) and the sum of all lanes' walk lengths */
BM_API int bm_sched_detail_read(bm_scene* scene, uint64_t* out8);

/* ---- multi-GPU: the frame's interleaved row bands, one rank per GPU, gathered to the root over RCCL / xGMI.
 * No counterpart in the reference (single GPU: src/main.cpp:89 computes `multi_gpu` and never uses it).  Every rank holds a full
 * scene replica and renders its shard (band_rows / shard_rank / shard_count of bm_frame_params) into a packed local buffer;
 * bm_gather_frame is the one exchange per frame: ncclGroupStart / the root's ncclRecv from every peer into one stacked buffer /
 * the peers' ncclSend / ncclGroupEnd, then one kernel on the root that puts row y where it belongs.  RCCL is bound at run time
 * (dlopen of librccl.so.1 on the first call): a process that already holds an RCCL (torch.distributed, a host linked against
 * /opt/rocm/lib/librccl.so) shares it.  Error codes: 20000 + ncclResult_t. */
typedef struct bm_comm bm_comm;
#define BM_COMM_ID_BYTES 128 /* sizeof(ncclUniqueId) */
/* ncclGetUniqueId: called by ONE rank; the 128 bytes reach the others by the host's own means (MPI_Bcast, a file, a socket) */
BM_API int bm_comm_unique_id(void* id128);
/* ncclCommInitRank on `device`: collective over all `world` ranks (one process or thread per GPU) */
BM_API int bm_comm_create(int device, int rank, int world, const void* id128, bm_comm** out);
BM_API void bm_comm_destroy(bm_comm* comm);
/* rank and size AS THE LIBRARY REPORTS THEM (ncclCommUserRank / ncclCommCount) -- "did RCCL see N ranks" -- or, for a transport
 * without those entry points, what the communicator was created with */
BM_API int bm_comm_info(bm_comm* comm, int* rank, int* world);
/* 1 if the RCCL library can be bound in this process (dlopen + the entry points), 0 if not; starts nothing, allocates nothing */
BM_API int bm_comm_available(void);
/* packed_dev: this rank's bm_local_rows x width float4 (what bm_render_frame accumulated for its shard); frame_dev: height x width
 * float4 on the root (ignored elsewhere).  Enqueued on hip_stream: ordered behind the frame that produced packed_dev; the host
 * does not wait.  The row-to-rank map is that of bm_frame_params: row y belongs to rank (y / band_rows) % world. */
BM_API int bm_gather_frame(bm_comm* comm, const float* packed_dev, float* frame_dev, int height, int width, int band_rows, int root,
                           void* hip_stream);
/* The same exchange for a batch of `count` (1 ... 256) frames -- what a rank rendered with ONE bm_render_frames launch into ONE
 * allocation: packed_dev = count x bm_local_rows x width float4 (frame 0's rows, then frame 1's, ...), frames_dev = count x height x
 * width float4 on the root.  One ncclSend per peer for the whole batch, one group, one assembly kernel. */
BM_API int bm_gather_frames(bm_comm* comm, const float* packed_dev, float* frames_dev, int count, int height, int width, int band_rows,
                            int root, void* hip_stream);
/* sample-sharded frames (every rank renders the whole frame with its own sample_base slice): ncclReduce(sum) to the root */
BM_API int bm_reduce_frame(bm_comm* comm, const float* in_dev, float* out_dev, int64_t n_floats, int root, void* hip_stream);
/* all ranks have reached this point (an all-reduce of one word, then the host waits for the stream) */
BM_API int bm_comm_barrier(bm_comm* comm, void* hip_stream);
/* start-up check of the transport: a grouped send / receive round the ring of ranks and an all-reduce, data verified */
BM_API int bm_comm_selftest(bm_comm* comm, void* hip_stream);
/* Streams that demonstrably run side by side.  HIP maps streams onto a few hardware queues and a queue runs its packets in order:
 * two streams that share a queue do not overlap, and an exchange that waits for its frame at the head of a queue holds up whatever
 * another stream queued behind it.  A host that pipelines frames over two streams with bm_gather_frame on a third (INTEGRATION.md 1a;
 * measured: 0.98 against 1.20 ms per 1/8-shard step) takes its streams from here: `count` (1 ... 4) non-blocking streams on `device`,
 * picked from a few more candidates by timing a short spin kernel on every combination.  The caller owns them
 * (bm_release_streams, or hipStreamDestroy on each). */
BM_API int bm_probe_streams(int device, int count, void** streams_out);
BM_API void bm_release_streams(int count, void** streams);
/* test door: the root's assembly kernel alone, on one GPU -- frame row y <- packed row of rank (y / band_rows) % world, taken from
 * own_packed_dev for rank `me` and from stacked_dev (world x max_rows x width float4, rank-major) for everybody else */
BM_API int bm_debug_assemble_frame(int device, const float* own_packed_dev, const float* stacked_dev, float* frame_dev, int height, int width,
                                   int band_rows, int world, int me, int max_rows, void* hip_stream);

/* ---- wavefront mode: launch_kernels exactly as the reference schedules it (kernel.cu:366-439) --
 * one call traces ONE segment of every path in flight: primary_rays tops the work queue up to `queue_size`
 * (ray_queue_buffer_size, variables.h:61), extend, shade (survivors -> next queue, shadow rays -> shadow queue),
 * connect; then the queues are swapped (main.cpp:146).  The reference's statics / __device__ globals (frame,
 * start_position, primary_ray_cnt; kernel.cu:106-119,369) live in the bm_wavefront object.  Of `params` the
 * fields width, height, max_bounces, sun_position and flags (BM_FLAG_COUNTERS) are used; the frame number is
 * the object's own counter (starts at 1).  Survivors and shadow rays are compacted in slot order, i.e. the
 * order a sequential run of the reference produces.  Single GPU: the queue schedule does not shard. */
typedef struct bm_wavefront bm_wavefront;
/* the scene must stay alive while frames are issued on the wavefront object (destroying either first is safe) */
BM_API int bm_wavefront_create(bm_scene* scene, uint32_t queue_size, bm_wavefront** out);
BM_API void bm_wavefront_destroy(bm_wavefront* wf);
/* the reset_buffer branch of launch_kernels (:397-403): drop the paths in flight; the caller zeroes accum_dev */
BM_API int bm_wavefront_reset(bm_wavefront* wf);
BM_API int bm_wavefront_frame(bm_wavefront* wf, const bm_camera* camera, const bm_frame_params* params, float* accum_dev,
                              void* hip_stream);
/* out6 = survivors and shadow rays of the last frame, start_position, frame (next), primary rays generated by the
 * last frame, primary_ray_cnt; blocks until the device is idle */
BM_API int bm_wavefront_stats(bm_wavefront* wf, uint32_t* out6);
/* copy queue records to the host: which = 0 the work queue (64-byte RayQueue records, variables.h:43-52; after a
 * frame its first `survivors` slots hold the paths that continue), 1 the shadow queue (40-byte ShadowQueue records,
 * variables.h:54-59; first `shadow` slots) */
BM_API int bm_wavefront_read_queue(bm_wavefront* wf, int which, uint32_t first, uint32_t count, void* host_out);
/* hipEvent durations (ms) of the last frame: total, primary_rays + globals, extend, shade, connect */
BM_API int bm_wavefront_times(bm_wavefront* wf, float* ms5);
/* traversal counters of the frames run with BM_FLAG_COUNTERS: which = 0 the extend kernel, 1 the connect kernel, 2 both */
BM_API int bm_wavefront_counters_read(bm_wavefront* wf, int which, bm_counters* out);
BM_API int bm_wavefront_counters_reset(bm_wavefront* wf);
/* wave-scheduler statistics of the BM_FLAG_COUNTERS frames for kernel `which` (0 extend, 1 connect): out6 = brick-grid
 * move rounds and the lanes active in them, candidate rounds and lanes, refills and rays handed out */
BM_API int bm_wavefront_sched_stats_read(bm_wavefront* wf, int which, uint64_t* out6);

/* ---- numeric-contract probes used by the parity tests (device side of detmath.h etc.) */
BM_API int bm_debug_sincos(int device, int n, const float* x_host, float* sin_host, float* cos_host);
BM_API int bm_debug_sky(int device, const float sun_position[2], int n, const float* viewdirs_host /*3n*/,
                        float* sun_host /*3n*/, float* sky_host /*3n*/, float* sunsky_host /*3n*/);

#ifdef __cplusplus
}
#endif
#endif /* BRICKMAP_H */
