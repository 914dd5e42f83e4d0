// device_types.h -- kernel argument blocks shared by the host scene code and the HIP kernels.
#pragma once
#include <cstdint>

namespace bm {

// Device view of the scene (the reference passes Scene::GPUScene by value, Scene.h:9-17).
//
// HBM layout (DESIGN.md "Data layout"):
//   index_grid  u32[supercells * 4096]   the reference's index words (variables.h:29-33), supercell-major; supercell id =
//                                         sx + sy*sg_xy + sz*sg_xy^2, word (lx + 16*ly + 256*lz) inside it -- the
//                                         reference's addressing (voxel.cuh:197-198) minus its pointer table
//   pool_base   u32[supercells]          first arena slot of each supercell's brick pool (replaces Brick** bricks,
//                                         Scene.h:11): a brick lives at arena slot pool_base[sc] + (word & 0xFFF), the
//                                         12-bit slot the index word carries, exactly as voxel.cuh:222 addresses it.
//                                         Pools start at 16 bricks and double (Scene.cpp:231-251); a grown pool moves to
//                                         a new region of the arena and its entry here is rewritten.
//   brick_arena 64 B * arena capacity    one allocation that all pools live in (power-of-two regions, free lists per
//                                         size; grows by residency, not by world size)
//   cube_field  8 planes, 1 B per cell   octant cube field (below): what the walk reads while it crosses empty space
//   load_queue  int3[queue_cap] + count  brick-request ring (voxel.cuh:228-245)
struct DeviceScene {
	uint32_t* index_grid;
	const uint32_t* pool_base;
	const uint32_t* brick_arena; // 16 words per brick
	// octant cube field (traverse.h "cube-field walk"): 8 planes of one byte per cell of the grid plus a one-cell border;
	// byte = edge of the largest empty cube with the cell as near corner along the plane's octant (bit 0 / 1 / 2 of the plane
	// number = direction negative in x / y / z), 0 = the cell holds a brick, 255 = border (outside).  Rows (x) are padded to
	// 2^cf_shift bytes, a slice is (cells + 2) rows: entry (x, y, z) of plane o is at
	//     o * cf_plane + (z + 1) * cf_pxy + ((y + 1) << cf_shift) + (x + 1)
	// and that offset is what a ray keeps as its current cell (traverse.h cell_offset).
	const uint8_t* cube_field;
	int cf_shift;        // log2 of the row pitch
	uint32_t cf_pxy;     // slice pitch in bytes: (cells + 2) << cf_shift
	uint32_t cf_plane;   // bytes per plane: (cells_height + 2) * cf_pxy
	uint32_t cf_magic;   // floor(n / cf_pxy) == umulhi(n, cf_magic) >> cf_magic_shift for every n < 2^30 (scene.cpp division_magic)
	int cf_magic_shift;
	int* load_queue;             // 3 ints per entry
	uint32_t* load_queue_count;
	uint32_t queue_cap;
	int cells, cells_height; // bricks
	int sg_xy, sg_xy2;       // supercells per axis, squared
	float grid_size_f, grid_height_f;
	int lod_distance_8x8x8, lod_distance_2x2x2;
};

// a pool that has grown: `count` bricks move from arena slot `src` to `dst` (upload path, Scene.cpp:231-251)
struct PoolMove {
	uint32_t src, dst, count, supercell;
};

// Per-launch constants computed on the host (launch_kernels:371-403 and the view-independent
// part of the sky model, sunsky.cu:34-44,66-67).
struct FrameConstants {
	// camera (launch_kernels:384-385, primary_rays:154)
	float right[3], up[3], dir[3], origin[3];
	int campos[3];      // ivec3(camera.position / 8.f), kernel.cu:418
	float focal3;       // focalDistance * ImGui_slider_hack(3), kernel.cu:191-192
	float lens_radius;
	// sky
	float sun_direction[3];
	float sun_angular_cos; // cos(1.5 deg), kernel.cu:374
	float cone_extent;     // 1 - sun_angular_cos, kernel.cu:274
	// view-independent head of getConeSample(sunDirection, ...) (sunsky.cu:172-174): normalize(dir), o1, o2
	float cone_dir[3], cone_o1[3], cone_o2[3];
	float sunE;            // SunIntensity(dot(sunDirection, up))
	float rayleigh[3];     // rayleighAtX
	float mie[3];          // mieAtX = totalMie(...) * mieCoefficient
	float inv_total[3];    // 1 / (rayleighAtX + mieAtX)
	float mixf;            // clamp(pow(1 - dot(up, sunDirection), 5), 0, 1)
	// frame
	int width, height;
	int spp, sample_base, max_bounces;
	uint32_t base_frame;
	uint32_t flags;
	int band_rows, shard_rank, shard_count, local_rows;
	// launch geometry
	int tiles_x, tiles_y; // 16x16-pixel tiles covering this shard's rows
	int xcd_handout;      // 1: XCD-aware hand-out (trace.hip refill: 256x256-pixel super-tiles dealt to the eight XCDs' counters); big frames only
	int refill_min;       // a wave takes new work items once this many of its lanes are idle (frame_constants(): by the length of an item)
	// the hand-out's divisions by per-frame constants as multiply-high + shift (scene.cpp division_magic; magic == 0: the divisor is 1):
	// samples per (chunk, sample) ticket group, tiles per row, rows per band, super-tiles per row.  Exact for dividends < 2^30.
	uint32_t div_samples_magic, div_tiles_x_magic, div_band_magic, div_st_x_magic;
	int div_samples_shift, div_tiles_x_shift, div_band_shift, div_st_x_shift;
	int helpers;          // 1: shadow rays may be traced by idle lanes of the wave and added with float atomics (trace.hip HELP); 0: every pixel's events
	                      // are accumulated in path order by the one lane that owns it (BM_FLAG_ORDERED, and every frame that writes hit records)
	// the frame ring (trace.hip): where this frame's results go, and how many frames of the launch follow it (the queue kernels of
	// wavefront.hip take their buffers as arguments)
	float* accum;          // float4 per pixel of the shard
	uint32_t* dbg;         // hit records, 8 words per pixel, or null
	int frames_after;      // 0: the launch's last frame
	// uniform launches (read from the FIRST frame's constants): all frames share everything but sample_base and buffers, which step by
	// constants -- then every entry carries the first frame's sample_base / buffers and a lane adds its frame's offsets itself
	int ring_uniform;
	int ring_sample_stride;      // sample_base of frame i = sample_base + i * ring_sample_stride
	uint32_t ring_pixel_stride;  // pixel record of frame i = accum + (local pixel + i * ring_pixel_stride) float4s
};

struct DeviceCounters { // v: same order as bm_counters; sched: same order as bm_sched_stats
	unsigned long long v[8];
	unsigned long long sched[8];
	unsigned long long cycles[8]; // s_memtime ticks per scheduler phase, summed over waves: A, B, C, D, total; jump runs, jump lanes; waves
	// -DBM_PHASE_TIMING builds only (bm_sched_detail_read): where a shade pass spends its time -- connect, shade (hit branch),
	// sky model, pixel hand-back + primary ray, ray set-up -- then the number of candidate passes that walked a brick and the
	// sum of their loop lengths (the longest 8^3 walk among the pass's lanes), and the sum of the lanes' own walk lengths
	unsigned long long detail[8];
};

// ---- wavefront mode (wavefront.hip): the reference's queue records and device globals
struct WfRay { // RayQueue, variables.h:43-52 -- 64 bytes, same field order (four 16-byte loads per record)
	float origin[3], direction[3], throughput[3], normal[3];
	float distance;
	int identifier;
	int bounces;
	uint32_t pixel_index;
};
static_assert(sizeof(WfRay) == 64, "RayQueue is 64 bytes");

struct WfShadow { // ShadowQueue, variables.h:54-59 -- 40 bytes
	float origin[3], direction[3], color[3];
	uint32_t pixel_index;
};
static_assert(sizeof(WfShadow) == 40, "ShadowQueue is 40 bytes");

struct WfState { // the __device__ globals of kernel.cu:106-119, plus the ticket counters of the persistent kernels
	uint32_t primary_ray_cnt; // survivors written to the next queue by shade = rays already in the work queue
	uint32_t shadow_ray_cnt;
	uint32_t start_position;
	uint32_t generated;       // primary rays generated by the last primary_rays launch
	uint32_t last_survivors, last_shadow;
	uint32_t reserved[26];
	uint32_t extend_ticket[32];  // [0] is used; each counter on its own 128-byte line
	uint32_t connect_ticket[32];
};
static_assert(sizeof(WfState) == 384, "three 128-byte lines");

} // namespace bm
