// scene.h -- device-resident brickmap scene for one GPU: residency, streaming, frame launch.
// Mirrors the device half of the reference's Scene (src/Scene.cpp:29-36,152-194,200-258) and the
// host orchestration of launch_kernels (src/kernel.cu:366-439).
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/brickmap.h"
#include "device_types.h"
#include "world.h"

namespace bm {

// thread-local error slot behind bm_last_error_string()
void set_error(const std::string& msg);
const char* last_error();
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define BM_HIP(expr)                                                        \
	do {                                                                    \
		hipError_t bm_e_ = (expr);                                          \
		if (bm_e_ != hipSuccess) return ::bm::hip_fail(bm_e_, #expr, __FILE__, __LINE__); \
	} while (0)

void division_magic(uint32_t d, uint32_t* magic, int* shift); // floor(n / d) = umulhi(n, magic) >> shift for n < 2^30 (scene.cpp)

// a wave of a multi-frame launch (the frame ring) with helper lanes takes new items once this many of its lanes are idle (scene.cpp render_frames)
constexpr int kRingRefillMin = 32;
inline int ring_refill_min(int single_frame_refill_min, bool helpers, int override_refill_min) {
	return (helpers && !(override_refill_min >= 1 && override_refill_min <= 64)) ? kRingRefillMin : single_frame_refill_min;
}

// tuning overrides from the environment (scene.cpp tuning()): 0 / -1 = not set
struct Tuning {
	int refill_min = 0, xcd_handout = -1, helpers = -1, blocks_per_cu = 0;
};
const Tuning& tuning();

class Scene {
public:
	explicit Scene(int device) : device_(device) {}
	~Scene();

	int init(int grid_size, int grid_height); // Scene::Scene: streams + pinned staging
	int set_lod(int lod8, int lod2);
	int set_queue_capacity(int cap);
	int set_streaming_mode(int overlapped);
	int generate(int threads);                // Scene::generate
	int generate_supercell(int sx, int sy, int sz);
	int preload_all();
	int reset_residency();
	int process_load_queue(uint32_t* serviced); // Scene::process_load_queue + upload
	int dump(const char* path);
	int info(bm_scene_info* out);
	int device_indices(int supercell, uint32_t* out4096);
	int device_brick(int supercell, uint32_t device_slot, uint32_t* out16);
	int render(const bm_camera* cam, const bm_frame_params* fp, float* accum, uint32_t* dbg, hipStream_t stream);
	// `count` consecutive frames as one launch (the frame ring, trace.hip); dbgs may be null, and so may any of its entries
	int render_frames(int count, const bm_camera* cams, const bm_frame_params* fps, float* const* accums, uint32_t* const* dbgs, hipStream_t stream);
	int resolve(const float* accum, float* out, long long n, hipStream_t stream);
	int synchronize();
	int last_render_ms(float* ms);
	int render_times(float* ms, int capacity, int* count); // durations of the most recent launches, oldest first
	int counters_read(bm_counters* out);
	int counters_reset();
	int sched_stats_read(bm_sched_stats* out);
	int sched_detail_read(uint64_t* out8);

	// Hooks for a frame issued by other code on this scene (the wavefront mode): begin_frame orders `stream` behind
	// pending brick uploads and hands out the current device view; end_frame records the "frame done" event process_load_queue waits for.
	int begin_frame(hipStream_t stream, DeviceScene* view, DeviceCounters** counters);
	void end_frame(hipStream_t stream);
	int compute_units() const { return compute_units_; }

	World world;
	int device() const { return device_; }

	// hit_records: the frame writes per-pixel hit records, which makes it an ORDERED frame (scene.cpp)
	static int fill_frame_constants(const bm_camera* cam, const bm_frame_params* fp, FrameConstants* fc, bool hit_records = false);

private:
	int allocate_device();
	void free_device();
	int alloc_queue();
	int service_ring(int ring, uint32_t count);

	int device_;
	bool on_device_ = false;
	hipStream_t load_stream_ = nullptr, kernel_stream_ = nullptr; // Scene.cpp:34-35
	static constexpr int kTimingRing = 256; // hipEvent pairs around the most recent render launches
	hipEvent_t ev_start_[kTimingRing] = {}, ev_stop_[kTimingRing] = {};
	hipEvent_t ev_upload_ = nullptr;
	long long launches_ = 0;       // render() launches (index into the timing ring)
	long long other_frames_ = 0;   // frames issued through begin_frame / end_frame
	// Ordering between frames and brick uploads.  Frames may be issued on any number of streams (bench.py --pipeline, the
	// multi-stream tests); uploads run on the load stream.  Every stream a frame was issued on has an entry here:
	// `done` is recorded behind its most recent frame (process_load_queue orders the ring copy-out and the scatter kernel
	// behind ALL of them, not only behind the last frame launched), `upload_seen` is the upload batch the stream has
	// already been ordered behind (a frame waits for ev_upload_ when its stream has not seen the latest batch).
	struct FrameStream {
		hipStream_t stream = nullptr;
		hipEvent_t done = nullptr;
		uint64_t upload_seen = 0;
		uint64_t last_use = 0;
	};
	static constexpr size_t kMaxFrameStreams = 16;
	std::vector<FrameStream> frame_streams_;
	uint64_t upload_seq_ = 0, frame_seq_ = 0; // upload batches queued on the load stream / frames issued
	bool staging_busy_ = false;               // the pinned staging buffers belong to an upload that may still be copying
	bool failed_ = false;                     // a streaming batch could not be completed (allocation failure): residency state is undefined until reset
	int frame_begin(hipStream_t stream);      // order `stream` behind pending uploads
	int frame_end(hipStream_t stream);        // record the stream's "frame done" event
	int wait_frames_on_host();                // host waits for every frame in flight
	int order_load_stream_behind_frames();    // load stream waits for every frame in flight
	void drop_frame_streams();

	// device memory (DeviceScene view)
	uint32_t* d_index_grid_ = nullptr;
	uint32_t* d_pool_base_ = nullptr;
	uint32_t* d_arena_ = nullptr;
	uint8_t* d_cube_field_ = nullptr;
	uint64_t cube_field_bytes_ = 0;
	// two request rings: the blocking (reference-order) mode only uses ring 0; the overlapped mode alternates them so
	// that a frame can raise requests while the previous frame's ring is being copied out and serviced
	int* d_load_queue_[2] = {nullptr, nullptr};
	uint32_t* d_load_count_[2] = {nullptr, nullptr};
	uint32_t* d_bricks_queue_ = nullptr;
	uint32_t* d_indices_queue_ = nullptr;
	DeviceCounters* d_counters_ = nullptr;
	// the frame ring: constants and ticket counters of the frames in flight -- a launch takes as many consecutive entries as it has frames
	static constexpr int kFrameRing = 1024, kMaxFramesPerLaunch = 256;
	FrameConstants* d_frame_constants_ = nullptr; // kFrameRing device copies
	FrameConstants* h_frame_constants_ = nullptr; // pinned source of the copies
	uint32_t* d_work_counter_ = nullptr; // chunk counters of the persistent trace kernel: kFrameRing blocks of kWorkCounterBytes, zeroed before the launch that uses them
	long long ring_owner_[kFrameRing];   // the launch (value of launches_) that used the entry last, -1 = none
	int ring_next_ = 0;
	int compute_units_ = 0, blocks_per_cu_[2] = {0, 0};
	// pinned staging (Scene.cpp:30-32)
	int* h_positions_[2] = {nullptr, nullptr};
	uint32_t* h_bricks_ = nullptr;
	uint32_t* h_indices_ = nullptr;
	uint32_t* h_count_[2] = {nullptr, nullptr};
	bool overlapped_ = false, snapshot_pending_ = false;
	int ring_cur_ = 0, ring_snapshot_ = 0;
	hipEvent_t ev_snapshot_ = nullptr;

	// ---- brick arena: one device address range that every supercell's pool lives in (Scene.cpp:152-175,231-251 made one
	// allocator).  Regions are powers of two of at least kStartingPool bricks, handed out from per-size free lists or
	// from the top of the arena.  The arena is a RESERVED VIRTUAL RANGE sized for the world's worst case into which
	// physical chunks are mapped as residency grows (hipMemAddressReserve / hipMemCreate / hipMemMap): growing it
	// neither copies a brick nor synchronises the device, and every pointer into it stays valid for frames in flight.
	// (Devices without virtual memory management fall back to reallocate + copy behind a device synchronisation.)
	bool arena_virtual_ = false;
	size_t arena_va_bytes_ = 0, arena_granularity_ = 0;
	struct ArenaChunk { hipMemGenericAllocationHandle_t handle; size_t offset, bytes; };
	std::vector<ArenaChunk> arena_chunks_;
	uint64_t arena_growths_ = 0, arena_copy_growths_ = 0; // times the arena grew / grew by synchronise + copy
	uint64_t stream_batches_ = 0, stream_host_ns_ = 0;    // upload batches since the last residency reset / host time staging them
	int arena_open(uint64_t max_bricks);  // reserve the address range (once per world)
	int arena_unmap_all();
	void arena_close();
	static constexpr uint32_t kStartingPool = 16; // supergrid_starting_size, variables.h:15
	uint64_t arena_capacity_ = 0, arena_top_ = 0; // bricks
	uint64_t pool_bricks_ = 0;                    // bricks of capacity currently handed to pools
	std::vector<uint32_t> free_regions_[32];      // [log2 size]: arena offsets of free regions
	std::vector<std::pair<int, uint32_t>> freed_this_batch_; // (log2 size, offset) of regions vacated by the batch being built
	int arena_reserve(uint64_t bricks, bool exact = false); // make the arena at least this large (contents kept); exact: (re)size an EMPTY arena to fit
	int region_alloc(uint32_t bricks, uint32_t* offset);
	void region_free_deferred(uint32_t bricks, uint32_t offset);
	void arena_reset();
	PoolMove* h_moves_ = nullptr;                 // pinned staging of one batch's pool moves
	PoolMove* d_moves_ = nullptr;
	uint32_t moves_cap_ = 0;
	uint64_t total_bricks_ = 0, resident_bricks_ = 0;
	int queue_cap_ = 1024;                       // variables.h:35
	int lod8_ = 600000, lod2_ = 100000;          // variables.h:24-27
	DeviceScene view_{};
};

} // namespace bm
