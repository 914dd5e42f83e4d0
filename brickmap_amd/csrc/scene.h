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
	int resolve(const float* accum, float* out, long long n, hipStream_t stream);
	int synchronize();
	int last_render_ms(float* ms);
	int render_times(float* ms, int capacity, int* count); // durations of the most recent launches, oldest first
	int counters_read(bm_counters* out);
	int counters_reset();
	int sched_stats_read(bm_sched_stats* out);

	// Hooks for a frame issued by other code on this scene (the wavefront mode): begin_frame orders `stream` behind
	// pending brick uploads and hands out the current device view; end_frame records the "frame done" event process_load_queue waits for.
	int begin_frame(hipStream_t stream, DeviceScene* view, DeviceCounters** counters);
	void end_frame(hipStream_t stream);
	int compute_units() const { return compute_units_; }

	World world;
	int device() const { return device_; }

	static int fill_frame_constants(const bm_camera* cam, const bm_frame_params* fp, FrameConstants* fc);

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
	bool any_frame() const { return launches_ + other_frames_ > 0; }
	bool upload_pending_ = false;

	// device memory (DeviceScene view)
	uint32_t* d_index_grid_ = nullptr;
	SuperInfo* d_super_info_ = nullptr;
	BlockInfo* d_block_grid_ = nullptr;
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
	FrameConstants* d_frame_constants_ = nullptr; // kTimingRing device copies, one per in-flight launch
	FrameConstants* h_frame_constants_ = nullptr; // pinned source of the copies
	uint32_t* d_work_counter_ = nullptr; // chunk counter of the persistent trace kernel, zeroed before each launch
	int compute_units_ = 0, blocks_per_cu_[2] = {0, 0};
	// pinned staging (Scene.cpp:30-32)
	int* h_positions_[2] = {nullptr, nullptr};
	uint32_t* h_bricks_ = nullptr;
	uint32_t* h_indices_ = nullptr;
	uint32_t* h_count_[2] = {nullptr, nullptr};
	bool overlapped_ = false, snapshot_pending_ = false;
	int ring_cur_ = 0, ring_snapshot_ = 0;
	hipEvent_t ev_snapshot_ = nullptr, ev_frame_done_ = nullptr;

	std::vector<uint32_t> brick_base_; // host copy of the prefix sums
	uint64_t total_bricks_ = 0, resident_bricks_ = 0;
	int queue_cap_ = 1024;                       // variables.h:35
	int lod8_ = 600000, lod2_ = 100000;          // variables.h:24-27
	DeviceScene view_{};
};

} // namespace bm
