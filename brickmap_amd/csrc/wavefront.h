// wavefront.h -- host side of the wavefront mode: the queues and per-call orchestration of the reference's
// launch_kernels (src/kernel.cu:366-439) with its static / __device__ state (frame, start_position,
// primary_ray_cnt; kernel.cu:106-119,369) held per object instead of in globals.
#pragma once
#include <hip/hip_runtime.h>

#include "scene.h"

namespace bm {

class Wavefront {
public:
	// the scene must outlive every frame() call; destruction itself only needs the device id
	Wavefront(Scene* scene, uint32_t queue_size) : scene_(scene), device_(scene->device()), queue_size_(queue_size) {}
	~Wavefront();
	int init();
	int reset(); // the reset_buffer branch of launch_kernels (:397-403): primary_ray_cnt = 0; the caller zeroes the frame buffer
	int frame(const bm_camera* cam, const bm_frame_params* fp, float* accum, hipStream_t stream);
	int stats(uint32_t* out6);
	int read_queue(int which, uint32_t first, uint32_t count, void* host_out);
	int times(float* ms5);
	int counters_read(int which, bm_counters* out); // 0 = extend kernel, 1 = connect kernel, 2 = both
	int counters_reset();
	int sched_stats_read(int which, unsigned long long* out6); // A runs, A lanes, B runs, B lanes, refills, rays handed out

private:
	Scene* scene_;
	int device_;
	uint32_t queue_size_;
	uint32_t frame_ = 1; // kernel.cu:369
	bool reset_pending_ = false;
	WfRay* d_work_ = nullptr;   // ray_buffer_work / ray_buffer_next (state.h:19-20), swapped after every frame (main.cpp:146)
	WfRay* d_next_ = nullptr;
	WfShadow* d_shadow_ = nullptr;
	WfState* d_state_ = nullptr;
	void* d_block_counts_ = nullptr;
	void* d_cold_ = nullptr; // 16 bytes per queue slot: ray state that only candidate resolution reads (wavefront.hip)
	DeviceCounters* d_counters_ = nullptr; // [0] extend, [1] connect (BM_FLAG_COUNTERS frames)
	static constexpr int kConstantsRing = 64;
	FrameConstants* d_frame_constants_ = nullptr;
	FrameConstants* h_frame_constants_ = nullptr;
	hipEvent_t ev_[5] = {};
	hipEvent_t ev_slot_[kConstantsRing] = {}; // "the frame that used this constants slot has finished"
	bool slot_used_[kConstantsRing] = {};
	bool timed_ = false;
	// resident workgroups per CU of the persistent kernels: the walk is instruction-bound, throughput saturates at 5 waves
	// per SIMD (2 / 3 / 4 / 5 / 6: 1546 / 1851 / 2014 / 2072 / 2062 Mrays/s, docs/HISTORY.md 5b) and a 6th only lengthens the tail
	static constexpr int kMaxBlocksPerCu = 5;
	int blocks_per_cu_[2][2] = {{0, 0}, {0, 0}}; // [connect][instrumented]
};

} // namespace bm
