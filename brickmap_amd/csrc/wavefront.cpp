// wavefront.cpp -- see wavefront.h
#include "wavefront.h"

#include <algorithm>
#include <cstdlib>

#include "kernels.h"

namespace bm {

Wavefront::~Wavefront() {
	if (hipSetDevice(device_) != hipSuccess) return;
	(void)hipDeviceSynchronize();
	(void)hipFree(d_work_); (void)hipFree(d_next_); (void)hipFree(d_shadow_); (void)hipFree(d_state_); (void)hipFree(d_block_counts_); (void)hipFree(d_cold_); (void)hipFree(d_counters_);
	(void)hipFree(d_frame_constants_);
	if (h_frame_constants_) (void)hipHostFree(h_frame_constants_);
	for (auto& e : ev_) if (e) (void)hipEventDestroy(e);
	for (auto& e : ev_slot_) if (e) (void)hipEventDestroy(e);
}

int Wavefront::init() {
	if (queue_size_ == 0 || queue_size_ > (1u << 30)) { set_error("bad queue size"); return BM_EINVAL; }
	BM_HIP(hipSetDevice(device_));
	BM_HIP(hipMalloc(&d_work_, static_cast<size_t>(queue_size_) * sizeof(WfRay)));
	BM_HIP(hipMalloc(&d_next_, static_cast<size_t>(queue_size_) * sizeof(WfRay)));
	BM_HIP(hipMalloc(&d_shadow_, static_cast<size_t>(queue_size_) * sizeof(WfShadow)));
	BM_HIP(hipMalloc(&d_state_, sizeof(WfState)));
	BM_HIP(hipMalloc(&d_block_counts_, (static_cast<size_t>(queue_size_) / 256 + 1) * 8));
	BM_HIP(hipMalloc(&d_cold_, static_cast<size_t>(queue_size_) * 16));
	BM_HIP(hipMalloc(&d_frame_constants_, kConstantsRing * sizeof(FrameConstants)));
	BM_HIP(hipHostMalloc(&h_frame_constants_, kConstantsRing * sizeof(FrameConstants), hipHostMallocDefault));
	BM_HIP(hipMemset(d_work_, 0, static_cast<size_t>(queue_size_) * sizeof(WfRay)));
	BM_HIP(hipMemset(d_next_, 0, static_cast<size_t>(queue_size_) * sizeof(WfRay)));
	BM_HIP(hipMemset(d_shadow_, 0, static_cast<size_t>(queue_size_) * sizeof(WfShadow)));
	BM_HIP(hipMemset(d_state_, 0, sizeof(WfState)));
	BM_HIP(hipMalloc(&d_counters_, 2 * sizeof(DeviceCounters)));
	BM_HIP(hipMemset(d_counters_, 0, 2 * sizeof(DeviceCounters)));
	for (auto& e : ev_) BM_HIP(hipEventCreate(&e));
	for (auto& e : ev_slot_) BM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
	for (int c = 0; c < 2; ++c)
		for (int i = 0; i < 2; ++i) blocks_per_cu_[c][i] = std::min(wavefront_blocks_per_cu(c != 0, i != 0), kMaxBlocksPerCu);
	if (const char* cap = std::getenv("BM_WF_BLOCKS_PER_CU")) { // experiment knob: fewer resident waves per SIMD
		const int n = std::atoi(cap);
		for (int c = 0; c < 2; ++c)
			for (int i = 0; i < 2; ++i)
				if (n > 0 && n < blocks_per_cu_[c][i]) blocks_per_cu_[c][i] = n;
	}
	return 0;
}

int Wavefront::reset() {
	reset_pending_ = true;
	return 0;
}

// One call of launch_kernels (kernel.cu:366-439) followed by the buffer swap of main.cpp:146.
int Wavefront::frame(const bm_camera* cam, const bm_frame_params* fp, float* accum, hipStream_t stream) {
	if (!accum) { set_error("null accumulation buffer"); return BM_EINVAL; }
	bm_frame_params p = *fp;
	p.base_frame = frame_;
	p.spp = 1; p.sample_base = 0;
	p.band_rows = 16; p.shard_rank = 0; p.shard_count = 1; // the queue schedule does not shard: replicas only
	FrameConstants fc;
	if (int e = Scene::fill_frame_constants(cam, &p, &fc)) return e;
	const unsigned long long pixels = static_cast<unsigned long long>(fp->width) * static_cast<unsigned long long>(fp->height);
	if (pixels > 0xFFFFFFFFull) { set_error("frame too large"); return BM_EINVAL; }
	DeviceScene view;
	if (int e = scene_->begin_frame(stream, &view, nullptr)) return e;
	const bool instrumented = (fp->flags & BM_FLAG_COUNTERS) != 0;
	DeviceCounters* const counters_extend = instrumented ? d_counters_ : nullptr;
	DeviceCounters* const counters_connect = instrumented ? d_counters_ + 1 : nullptr;
	const int slot = static_cast<int>(frame_ % kConstantsRing);
	// the pinned slot is reused every kConstantsRing frames: wait until the copy that read it last has run (see Scene::render)
	if (slot_used_[slot]) BM_HIP(hipEventSynchronize(ev_slot_[slot]));
	h_frame_constants_[slot] = fc;
	const FrameConstants* fc_dev = d_frame_constants_ + slot;
	BM_HIP(hipMemcpyAsync(d_frame_constants_ + slot, h_frame_constants_ + slot, sizeof(FrameConstants), hipMemcpyHostToDevice, stream));
	if (reset_pending_) {
		BM_HIP(hipMemsetAsync(&d_state_->primary_ray_cnt, 0, sizeof(uint32_t), stream));
		reset_pending_ = false;
	}
	const int cus = scene_->compute_units();
	BM_HIP(hipEventRecord(ev_[0], stream));
	launch_wf_primary(d_state_, d_work_, fc_dev, queue_size_, static_cast<uint32_t>(pixels), stream);
	BM_HIP(hipEventRecord(ev_[1], stream));
	launch_wf_trace(false, view, fc_dev, d_state_, d_work_, d_shadow_, accum, counters_extend, queue_size_, cus * blocks_per_cu_[0][instrumented ? 1 : 0], d_cold_, stream);
	BM_HIP(hipEventRecord(ev_[2], stream));
	launch_wf_shade(d_work_, d_next_, d_shadow_, accum, d_block_counts_, d_state_, fc_dev, queue_size_, stream);
	BM_HIP(hipEventRecord(ev_[3], stream));
	launch_wf_trace(true, view, fc_dev, d_state_, d_work_, d_shadow_, accum, counters_connect, queue_size_, cus * blocks_per_cu_[1][instrumented ? 1 : 0], d_cold_, stream);
	BM_HIP(hipEventRecord(ev_[4], stream));
	BM_HIP(hipEventRecord(ev_slot_[slot], stream)); // this frame's kernels have read the device copy of the constants
	slot_used_[slot] = true;
	BM_HIP(hipGetLastError());
	timed_ = true;
	frame_++;
	std::swap(d_work_, d_next_);
	scene_->end_frame(stream);
	return 0;
}

int Wavefront::stats(uint32_t* out6) {
	if (!out6) { set_error("null argument"); return BM_EINVAL; }
	BM_HIP(hipSetDevice(device_));
	BM_HIP(hipDeviceSynchronize());
	WfState st;
	BM_HIP(hipMemcpy(&st, d_state_, sizeof st, hipMemcpyDeviceToHost));
	out6[0] = st.last_survivors; out6[1] = st.last_shadow; out6[2] = st.start_position; out6[3] = frame_; out6[4] = st.generated;
	out6[5] = reset_pending_ ? 0u : st.primary_ray_cnt;
	return 0;
}

int Wavefront::read_queue(int which, uint32_t first, uint32_t count, void* host_out) {
	if (!host_out || which < 0 || which > 1 || first > queue_size_ || count > queue_size_ - first) { set_error("bad argument"); return BM_EINVAL; }
	BM_HIP(hipSetDevice(device_));
	BM_HIP(hipDeviceSynchronize());
	if (which == 0) BM_HIP(hipMemcpy(host_out, d_work_ + first, static_cast<size_t>(count) * sizeof(WfRay), hipMemcpyDeviceToHost));
	else BM_HIP(hipMemcpy(host_out, d_shadow_ + first, static_cast<size_t>(count) * sizeof(WfShadow), hipMemcpyDeviceToHost));
	return 0;
}

int Wavefront::counters_read(int which, bm_counters* out) {
	if (!out || which < 0 || which > 2) { set_error("bad argument"); return BM_EINVAL; }
	BM_HIP(hipSetDevice(device_));
	BM_HIP(hipDeviceSynchronize());
	DeviceCounters c[2];
	BM_HIP(hipMemcpy(c, d_counters_, sizeof c, hipMemcpyDeviceToHost));
	unsigned long long v[8];
	for (int k = 0; k < 8; ++k) v[k] = (which != 1 ? c[0].v[k] : 0ull) + (which != 0 ? c[1].v[k] : 0ull);
	out->index_loads = v[0]; out->brick_tests = v[1]; out->byte_tests = v[2]; out->voxel_steps = v[3];
	out->extend_rays = v[4]; out->shadow_rays = v[5]; out->requests = v[6]; out->paths = v[7];
	return 0;
}

int Wavefront::sched_stats_read(int which, unsigned long long* out6) {
	if (!out6 || which < 0 || which > 1) { set_error("bad argument"); return BM_EINVAL; }
	BM_HIP(hipSetDevice(device_));
	BM_HIP(hipDeviceSynchronize());
	DeviceCounters c;
	BM_HIP(hipMemcpy(&c, d_counters_ + which, sizeof c, hipMemcpyDeviceToHost));
	for (int k = 0; k < 6; ++k) out6[k] = c.sched[k];
	return 0;
}

int Wavefront::counters_reset() {
	BM_HIP(hipSetDevice(device_));
	BM_HIP(hipDeviceSynchronize());
	BM_HIP(hipMemset(d_counters_, 0, 2 * sizeof(DeviceCounters)));
	return 0;
}

int Wavefront::times(float* ms5) {
	if (!ms5) { set_error("null argument"); return BM_EINVAL; }
	if (!timed_) { set_error("no frame rendered yet"); return BM_ESTATE; }
	BM_HIP(hipSetDevice(device_));
	BM_HIP(hipEventSynchronize(ev_[4]));
	BM_HIP(hipEventElapsedTime(&ms5[0], ev_[0], ev_[4]));
	for (int k = 0; k < 4; ++k) BM_HIP(hipEventElapsedTime(&ms5[1 + k], ev_[k], ev_[k + 1]));
	return 0;
}

} // namespace bm
