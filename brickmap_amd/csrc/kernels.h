// kernels.h -- host-callable launchers implemented in trace.hip and wavefront.hip
#pragma once
#include <hip/hip_runtime.h>

#include "device_types.h"

namespace bm {
constexpr size_t kWorkCounterBytes = 64 * 32 * sizeof(uint32_t); // up to 64 chunk counters, one 128-byte line each
int trace_blocks_per_cu(bool instrumented, bool xcd_handout, bool helpers, int ring = 0); // resident workgroups per CU of that instantiation (ring: 0 = a launch of one frame, 1 = of several, 2 = of several uniform ones)
// blocks_per_cu_cap: 0 = as many workgroups per CU as the instantiation keeps resident; > 0 = at most that many (tuning runs)
// fc_dev[0], fc_dev[1], ... are the constants of the frames of this launch (the frame ring, trace.hip; the last entry has frames_after == 0): each
// names its own accumulation / hit-record buffers; work_counter is the first of as many zeroed blocks of kWorkCounterBytes; fc = host copy of fc_dev[0]
void launch_trace(const DeviceScene& sc, const FrameConstants& fc, const FrameConstants* fc_dev, DeviceCounters* counters,
				  uint32_t* work_counter, bool instrumented, int compute_units, int blocks_per_cu_cap, hipStream_t stream);
void launch_upload(const DeviceScene& sc, const uint32_t* bricks_queue, const uint32_t* indices_queue, uint32_t* arena, uint32_t count,
				   hipStream_t stream);
void launch_pool_moves(const PoolMove* moves, uint32_t count, uint32_t* arena, uint32_t* pool_base, hipStream_t stream);
void launch_snapshot_ring(const int* queue, const uint32_t* count, int* host_positions, uint32_t* host_count, uint32_t capacity, hipStream_t stream);
void launch_resolve(const float* accum, float* out, long long n, hipStream_t stream);
void launch_debug_sincos(int n, const float* x, float* s, float* c, hipStream_t stream);
void launch_debug_sky(const FrameConstants& fc, int n, const float* v, float* sun, float* sky, float* sunsky, hipStream_t stream);

// wavefront mode (wavefront.hip)
int wavefront_blocks_per_cu(bool connect, bool instrumented);
void launch_wf_primary(WfState* st, WfRay* work, const FrameConstants* fc_dev, uint32_t queue_size, uint32_t pixels, hipStream_t stream);
void launch_wf_trace(bool connect, const DeviceScene& sc, const FrameConstants* fc_dev, WfState* st, WfRay* work, const WfShadow* shadow, float* accum,
					 DeviceCounters* counters, uint32_t queue_size, int resident_blocks, void* cold_scratch, hipStream_t stream);
void launch_wf_shade(const WfRay* work, WfRay* next, WfShadow* shadow, float* accum, void* block_counts, WfState* st, const FrameConstants* fc_dev,
					 uint32_t queue_size, hipStream_t stream);
} // namespace bm
