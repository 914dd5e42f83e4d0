// capi.cpp -- extern "C" surface of libbrickmap_hip.so (declared in include/brickmap.h).
#include <cstring>
#include <string>
#include <vector>
#include <new>

#include "kernels.h"
#include "scene.h"
#include "wavefront.h"

struct bm_scene {
	bm::Scene impl;
	explicit bm_scene(int device) : impl(device) {}
};

struct bm_wavefront {
	bm::Wavefront impl;
	bm_wavefront(bm::Scene* scene, uint32_t queue_size) : impl(scene, queue_size) {}
};

using bm::set_error;

#define BM_NEED(scene)                                   \
	do {                                                 \
		if (!(scene)) {                                  \
			set_error("null scene handle");              \
			return BM_EINVAL;                            \
		}                                                \
	} while (0)

extern "C" {

const char* bm_last_error_string(void) { return bm::last_error(); }

int bm_device_count(int* count) {
	if (!count) { set_error("null argument"); return BM_EINVAL; }
	BM_HIP(hipGetDeviceCount(count));
	return 0;
}

int bm_device_name(int device, char* buf, size_t buflen, int* compute_units) {
	hipDeviceProp_t prop;
	BM_HIP(hipGetDeviceProperties(&prop, device));
	if (buf && buflen) {
		std::strncpy(buf, prop.gcnArchName, buflen - 1);
		buf[buflen - 1] = 0;
	}
	if (compute_units) *compute_units = prop.multiProcessorCount; // main.cpp:97 sm_cores
	return 0;
}

int bm_scene_create(int device, int grid_size, int grid_height, bm_scene** out) {
	if (!out) { set_error("null argument"); return BM_EINVAL; }
	*out = nullptr;
	bm_scene* s = new (std::nothrow) bm_scene(device);
	if (!s) { set_error("out of host memory"); return BM_EINVAL; }
	if (int e = s->impl.init(grid_size, grid_height)) {
		delete s;
		return e;
	}
	*out = s;
	return 0;
}

void bm_scene_destroy(bm_scene* scene) { delete scene; }

int bm_scene_set_lod(bm_scene* scene, int lod8, int lod2) { BM_NEED(scene); return scene->impl.set_lod(lod8, lod2); }
int bm_scene_set_queue_capacity(bm_scene* scene, int capacity) { BM_NEED(scene); return scene->impl.set_queue_capacity(capacity); }
int bm_scene_set_streaming_mode(bm_scene* scene, int overlapped) { BM_NEED(scene); return scene->impl.set_streaming_mode(overlapped); }
int bm_scene_generate(bm_scene* scene, int threads) { BM_NEED(scene); return scene->impl.generate(threads); }
int bm_scene_generate_supercell(bm_scene* scene, int sx, int sy, int sz) { BM_NEED(scene); return scene->impl.generate_supercell(sx, sy, sz); }
int bm_scene_preload_all(bm_scene* scene) { BM_NEED(scene); return scene->impl.preload_all(); }
int bm_scene_reset_residency(bm_scene* scene) { BM_NEED(scene); return scene->impl.reset_residency(); }
int bm_scene_process_load_queue(bm_scene* scene, uint32_t* serviced) { BM_NEED(scene); return scene->impl.process_load_queue(serviced); }
int bm_scene_dump(bm_scene* scene, const char* path) { BM_NEED(scene); return scene->impl.dump(path); }
int bm_scene_get_info(bm_scene* scene, bm_scene_info* info) { BM_NEED(scene); return scene->impl.info(info); }

int bm_scene_host_supercell(bm_scene* scene, int supercell, uint32_t* indices4096, uint32_t* brick_count, uint32_t* bricks, uint32_t brick_capacity) {
	BM_NEED(scene);
	bm::World& w = scene->impl.world;
	if (supercell < 0 || supercell >= static_cast<int>(w.supercells.size())) { set_error("bad supercell (or world not generated)"); return BM_EINVAL; }
	const bm::HostSupercell& c = w.supercells[supercell];
	if (c.indices.size() != static_cast<size_t>(bm::kCellsPerSupercell)) { set_error("supercell not generated"); return BM_ESTATE; }
	if (indices4096) std::memcpy(indices4096, c.indices.data(), bm::kCellsPerSupercell * sizeof(uint32_t));
	if (brick_count) *brick_count = static_cast<uint32_t>(c.bricks.size());
	if (bricks) {
		const size_t n = c.bricks.size() < brick_capacity ? c.bricks.size() : brick_capacity;
		std::memcpy(bricks, c.bricks.data(), n * sizeof(bm::Brick));
	}
	return 0;
}

int bm_scene_device_indices(bm_scene* scene, int supercell, uint32_t* indices4096) { BM_NEED(scene); return scene->impl.device_indices(supercell, indices4096); }

int bm_scene_device_brick(bm_scene* scene, int supercell, uint32_t device_slot, uint32_t* out16) { BM_NEED(scene); return scene->impl.device_brick(supercell, device_slot, out16); }

int bm_scene_column_heights(bm_scene* scene, int sx, int sy, float* heights) {
	BM_NEED(scene);
	const bm::WorldDims& d = scene->impl.world.dims;
	if (!heights || sx < 0 || sy < 0 || sx >= d.supergrid_xy || sy >= d.supergrid_xy) { set_error("bad column"); return BM_EINVAL; }
	scene->impl.world.column_heights(sx, sy, heights);
	return 0;
}

int bm_host_column_heights(int grid_size, int grid_height, int sx, int sy, float* heights) {
	bm::World w;
	if (!heights || !w.dims.set(grid_size, grid_height) || sx < 0 || sy < 0 || sx >= w.dims.supergrid_xy || sy >= w.dims.supergrid_xy) {
		set_error("bad world dimensions or column");
		return BM_EINVAL;
	}
	w.column_heights(sx, sy, heights);
	return 0;
}

int bm_host_generate_supercell(int grid_size, int grid_height, int sx, int sy, int sz, uint32_t* indices4096, uint32_t* brick_count,
							   uint32_t* bricks, uint32_t brick_capacity) {
	bm::World w;
	if (!w.dims.set(grid_size, grid_height) || sx < 0 || sy < 0 || sz < 0 || sx >= w.dims.supergrid_xy || sy >= w.dims.supergrid_xy ||
		sz >= w.dims.supergrid_z) {
		set_error("bad world dimensions or supercell");
		return BM_EINVAL;
	}
	w.generate_supercell(sx, sy, sz);
	const bm::HostSupercell& c = w.supercells[w.dims.supercell_id(sx, sy, sz)];
	if (indices4096) std::memcpy(indices4096, c.indices.data(), bm::kCellsPerSupercell * sizeof(uint32_t));
	if (brick_count) *brick_count = static_cast<uint32_t>(c.bricks.size());
	if (bricks) {
		const size_t n = c.bricks.size() < brick_capacity ? c.bricks.size() : brick_capacity;
		std::memcpy(bricks, c.bricks.data(), n * sizeof(bm::Brick));
	}
	return 0;
}

int bm_debug_division_magic(uint32_t divisor, uint32_t* magic, int* shift) {
	if (!magic || !shift || divisor < 3u || divisor >= (1u << 23)) { bm::set_error("bad argument"); return BM_EINVAL; }
	bm::division_magic(divisor, magic, shift);
	return 0;
}

int bm_host_cube_field(int grid_size, int grid_height, uint8_t* field, size_t capacity, size_t* bytes) {
	bm::World w;
	if (!w.dims.set(grid_size, grid_height)) { set_error("bad world dimensions"); return BM_EINVAL; }
	const size_t need = 8ull * (w.dims.cells + 2) * (w.dims.cells + 2) * (w.dims.cells_height + 2);
	if (bytes) *bytes = need;
	if (!field) return 0;
	if (capacity < need) { set_error("cube field buffer too small"); return BM_EINVAL; }
	w.generate(8);
	std::vector<uint8_t> f;
	w.build_cube_field(f, 8);
	std::memcpy(field, f.data(), need);
	return 0;
}

int bm_buffer_alloc(int device, size_t bytes, void** dev_ptr) {
	if (!dev_ptr) { set_error("null argument"); return BM_EINVAL; }
	BM_HIP(hipSetDevice(device));
	BM_HIP(hipMalloc(dev_ptr, bytes ? bytes : 4));
	return 0;
}
int bm_buffer_free(int device, void* dev_ptr) {
	BM_HIP(hipSetDevice(device));
	BM_HIP(hipFree(dev_ptr));
	return 0;
}
int bm_buffer_zero(int device, void* dev_ptr, size_t bytes, void* hip_stream) {
	BM_HIP(hipSetDevice(device));
	BM_HIP(hipMemsetAsync(dev_ptr, 0, bytes, static_cast<hipStream_t>(hip_stream))); // launch_kernels:399
	return 0;
}
int bm_buffer_read(int device, void* host_dst, const void* dev_src, size_t bytes) {
	BM_HIP(hipSetDevice(device));
	BM_HIP(hipDeviceSynchronize());
	BM_HIP(hipMemcpy(host_dst, dev_src, bytes, hipMemcpyDeviceToHost));
	return 0;
}
int bm_buffer_write(int device, void* dev_dst, const void* host_src, size_t bytes) {
	BM_HIP(hipSetDevice(device));
	BM_HIP(hipMemcpy(dev_dst, host_src, bytes, hipMemcpyHostToDevice));
	return 0;
}

int bm_local_rows(const bm_frame_params* p) {
	if (!p || p->band_rows <= 0 || p->shard_count <= 0 || p->height <= 0) return 0;
	// rows y with (y / band) % count == rank
	const int band = p->band_rows, count = p->shard_count, rank = p->shard_rank;
	const int full_bands = p->height / band, tail = p->height % band;
	int rows = 0;
	if (rank < full_bands) rows = ((full_bands - 1 - rank) / count + 1) * band;
	if (tail && full_bands % count == rank) rows += tail;
	return rows;
}

int bm_render_frame(bm_scene* scene, const bm_camera* camera, const bm_frame_params* params, float* accum_dev, uint32_t* debug_dev, void* hip_stream) {
	BM_NEED(scene);
	return scene->impl.render(camera, params, accum_dev, debug_dev, static_cast<hipStream_t>(hip_stream));
}
int bm_render_frames(bm_scene* scene, int count, const bm_camera* cameras, const bm_frame_params* params, float* const* accum_dev, uint32_t* const* debug_dev,
					 void* hip_stream) {
	BM_NEED(scene);
	return scene->impl.render_frames(count, cameras, params, accum_dev, debug_dev, static_cast<hipStream_t>(hip_stream));
}
int bm_frame_plan_of(const bm_frame_params* params, int hit_records, bm_frame_plan* out) {
	if (!params || !out) { set_error("null argument"); return BM_EINVAL; }
	bm_camera cam{};
	cam.direction[0] = 1.f; cam.up[2] = 1.f; cam.focal_distance = 1.f; // (the plan does not depend on the view)
	bm::FrameConstants fc;
	if (int e = bm::Scene::fill_frame_constants(&cam, params, &fc, hit_records != 0)) return e;
	std::memset(out, 0, sizeof *out);
	out->flags = fc.flags;
	out->helpers = fc.helpers;
	out->sample_items = (fc.flags & BM_FLAG_SAMPLE_ITEMS) ? 1 : 0;
	out->ordered = (fc.helpers || out->sample_items) ? 0 : 1;
	out->xcd_handout = fc.xcd_handout;
	out->refill_min = fc.refill_min;
	out->refill_min_in_ring = bm::ring_refill_min(fc.refill_min, fc.helpers != 0, bm::tuning().refill_min);
	out->tiles_x = fc.tiles_x; out->tiles_y = fc.tiles_y; out->local_rows = fc.local_rows;
	out->instrumented = (hit_records || (fc.flags & BM_FLAG_COUNTERS)) ? 1 : 0;
	return 0;
}
int bm_trace_waves_per_simd(int device, int instrumented, int xcd_handout, int helpers, int* waves) {
	if (!waves) { set_error("null argument"); return BM_EINVAL; }
	BM_HIP(hipSetDevice(device));
	// a 256-thread workgroup is one wave on each of a compute unit's four SIMDs: resident workgroups per CU = waves per SIMD
	*waves = bm::trace_blocks_per_cu(instrumented != 0, xcd_handout != 0, helpers != 0);
	if (bm::tuning().blocks_per_cu > 0 && !instrumented && *waves > bm::tuning().blocks_per_cu) *waves = bm::tuning().blocks_per_cu;
	return 0;
}
int bm_tuning_overrides(char* buf, size_t buflen) {
	if (!buf || buflen == 0) { set_error("null argument"); return BM_EINVAL; }
	const bm::Tuning& t = bm::tuning();
	std::string s;
	auto add = [&s](const char* name, int v) { if (!s.empty()) s += ' '; s += name; s += '='; s += std::to_string(v); };
	if (t.refill_min >= 1 && t.refill_min <= 64) add("BM_REFILL_MIN", t.refill_min);
	if (t.xcd_handout == 0 || t.xcd_handout == 1) add("BM_XCD_HANDOUT", t.xcd_handout);
	if (t.helpers == 0 || t.helpers == 1) add("BM_HELPERS", t.helpers);
	if (t.blocks_per_cu > 0) add("BM_TRACE_BLOCKS_PER_CU", t.blocks_per_cu);
	if (s.size() + 1 > buflen) { set_error("buffer too small"); return BM_EINVAL; }
	std::memcpy(buf, s.c_str(), s.size() + 1);
	return 0;
}
int bm_resolve(bm_scene* scene, const float* accum_dev, float* out_dev, int64_t n_pixels, void* hip_stream) {
	BM_NEED(scene);
	return scene->impl.resolve(accum_dev, out_dev, n_pixels, static_cast<hipStream_t>(hip_stream));
}
int bm_synchronize(bm_scene* scene) { BM_NEED(scene); return scene->impl.synchronize(); }
int bm_last_render_ms(bm_scene* scene, float* ms) { BM_NEED(scene); return scene->impl.last_render_ms(ms); }
int bm_render_times(bm_scene* scene, float* ms, int capacity, int* count) { BM_NEED(scene); return scene->impl.render_times(ms, capacity, count); }
int bm_counters_read(bm_scene* scene, bm_counters* out) { BM_NEED(scene); return scene->impl.counters_read(out); }
int bm_counters_reset(bm_scene* scene) { BM_NEED(scene); return scene->impl.counters_reset(); }
int bm_sched_stats_read(bm_scene* scene, bm_sched_stats* out) { BM_NEED(scene); return scene->impl.sched_stats_read(out); }
int bm_sched_detail_read(bm_scene* scene, uint64_t* out8) { BM_NEED(scene); return scene->impl.sched_detail_read(out8); }

int bm_wavefront_create(bm_scene* scene, uint32_t queue_size, bm_wavefront** out) {
	BM_NEED(scene);
	if (!out) { set_error("null argument"); return BM_EINVAL; }
	*out = nullptr;
	bm_wavefront* w = new (std::nothrow) bm_wavefront(&scene->impl, queue_size);
	if (!w) { set_error("out of host memory"); return BM_EINVAL; }
	if (int e = w->impl.init()) {
		delete w;
		return e;
	}
	*out = w;
	return 0;
}
void bm_wavefront_destroy(bm_wavefront* wf) { delete wf; }
#define BM_NEED_WF(wf)                                   \
	do {                                                 \
		if (!(wf)) {                                     \
			set_error("null wavefront handle");          \
			return BM_EINVAL;                            \
		}                                                \
	} while (0)
int bm_wavefront_reset(bm_wavefront* wf) { BM_NEED_WF(wf); return wf->impl.reset(); }
int bm_wavefront_frame(bm_wavefront* wf, const bm_camera* camera, const bm_frame_params* params, float* accum_dev, void* hip_stream) {
	BM_NEED_WF(wf);
	if (!camera || !params) { set_error("null argument"); return BM_EINVAL; }
	return wf->impl.frame(camera, params, accum_dev, static_cast<hipStream_t>(hip_stream));
}
int bm_wavefront_stats(bm_wavefront* wf, uint32_t* out6) { BM_NEED_WF(wf); return wf->impl.stats(out6); }
int bm_wavefront_read_queue(bm_wavefront* wf, int which, uint32_t first, uint32_t count, void* host_out) {
	BM_NEED_WF(wf);
	return wf->impl.read_queue(which, first, count, host_out);
}
int bm_wavefront_times(bm_wavefront* wf, float* ms5) { BM_NEED_WF(wf); return wf->impl.times(ms5); }
int bm_wavefront_counters_read(bm_wavefront* wf, int which, bm_counters* out) { BM_NEED_WF(wf); return wf->impl.counters_read(which, out); }
int bm_wavefront_sched_stats_read(bm_wavefront* wf, int which, uint64_t* out6) {
	BM_NEED_WF(wf);
	return wf->impl.sched_stats_read(which, reinterpret_cast<unsigned long long*>(out6));
}
int bm_wavefront_counters_reset(bm_wavefront* wf) { BM_NEED_WF(wf); return wf->impl.counters_reset(); }

int bm_debug_sincos(int device, int n, const float* x_host, float* sin_host, float* cos_host) {
	if (n <= 0 || !x_host || !sin_host || !cos_host) { set_error("bad argument"); return BM_EINVAL; }
	BM_HIP(hipSetDevice(device));
	float *dx = nullptr, *ds = nullptr, *dc = nullptr;
	const size_t bytes = static_cast<size_t>(n) * sizeof(float);
	BM_HIP(hipMalloc(reinterpret_cast<void**>(&dx), bytes));
	BM_HIP(hipMalloc(reinterpret_cast<void**>(&ds), bytes));
	BM_HIP(hipMalloc(reinterpret_cast<void**>(&dc), bytes));
	BM_HIP(hipMemcpy(dx, x_host, bytes, hipMemcpyHostToDevice));
	bm::launch_debug_sincos(n, dx, ds, dc, nullptr);
	BM_HIP(hipGetLastError());
	BM_HIP(hipDeviceSynchronize());
	BM_HIP(hipMemcpy(sin_host, ds, bytes, hipMemcpyDeviceToHost));
	BM_HIP(hipMemcpy(cos_host, dc, bytes, hipMemcpyDeviceToHost));
	(void)hipFree(dx); (void)hipFree(ds); (void)hipFree(dc);
	return 0;
}

int bm_debug_sky(int device, const float sun_position[2], int n, const float* viewdirs_host, float* sun_host, float* sky_host, float* sunsky_host) {
	if (n <= 0 || !sun_position || !viewdirs_host || !sun_host || !sky_host || !sunsky_host) { set_error("bad argument"); return BM_EINVAL; }
	bm_camera cam{};
	cam.direction[0] = 1.f; cam.up[2] = 1.f; cam.focal_distance = 1.f;
	bm_frame_params fp{};
	fp.width = 16; fp.height = 16; fp.spp = 1; fp.max_bounces = 3; fp.base_frame = 1; fp.band_rows = 16; fp.shard_count = 1;
	fp.sun_position[0] = sun_position[0]; fp.sun_position[1] = sun_position[1];
	bm::FrameConstants fc;
	if (int e = bm::Scene::fill_frame_constants(&cam, &fp, &fc)) return e;
	BM_HIP(hipSetDevice(device));
	const size_t bytes = static_cast<size_t>(n) * 3 * sizeof(float);
	float* d[4] = {nullptr, nullptr, nullptr, nullptr};
	for (auto& p : d) BM_HIP(hipMalloc(reinterpret_cast<void**>(&p), bytes));
	BM_HIP(hipMemcpy(d[0], viewdirs_host, bytes, hipMemcpyHostToDevice));
	bm::launch_debug_sky(fc, n, d[0], d[1], d[2], d[3], nullptr);
	BM_HIP(hipGetLastError());
	BM_HIP(hipDeviceSynchronize());
	BM_HIP(hipMemcpy(sun_host, d[1], bytes, hipMemcpyDeviceToHost));
	BM_HIP(hipMemcpy(sky_host, d[2], bytes, hipMemcpyDeviceToHost));
	BM_HIP(hipMemcpy(sunsky_host, d[3], bytes, hipMemcpyDeviceToHost));
	for (auto& p : d) (void)hipFree(p);
	return 0;
}

} // extern "C"
