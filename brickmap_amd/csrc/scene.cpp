// scene.cpp -- device residency, brick streaming and frame launch for one GPU.
#include "scene.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <unordered_map>

#include "kernels.h"

namespace bm {

// ---------------------------------------------------------------- errors
static thread_local std::string g_error;
void set_error(const std::string& msg) { g_error = msg; }
const char* last_error() { return g_error.c_str(); }
int hip_fail(hipError_t e, const char* what, const char* file, int line) {
	// the reference prints "cuda_assert: <string> <file> <line>" and exits (assert_cuda.cpp:3-13);
	// the C-ABI reports instead and lets the caller decide.
	char buf[512];
	std::snprintf(buf, sizeof buf, "hip_assert: %s (%s) %s %d", hipGetErrorString(e), what, file, line);
	set_error(buf);
	return static_cast<int>(e);
}

// ---------------------------------------------------------------- small host vector helpers (GLM operation order)
namespace {
struct V3 {
	float x, y, z;
};
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float dot3(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline V3 cross3(V3 x, V3 y) { return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y}; }
inline V3 normalize3(V3 v) { return v * (1.0f / std::sqrt(dot3(v, v))); }
constexpr float kPi = 3.1415926535897932f;

// sunsky.cu:24-26 -- double literals make the tail of this expression double
float sun_intensity(float zenith_cos) {
	const float cutoff = kPi / 1.95f, steepness = 1.5f;
	const double e = 1.0 - static_cast<double>(std::exp(-((cutoff - std::acos(zenith_cos)) / steepness)));
	return static_cast<float>(1000.0 * (0.0 < e ? e : 0.0));
}
} // namespace

void division_magic(uint32_t d, uint32_t* magic, int* shift);

// Tuning overrides, read from the environment ONCE per process (A/B runs, tools/): BM_REFILL_MIN (1 ... 64), BM_XCD_HANDOUT (0 / 1),
// BM_HELPERS (0 / 1), BM_TRACE_BLOCKS_PER_CU (> 0).  None set = the product's own rules.  bm_tuning_overrides() reports them, so that
// a measurement can say what it ran under (bench.py echoes them into its line and refuses to go on under BM_BENCH_STRICT=1).
const Tuning& tuning() {
	static const Tuning t = [] {
		Tuning v;
		auto num = [](const char* name, int unset) { const char* e = std::getenv(name); return e && *e ? std::atoi(e) : unset; };
		v.refill_min = num("BM_REFILL_MIN", 0);
		v.xcd_handout = num("BM_XCD_HANDOUT", -1);
		v.helpers = num("BM_HELPERS", -1);
		v.blocks_per_cu = num("BM_TRACE_BLOCKS_PER_CU", 0);
		return v;
	}();
	return t;
}

int Scene::fill_frame_constants(const bm_camera* cam, const bm_frame_params* fp_in, FrameConstants* fc, bool hit_records) {
	if (!fp_in) { set_error("null argument"); return BM_EINVAL; }
	if (fp_in->flags & ~(BM_FLAG_PRIMARY_ONLY | BM_FLAG_COUNTERS | BM_FLAG_SAMPLE_ITEMS | BM_FLAG_ORDERED | BM_FLAG_RAY_DIGEST)) {
		set_error("unknown frame flag (bit 8 was the retired K-slot schedule's)");
		return BM_EINVAL;
	}
	// Which frames are ORDERED (every pixel's events accumulated in path order by one lane, one plain write-back: reproducible sums)?
	// Those that ask for it, those that write hit records, primary-only frames.  Every other frame -- the production
	// default -- may add in any order, like the reference's own atomicAdds (kernel.cu:319-322,341-343): it runs with helper lanes
	// (trace.hip HELP) and, when a pixel has several samples, with (4x4 chunk, sample) work items: shorter items, a shorter tail,
	// coherent neighbouring samples (1080p at 4 spp 4.0 -> 3.4 ms, config 3 -4 %).  BM_HELPERS=0 / 1 overrides helper lanes (A/B runs).
	bm_frame_params promoted = *fp_in;
	// (hit records are chains in path order -- an ordered frame -- unless the caller asked for the order-independent ray digest)
	const bool ordered = (promoted.flags & (BM_FLAG_ORDERED | BM_FLAG_PRIMARY_ONLY)) != 0 || (hit_records && !(promoted.flags & BM_FLAG_RAY_DIGEST)) || promoted.spp < 1; // (spp = 0: nothing to trace)
	const bm_frame_params* const fp = &promoted;
	if (!cam || !fp || !fc) { set_error("null argument"); return BM_EINVAL; }
	if (fp->width <= 0 || fp->height <= 0 || fp->spp < 0 || fp->max_bounces < 0 || fp->band_rows <= 0 || fp->shard_count <= 0 ||
		fp->shard_rank < 0 || fp->shard_rank >= fp->shard_count) {
		set_error("bad frame parameters");
		return BM_EINVAL;
	}
	// the kernels pack a pixel as x | y << 16 and index the shard's packed buffers with 32-bit pixel numbers
	if (fp->width > 65535 || fp->height > 65535 || static_cast<long long>(bm_local_rows(fp)) * fp->width >= (1ll << 32)) {
		set_error("frame too large: width and height are limited to 65535 and a shard to 2^32 pixels");
		return BM_EINVAL;
	}
	if ((fp->flags & BM_FLAG_RAY_DIGEST) && (fp->max_bounces >= 255 || static_cast<long long>(fp->spp) * (fp->max_bounces + 1) >= 65536)) {
		set_error("BM_FLAG_RAY_DIGEST: the digest counts a pixel's rays in 16 bits and keys them with 8 bits of segment: spp x segments < 65536, max_bounces < 255");
		return BM_EINVAL;
	}
	const int geo_tiles_x = (fp->width + 15) / 16, geo_tiles_y = (bm_local_rows(fp) + 15) / 16;
	// XCD-aware hand-out: neighbouring rays behind ONE L2 instead of all eight.  Pays where the scene does not fit the caches and the
	// frame has enough 256x256-pixel super-tiles for eight even shares (8K: 510, 4K: 135) -- config 5 115.0 -> 111.5 ms, config 3
	// 24.05 -> 23.90; on a 1080p frame (40 super-tiles) the shares are too uneven: +8 % (profiles/r04_xcd_handout.txt)
	int geo_xcd = (static_cast<long long>(geo_tiles_x) * geo_tiles_y >= 32000) ? 1 : 0;
	if (tuning().xcd_handout == 0 || tuning().xcd_handout == 1) geo_xcd = tuning().xcd_handout;
	// The hand-out counts tickets in 32 bits (trace.hip: `my_tickets`, `base + want`).  The busiest counter owns a 1/8 share of the
	// units -- groups of four chunks, or 256x256-pixel super-tiles of 4096 chunks -- times 16 tickets per chunk and, with (chunk,
	// sample) items, per sample; every wave may overshoot a used-up counter once by up to 64.
	auto tickets_fit = [&](bool sample_items) {
		const long long tiles = static_cast<long long>(geo_tiles_x) * geo_tiles_y;
		const long long per_chunk = 16ll * (sample_items ? std::max(fp->spp, 1) : 1);
		long long share;
		if (geo_xcd) {
			const long long st = static_cast<long long>((geo_tiles_x + 15) / 16) * ((geo_tiles_y + 15) / 16);
			share = ((st + 7) / 8) * 4096ll * per_chunk;
		} else {
			share = ((tiles * 4 + 7) / 8) * 4ll * per_chunk;
		}
		return share < (1ll << 30) - (1ll << 24); // (2^30: the hand-out divides ticket numbers with 30-bit-exact multiply-high constants)
	};
	if (!tickets_fit((promoted.flags & BM_FLAG_SAMPLE_ITEMS) != 0)) { // what the caller asked for does not fit: refuse
		set_error("frame too large for the 32-bit ticket counters: tiles x samples per launch (lower spp per call, or render row-band shards)");
		return BM_EINVAL;
	}
	// (chunk, sample) items as the library's own choice -- only where their tickets fit; pixel items carry no spp factor and always
	// do at this point (helper lanes work with either: atomic_acc = HELP)
	if (!ordered && promoted.spp >= 2 && tickets_fit(true)) promoted.flags |= BM_FLAG_SAMPLE_ITEMS;
	std::memset(fc, 0, sizeof *fc);
	const V3 dir{cam->direction[0], cam->direction[1], cam->direction[2]};
	const V3 upv{cam->up[0], cam->up[1], cam->up[2]};
	const float aspect = static_cast<float>(fp->width) / static_cast<float>(fp->height);
	const V3 right = (normalize3(cross3(dir, upv)) * 1.5f) * aspect;   // launch_kernels:384
	const V3 up = normalize3(cross3(right, dir)) * 1.5f;               // launch_kernels:385
	fc->right[0] = right.x; fc->right[1] = right.y; fc->right[2] = right.z;
	fc->up[0] = up.x; fc->up[1] = up.y; fc->up[2] = up.z;
	for (int i = 0; i < 3; ++i) {
		fc->dir[i] = cam->direction[i];
		fc->origin[i] = cam->position[i];
		fc->campos[i] = static_cast<int>(cam->position[i] / 8.f); // kernel.cu:418
	}
	fc->focal3 = cam->focal_distance * 3; // kernel.cu:191-192 (int 3)
	fc->lens_radius = cam->lens_radius;

	// sky constants (kernel.cu:374,393; sunsky.cu:14-18,28-30,34-44,66-67)
	fc->sun_angular_cos = std::cos(1.5f * kPi / 180.f);
	fc->cone_extent = 1.0f - fc->sun_angular_cos;
	const float px = (fp->sun_position[0] - 0.0f) * 6.28f, py = (fp->sun_position[1] - 0.5f) * 3.14f;
	const V3 sun = normalize3(V3{std::cos(px) * std::sin(py), std::sin(px) * std::sin(py), std::cos(py)});
	fc->sun_direction[0] = sun.x; fc->sun_direction[1] = sun.y; fc->sun_direction[2] = sun.z;
	{ // getConeSample's frame around the sun direction (sunsky.cu:163-174), same fp32 operations as the reference
		const V3 cd = normalize3(sun);
		const V3 ortho = std::fabs(cd.x) > std::fabs(cd.z) ? V3{-cd.y, cd.x, 0.0f} : V3{0.0f, -cd.z, cd.y};
		const V3 o1 = normalize3(ortho);
		const V3 o2 = normalize3(cross3(cd, o1));
		fc->cone_dir[0] = cd.x; fc->cone_dir[1] = cd.y; fc->cone_dir[2] = cd.z;
		fc->cone_o1[0] = o1.x; fc->cone_o1[1] = o1.y; fc->cone_o1[2] = o1.z;
		fc->cone_o2[0] = o2.x; fc->cone_o2[1] = o2.y; fc->cone_o2[2] = o2.z;
	}
	const V3 sky_up{0.0f, 0.0f, 1.0f};
	fc->sunE = sun_intensity(dot3(sun, sky_up));
	const float rayleigh[3] = {5.176821E-6f, 1.2785348E-5f, 2.8530756E-5f};
	const float lambda[3] = {680E-9f, 550E-9f, 450E-9f};
	const float K[3] = {0.686f, 0.678f, 0.666f};
	const float c = static_cast<float>((0.2 * static_cast<double>(1.f)) * 10E-18); // turbidity 1
	const float mie_scale = 0.434f * c * kPi;
	const float expo = static_cast<float>(static_cast<double>(4.0f) - 2.0);
	for (int i = 0; i < 3; ++i) {
		const float total_mie = (std::pow((2.0f * kPi) / lambda[i], expo) * mie_scale) * K[i];
		fc->rayleigh[i] = rayleigh[i];
		fc->mie[i] = total_mie * 0.005f;
		fc->inv_total[i] = 1.0f / (fc->rayleigh[i] + fc->mie[i]);
	}
	const float m = std::pow(1.0f - dot3(sky_up, sun), 5.0f);
	fc->mixf = std::min(std::max(m, 0.0f), 1.0f);

	fc->width = fp->width; fc->height = fp->height;
	fc->spp = fp->spp; fc->sample_base = fp->sample_base; fc->max_bounces = fp->max_bounces;
	fc->base_frame = fp->base_frame; fc->flags = fp->flags;
	fc->band_rows = fp->band_rows; fc->shard_rank = fp->shard_rank; fc->shard_count = fp->shard_count;
	fc->local_rows = bm_local_rows(fp);
	fc->tiles_x = geo_tiles_x;
	fc->tiles_y = geo_tiles_y;
	// When does a wave stop to refill?  Every refill costs the whole wave an atomic's round trip and ~110 instructions, every idle
	// lane costs its share of all passes until then.  An item is all samples of a pixel (or ONE with BM_FLAG_SAMPLE_ITEMS): the
	// longer it is, the rarer the refills, the earlier they pay (measured per workload, profiles/r04_refill_sweep.txt).
	const int refill_override = tuning().refill_min;
	const int samples_per_item = (fp->flags & BM_FLAG_SAMPLE_ITEMS) ? 1 : fp->spp;
	fc->refill_min = samples_per_item >= 4 ? 4 : (samples_per_item >= 2 ? 8 : 16);
	fc->xcd_handout = geo_xcd;
	if (refill_override >= 1 && refill_override <= 64) fc->refill_min = refill_override;
	// shadow rays on helper lanes (trace.hip HELP): every frame that is not ordered (above)
	fc->helpers = ordered ? 0 : 1;
	const int help_override = tuning().helpers;
	if (help_override == 0 || (help_override == 1 && !ordered)) fc->helpers = help_override;
	// with helper lanes an idle lane is not wasted while it waits for the refill -- it takes shadow rays -- so the wave refills later:
	// 24 idle lanes instead of 16 (config 2 -0.2 %, 1080p at 4 spp -1.1 %, config 3 -0.8 %; 32: worse again; profiles/r05_refill_sweep.txt)
	if (fc->helpers && refill_override <= 0) fc->refill_min = 24;
	{ // divisions of the hand-out by per-frame constants (trace.hip refill): multiply-high + shift
		auto set = [](uint32_t d, uint32_t* magic, int* shift) { if (d <= 1u) { *magic = 0u; *shift = 0; } else division_magic(d, magic, shift); };
		set((fp->flags & BM_FLAG_SAMPLE_ITEMS) ? static_cast<uint32_t>(std::max(fp->spp, 1)) : 1u, &fc->div_samples_magic, &fc->div_samples_shift);
		set(static_cast<uint32_t>(fc->tiles_x), &fc->div_tiles_x_magic, &fc->div_tiles_x_shift);
		set(static_cast<uint32_t>(fc->band_rows), &fc->div_band_magic, &fc->div_band_shift);
		set(static_cast<uint32_t>((fc->tiles_x + 15) / 16), &fc->div_st_x_magic, &fc->div_st_x_shift);
	}
	return 0;
}

// ---------------------------------------------------------------- lifetime
Scene::~Scene() {
	hipSetDevice(device_);
	free_device();
	for (int r = 0; r < 2; ++r) {
		if (h_positions_[r]) hipHostFree(h_positions_[r]);
		if (h_count_[r]) hipHostFree(h_count_[r]);
		if (d_load_queue_[r]) hipFree(d_load_queue_[r]);
		if (d_load_count_[r]) hipFree(d_load_count_[r]);
	}
	if (ev_snapshot_) hipEventDestroy(ev_snapshot_);
	drop_frame_streams();
	if (h_bricks_) hipHostFree(h_bricks_);
	if (h_indices_) hipHostFree(h_indices_);
	if (h_moves_) hipHostFree(h_moves_);
	if (d_moves_) hipFree(d_moves_);
	if (d_bricks_queue_) hipFree(d_bricks_queue_);
	if (d_indices_queue_) hipFree(d_indices_queue_);
	if (d_counters_) hipFree(d_counters_);
	if (d_work_counter_) hipFree(d_work_counter_);
	if (d_frame_constants_) hipFree(d_frame_constants_);
	if (h_frame_constants_) hipHostFree(h_frame_constants_);
	for (int i = 0; i < kTimingRing; ++i) {
		if (ev_start_[i]) hipEventDestroy(ev_start_[i]);
		if (ev_stop_[i]) hipEventDestroy(ev_stop_[i]);
	}
	if (ev_upload_) hipEventDestroy(ev_upload_);
	if (load_stream_) hipStreamDestroy(load_stream_);
	if (kernel_stream_) hipStreamDestroy(kernel_stream_);
}

int Scene::init(int grid_size, int grid_height) {
	if (!world.dims.set(grid_size, grid_height)) {
		set_error("grid_size and grid_height must be positive multiples of 128 voxels");
		return BM_EINVAL;
	}
	if (world.dims.cells > 1024 || world.dims.cells_height > 1024) { // candidates take 24-bit products of brick coordinates and of their distance to the camera (traverse.h)
		set_error("world too large: at most 8192 voxels along an axis (and see the cube-field limit: cubic worlds up to 5760 voxels a side)");
		return BM_EINVAL;
	}
	// the walk keeps a ray's cell as ONE 32-bit byte offset into the 8 planes of the octant cube field, whose rows are padded to a
	// power of two (traverse.h cell_offset; allocate_device lays it out): 8 x (cells_h + 2) x (cells + 2) x 2^shift bytes < 4 GiB
	{
		int shift = 2;
		while ((1 << shift) < world.dims.cells + 2) ++shift;
		const uint64_t plane = (static_cast<uint64_t>(world.dims.cells + 2) << shift) * static_cast<uint64_t>(world.dims.cells_height + 2);
		if (plane * 8 >= (1ull << 32)) {
			set_error("world too large: the octant cube field (8 planes of (cells_h + 2) x (cells + 2) rows padded to a power of two) must stay below 4 GiB -- cubic worlds up to 5760 voxels a side");
			return BM_EINVAL;
		}
	}
	BM_HIP(hipSetDevice(device_));
	BM_HIP(hipStreamCreateWithFlags(&load_stream_, hipStreamNonBlocking));
	BM_HIP(hipStreamCreateWithFlags(&kernel_stream_, hipStreamNonBlocking));
	for (int i = 0; i < kTimingRing; ++i) {
		BM_HIP(hipEventCreate(&ev_start_[i]));
		BM_HIP(hipEventCreate(&ev_stop_[i]));
	}
	BM_HIP(hipEventCreateWithFlags(&ev_upload_, hipEventDisableTiming));
	BM_HIP(hipEventCreateWithFlags(&ev_snapshot_, hipEventDisableTiming));
	for (int r = 0; r < 2; ++r) {
		BM_HIP(hipHostMalloc(reinterpret_cast<void**>(&h_count_[r]), sizeof(uint32_t), hipHostMallocDefault));
		BM_HIP(hipMalloc(reinterpret_cast<void**>(&d_load_count_[r]), sizeof(uint32_t)));
		BM_HIP(hipMemset(d_load_count_[r], 0, sizeof(uint32_t)));
	}
	BM_HIP(hipMalloc(reinterpret_cast<void**>(&d_counters_), sizeof(DeviceCounters)));
	BM_HIP(hipMemset(d_counters_, 0, sizeof(DeviceCounters)));
	BM_HIP(hipMalloc(reinterpret_cast<void**>(&d_work_counter_), kWorkCounterBytes * kFrameRing)); // one block of counters per frame in flight
	BM_HIP(hipMalloc(reinterpret_cast<void**>(&d_frame_constants_), kFrameRing * sizeof(FrameConstants)));
	BM_HIP(hipHostMalloc(reinterpret_cast<void**>(&h_frame_constants_), kFrameRing * sizeof(FrameConstants), hipHostMallocDefault));
	std::fill(ring_owner_, ring_owner_ + kFrameRing, -1ll);
	hipDeviceProp_t prop;
	BM_HIP(hipGetDeviceProperties(&prop, device_));
	compute_units_ = prop.multiProcessorCount; // main.cpp:97 sm_cores
	blocks_per_cu_[0] = blocks_per_cu_[1] = 0; // no cap: every instantiation of the fused kernel runs at its own occupancy (trace.hip launch_trace)
	if (tuning().blocks_per_cu > 0) blocks_per_cu_[0] = tuning().blocks_per_cu; // experiment knob: fewer resident waves per SIMD (1 block = 1 wave per SIMD)

	return alloc_queue();
}

int Scene::alloc_queue() {
	BM_HIP(hipSetDevice(device_));
	if (h_bricks_) { hipHostFree(h_bricks_); h_bricks_ = nullptr; }
	if (h_indices_) { hipHostFree(h_indices_); h_indices_ = nullptr; }
	if (d_bricks_queue_) { hipFree(d_bricks_queue_); d_bricks_queue_ = nullptr; }
	if (d_indices_queue_) { hipFree(d_indices_queue_); d_indices_queue_ = nullptr; }
	const size_t n = static_cast<size_t>(queue_cap_);
	for (int r = 0; r < 2; ++r) {
		if (h_positions_[r]) { hipHostFree(h_positions_[r]); h_positions_[r] = nullptr; }
		if (d_load_queue_[r]) { hipFree(d_load_queue_[r]); d_load_queue_[r] = nullptr; }
		BM_HIP(hipHostMalloc(reinterpret_cast<void**>(&h_positions_[r]), n * 3 * sizeof(int), hipHostMallocDefault)); // Scene.cpp:30
		BM_HIP(hipMalloc(reinterpret_cast<void**>(&d_load_queue_[r]), n * 3 * sizeof(int)));                          // Scene.cpp:186
		BM_HIP(hipMemset(d_load_queue_[r], 0, n * 3 * sizeof(int)));
		BM_HIP(hipMemset(d_load_count_[r], 0, sizeof(uint32_t)));
	}
	BM_HIP(hipHostMalloc(reinterpret_cast<void**>(&h_bricks_), n * sizeof(Brick), hipHostMallocDefault));      // Scene.cpp:31
	BM_HIP(hipHostMalloc(reinterpret_cast<void**>(&h_indices_), n * sizeof(uint32_t), hipHostMallocDefault));  // Scene.cpp:32
	BM_HIP(hipMalloc(reinterpret_cast<void**>(&d_bricks_queue_), n * sizeof(Brick)));                          // Scene.cpp:189
	BM_HIP(hipMalloc(reinterpret_cast<void**>(&d_indices_queue_), n * sizeof(uint32_t)));                      // Scene.cpp:190
	// a batch of n requests can make at most n pools grow
	if (h_moves_) { hipHostFree(h_moves_); h_moves_ = nullptr; }
	if (d_moves_) { hipFree(d_moves_); d_moves_ = nullptr; }
	BM_HIP(hipHostMalloc(reinterpret_cast<void**>(&h_moves_), n * sizeof(PoolMove), hipHostMallocDefault));
	BM_HIP(hipMalloc(reinterpret_cast<void**>(&d_moves_), n * sizeof(PoolMove)));
	moves_cap_ = static_cast<uint32_t>(n);
	ring_cur_ = 0;
	snapshot_pending_ = false;
	view_.load_queue = d_load_queue_[0];
	view_.load_queue_count = d_load_count_[0];
	view_.queue_cap = static_cast<uint32_t>(queue_cap_);
	return 0;
}

int Scene::set_streaming_mode(int overlapped) {
	BM_HIP(hipSetDevice(device_));
	BM_HIP(hipDeviceSynchronize());
	if (snapshot_pending_) { // drain: the copied-out ring is serviced now, so no request is lost by switching modes
		snapshot_pending_ = false;
		const uint32_t count = std::min<uint32_t>(static_cast<uint32_t>(queue_cap_), *h_count_[ring_snapshot_]);
		if (count > 0) {
			if (int e = service_ring(ring_snapshot_, count)) return e;
		}
		BM_HIP(hipDeviceSynchronize());
	}
	if (overlapped_ && !overlapped && ring_cur_ != 0) {
		// the blocking mode always works on ring 0: carry over whatever the last frame queued in ring 1
		BM_HIP(hipMemcpy(d_load_queue_[0], d_load_queue_[1], static_cast<size_t>(queue_cap_) * 3 * sizeof(int), hipMemcpyDeviceToDevice));
		BM_HIP(hipMemcpy(d_load_count_[0], d_load_count_[1], sizeof(uint32_t), hipMemcpyDeviceToDevice));
		BM_HIP(hipMemset(d_load_count_[1], 0, sizeof(uint32_t)));
		ring_cur_ = 0;
		view_.load_queue = d_load_queue_[0];
		view_.load_queue_count = d_load_count_[0];
	}
	overlapped_ = overlapped != 0;
	return 0;
}

int Scene::set_lod(int lod8, int lod2) {
	lod8_ = lod8;
	lod2_ = lod2;
	view_.lod_distance_8x8x8 = lod8;
	view_.lod_distance_2x2x2 = lod2;
	return 0;
}

int Scene::set_queue_capacity(int cap) {
	if (cap <= 0) { set_error("queue capacity must be positive"); return BM_EINVAL; }
	// resizing the ring drops what it holds while the REQUESTED bits of those bricks stay set (they would never be asked
	// for again): the capacity belongs to the scene's construction, like the reference's constexpr (variables.h:35)
	if (on_device_) { set_error("set the queue capacity before bm_scene_generate"); return BM_ESTATE; }
	BM_HIP(hipSetDevice(device_));
	BM_HIP(hipDeviceSynchronize());
	queue_cap_ = cap;
	return alloc_queue();
}

void Scene::free_device() {
	if (d_index_grid_) hipFree(d_index_grid_);
	if (d_pool_base_) hipFree(d_pool_base_);
	arena_close();
	if (d_cube_field_) hipFree(d_cube_field_);
	d_cube_field_ = nullptr;
	d_index_grid_ = d_arena_ = d_pool_base_ = nullptr;
	arena_capacity_ = arena_top_ = pool_bricks_ = 0;
	on_device_ = false;
}

// ---------------------------------------------------------------- brick arena
void Scene::arena_reset() {
	arena_top_ = 0;
	pool_bricks_ = 0;
	for (auto& f : free_regions_) f.clear();
	freed_this_batch_.clear();
}

// ---- the arena's address range.  Reserved once per world for the worst case -- every pool is a power of two >= its
// supercell's brick count, and a pool that doubles its way up leaves regions of every smaller size behind (reused only by
// pools of that size) -- i.e. below 4 x the world's bricks + 32 per supercell; address space costs nothing.
int Scene::arena_open(uint64_t max_bricks) {
	arena_close();
	int vmm = 0;
	if (hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, device_) != hipSuccess) vmm = 0;
	if (const char* e = std::getenv("BM_ARENA_VMM")) vmm = vmm && std::atoi(e) != 0; // experiment knob: 0 = reallocate + copy
	if (!vmm) { (void)hipGetLastError(); arena_virtual_ = false; return 0; }
	hipMemAllocationProp prop{};
	prop.type = hipMemAllocationTypePinned;
	prop.location.type = hipMemLocationTypeDevice;
	prop.location.id = device_;
	size_t gran = 0;
	BM_HIP(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
	if (gran == 0) gran = 2u << 20;
	arena_granularity_ = gran;
	uint64_t bytes = std::min<uint64_t>(max_bricks, (1ull << 32) - 1) * sizeof(Brick);
	bytes = (std::max<uint64_t>(bytes, 1ull << 22) + gran - 1) / gran * gran;
	void* va = nullptr;
	BM_HIP(hipMemAddressReserve(&va, bytes, 0, nullptr, 0));
	d_arena_ = static_cast<uint32_t*>(va);
	arena_va_bytes_ = bytes;
	arena_virtual_ = true;
	arena_capacity_ = 0;
	view_.brick_arena = d_arena_;
	return 0;
}

int Scene::arena_unmap_all() {
	// every chunk is taken off the list as it is processed (a chunk that failed to unmap must not be unmapped and released a
	// second time by a later call); the first error is reported after all of them have been tried
	hipError_t first = hipSuccess;
	const char* what = "";
	while (!arena_chunks_.empty()) {
		const ArenaChunk c = arena_chunks_.back();
		arena_chunks_.pop_back();
		if (hipError_t e = hipMemUnmap(reinterpret_cast<char*>(d_arena_) + c.offset, c.bytes); e != hipSuccess && first == hipSuccess) { first = e; what = "hipMemUnmap"; }
		if (hipError_t e = hipMemRelease(c.handle); e != hipSuccess && first == hipSuccess) { first = e; what = "hipMemRelease"; }
	}
	arena_capacity_ = 0;
	if (first != hipSuccess) return hip_fail(first, what, __FILE__, __LINE__);
	return 0;
}

void Scene::arena_close() {
	if (arena_virtual_) {
		// unlike hipFree, unmapping does not wait for work that still uses the range
		if (!arena_chunks_.empty()) (void)hipDeviceSynchronize();
		const int unmap_error = arena_unmap_all();
		// (a range that may still hold a mapping is not handed back: leaking address space is harmless, freeing a mapped range is not)
		if (d_arena_ && unmap_error == 0) (void)hipMemAddressFree(d_arena_, arena_va_bytes_);
	} else if (d_arena_) {
		(void)hipFree(d_arena_);
	}
	d_arena_ = nullptr;
	arena_virtual_ = false;
	arena_va_bytes_ = 0;
	arena_capacity_ = 0;
	arena_chunks_.clear();
}

// Make the arena at least `bricks` large, keeping what it holds.  Virtual arena: map one more physical chunk behind the
// mapped part (the mapped size doubles) -- no copy, no synchronisation, nothing moves, so frames in flight and upload
// batches already queued are not disturbed (the reference grows one pool at a time with a blocking cudaMemcpy,
// Scene.cpp:242-247).  exact: (re)size an EMPTY arena to fit a known residency (callers have synchronised the device).
int Scene::arena_reserve(uint64_t bricks, bool exact) {
	if (bricks <= arena_capacity_ && !(exact && arena_top_ == 0 && arena_capacity_ > 2 * std::max<uint64_t>(bricks, 1ull << 16))) return 0;
	if (bricks >= (1ull << 32)) { set_error("brick arena would exceed 2^32 bricks"); return BM_EINVAL; }
	if (arena_virtual_) {
		const size_t gran = arena_granularity_;
		auto round_up = [gran](uint64_t b) { return (b + gran - 1) / gran * gran; };
		uint64_t want_bytes;
		bool remapping_empty_arena = false;
		// (when the arena was unmapped for an exact re-size, any failure below marks the scene failed: frames are refused until
		// bm_scene_reset_residency / bm_scene_preload_all succeeds -- the device index words may still carry loaded bits)
		auto fail = [&](int code) { if (remapping_empty_arena) failed_ = true; return code; };
		if (exact && arena_top_ == 0) {
			if (!arena_chunks_.empty()) BM_HIP(hipDeviceSynchronize()); // (callers have synchronised already; unmapping itself does not wait)
			if (int e = arena_unmap_all()) { failed_ = true; return e; }
			remapping_empty_arena = true; // from here on a failure leaves view_.brick_arena pointing at an unmapped range
			want_bytes = round_up(std::max<uint64_t>(bricks, 1ull << 16) * sizeof(Brick));
		} else {
			want_bytes = round_up(std::max<uint64_t>(arena_capacity_, 1ull << 16) * sizeof(Brick)); // 4 MiB to start with
			while (want_bytes < bricks * sizeof(Brick)) want_bytes *= 2;
		}
		if (want_bytes > arena_va_bytes_) want_bytes = arena_va_bytes_;
		if (want_bytes < bricks * sizeof(Brick)) { set_error("brick arena: reserved address range exhausted"); return fail(BM_ESTATE); }
		const size_t have = static_cast<size_t>(arena_capacity_) * sizeof(Brick);
		if (want_bytes > have) {
			hipMemAllocationProp prop{};
			prop.type = hipMemAllocationTypePinned;
			prop.location.type = hipMemLocationTypeDevice;
			prop.location.id = device_;
			ArenaChunk c{};
			c.offset = have;
			c.bytes = want_bytes - have;
			if (hipError_t e = hipMemCreate(&c.handle, c.bytes, &prop, 0); e != hipSuccess) return fail(hip_fail(e, "hipMemCreate", __FILE__, __LINE__));
			char* at = reinterpret_cast<char*>(d_arena_) + c.offset;
			if (hipError_t e = hipMemMap(at, c.bytes, 0, c.handle, 0); e != hipSuccess) { (void)hipMemRelease(c.handle); return fail(hip_fail(e, "hipMemMap", __FILE__, __LINE__)); }
			hipMemAccessDesc access{};
			access.location = prop.location;
			access.flags = hipMemAccessFlagsProtReadWrite;
			// access is (re)declared for the WHOLE mapped range, from the base: on this runtime (ROCm 7.2) hipMemSetAccess on a
			// chunk at an offset fails sporadically with "invalid argument" when the chunks differ in size
			// (tools/ubench/vmm_probe2.hip: 33 of 144 growths; 0 of 144 this way, with kernels in flight over the range)
			if (hipError_t e = hipMemSetAccess(d_arena_, want_bytes, &access, 1); e != hipSuccess) {
				(void)hipMemUnmap(at, c.bytes); (void)hipMemRelease(c.handle);
				return fail(hip_fail(e, "hipMemSetAccess", __FILE__, __LINE__));
			}
			arena_chunks_.push_back(c);
			if (arena_capacity_ > 0) arena_growths_++;
			arena_capacity_ = want_bytes / sizeof(Brick);
		}
		return 0;
	}
	// ---- no virtual memory management on this device: reallocate + copy.  Synchronises the device: frames in flight may
	// still read the old allocation, and the copy must see every upload.
	uint64_t cap = bricks;
	if (!exact) { // growth by residency: double
		cap = std::max<uint64_t>(arena_capacity_, 1ull << 16); // 4 MiB to start with
		while (cap < bricks) cap *= 2;
	} else if (arena_top_ == 0) {
		cap = std::max<uint64_t>(bricks, 1ull << 16); // (re)sized for a known residency: exact fit, shrinking an oversized arena
	}
	if (cap >= (1ull << 32)) { set_error("brick arena would exceed 2^32 bricks"); return BM_EINVAL; }
	BM_HIP(hipDeviceSynchronize());
	uint32_t* fresh = nullptr;
	BM_HIP(hipMalloc(reinterpret_cast<void**>(&fresh), cap * sizeof(Brick)));
	if (d_arena_ && arena_top_ > 0) {
		BM_HIP(hipMemcpy(fresh, d_arena_, arena_top_ * sizeof(Brick), hipMemcpyDeviceToDevice));
		arena_growths_++; arena_copy_growths_++;
	}
	if (d_arena_) BM_HIP(hipFree(d_arena_));
	d_arena_ = fresh;
	arena_capacity_ = cap;
	view_.brick_arena = d_arena_;
	return 0;
}

// A region of `bricks` (a power of two >= kStartingPool) for one pool: from the free list of that size, else from the top.
int Scene::region_alloc(uint32_t bricks, uint32_t* offset) {
	int cls = 0;
	while ((1u << cls) < bricks) ++cls;
	if (!free_regions_[cls].empty()) {
		*offset = free_regions_[cls].back();
		free_regions_[cls].pop_back();
	} else {
		if (int e = arena_reserve(arena_top_ + bricks)) return e;
		*offset = static_cast<uint32_t>(arena_top_);
		arena_top_ += bricks;
	}
	pool_bricks_ += bricks;
	return 0;
}

// A vacated region becomes reusable once the batch that vacates it has been queued: its move kernel still reads it, and
// a pool growing in the SAME batch must not be given it (later batches are ordered behind this one on the load stream).
void Scene::region_free_deferred(uint32_t bricks, uint32_t offset) {
	int cls = 0;
	while ((1u << cls) < bricks) ++cls;
	freed_this_batch_.emplace_back(cls, offset);
	pool_bricks_ -= bricks;
}

// floor(n / d) for every n < 2^30 as umulhi(n, magic) >> shift (Granlund-Montgomery: with l = ceil(log2 d) and
// magic = ceil(2^(30 + l) / d) one has 2^(30+l) <= magic * d < 2^(30+l) + 2^l, which makes the truncated product exact for 30-bit n;
// magic < 2^31 + 1 fits 32 bits; tests/test_host_logic.py replays it against integer division)
void division_magic(uint32_t d, uint32_t* magic, int* shift) {
	int l = 0;
	while ((1ull << l) < d) ++l;
	if (l < 2) l = 2;
	const unsigned __int128 one = static_cast<unsigned __int128>(1) << (30 + l);
	*magic = static_cast<uint32_t>((one + d - 1) / d);
	*shift = l - 2; // (30 + l) - 32
}

// device half of Scene::generate (Scene.cpp:152-190): one flat index grid, one pool-base word per supercell and one
// brick arena instead of 2 x supercells cudaMallocs and two pointer tables; plus the octant cube field of the walk.
int Scene::allocate_device() {
	BM_HIP(hipSetDevice(device_));
	free_device();
	const WorldDims& d = world.dims;
	uint64_t run = 0;
	for (int i = 0; i < d.supercells; ++i) run += world.supercells[i].bricks.size();
	if (run >= (1ull << 32)) { set_error("world has more than 2^32 bricks"); return BM_EINVAL; }
	total_bricks_ = run;
	const size_t index_bytes = static_cast<size_t>(d.supercells) * kCellsPerSupercell * sizeof(uint32_t);
	BM_HIP(hipMalloc(reinterpret_cast<void**>(&d_index_grid_), index_bytes));
	BM_HIP(hipMalloc(reinterpret_cast<void**>(&d_pool_base_), static_cast<size_t>(d.supercells) * sizeof(uint32_t)));
	BM_HIP(hipMemset(d_pool_base_, 0, static_cast<size_t>(d.supercells) * sizeof(uint32_t)));
	view_.index_grid = d_index_grid_;
	view_.pool_base = d_pool_base_;
	view_.brick_arena = nullptr;
	if (int e = arena_open(4 * run + 32ull * static_cast<uint64_t>(d.supercells) + (1ull << 16))) return e;
	{ // octant cube field: what the walk reads instead of index words while it crosses empty space.  Device layout: rows padded to a
	  // power of two, so that a cell's entry offset -- which is what a ray carries as its position (traverse.h cell_offset) -- moves by
	  // +-1 / +- 2^shift / +- slice pitch; the host builds the field with tight rows (world.cpp) and every slice is copied row by row.
		std::vector<uint8_t> field;
		world.build_cube_field(field, 8);
		const int X = d.cells + 2, Z = d.cells_height + 2;
		int shift = 2;
		while ((1 << shift) < X) ++shift;
		const uint64_t pxy = static_cast<uint64_t>(X) << shift, plane = pxy * static_cast<uint64_t>(Z);
		if (pxy >= (1ull << 23) || plane * 8 >= (1ull << 32)) { set_error("world too large for the 32-bit cube-field offsets of the walk"); return BM_EINVAL; }
		BM_HIP(hipMalloc(reinterpret_cast<void**>(&d_cube_field_), plane * 8));
		BM_HIP(hipMemset(d_cube_field_, 255, plane * 8)); // the row padding reads as border cells: a stray offset ends a walk instead of reading whatever was there
		// 8 planes x Z slices, each X rows of X bytes -> rows of 2^shift bytes (the padding is never read)
		for (int o = 0; o < 8; ++o)
			BM_HIP(hipMemcpy2D(d_cube_field_ + static_cast<size_t>(o) * plane, static_cast<size_t>(1) << shift, field.data() + static_cast<size_t>(o) * X * X * Z, X, X,
							   static_cast<size_t>(X) * Z, hipMemcpyHostToDevice));
		view_.cf_shift = shift;
		view_.cf_pxy = static_cast<uint32_t>(pxy);
		view_.cf_plane = static_cast<uint32_t>(plane);
		division_magic(view_.cf_pxy, &view_.cf_magic, &view_.cf_magic_shift);
		view_.cube_field = d_cube_field_;
		cube_field_bytes_ = plane * 8;
	}
	view_.cells = d.cells;
	view_.cells_height = d.cells_height;
	view_.sg_xy = d.supergrid_xy;
	view_.sg_xy2 = d.supergrid_xy * d.supergrid_xy;
	view_.grid_size_f = static_cast<float>(d.grid_size);
	view_.grid_height_f = static_cast<float>(d.grid_height);
	view_.lod_distance_8x8x8 = lod8_;
	view_.lod_distance_2x2x2 = lod2_;
	on_device_ = true;
	return reset_residency();
}

int Scene::generate(int threads) {
	world.generate(threads);
	return allocate_device();
}

int Scene::generate_supercell(int sx, int sy, int sz) {
	const WorldDims& d = world.dims;
	if (sx < 0 || sy < 0 || sz < 0 || sx >= d.supergrid_xy || sy >= d.supergrid_xy || sz >= d.supergrid_z) {
		set_error("supercell coordinates out of range");
		return BM_EINVAL;
	}
	// Once the world is on the device its pools hold bricks in request order and the index words name those slots:
	// rebuilding a host supercell would reset its slot counter under them (the regenerated content is identical anyway --
	// the terrain is a pure function of the coordinates).  The reference never calls it after generate() either.
	if (on_device_) { set_error("bm_scene_generate_supercell: the scene is on the device (call it before bm_scene_generate)"); return BM_ESTATE; }
	world.generate_supercell(sx, sy, sz);
	return 0;
}

// reference initial state: every non-empty brick is "unloaded | lod", nothing resident (Scene.cpp:157-175)
int Scene::reset_residency() {
	if (!on_device_) { set_error("scene not generated"); return BM_ESTATE; }
	BM_HIP(hipSetDevice(device_));
	BM_HIP(hipDeviceSynchronize());
	const WorldDims& d = world.dims;
	std::vector<uint32_t> words(static_cast<size_t>(d.supercells) * kCellsPerSupercell);
	std::vector<uint32_t> bases(d.supercells, 0u);
	// every supercell that holds bricks starts with a pool of kStartingPool bricks (Scene.cpp:157-175, variables.h:15)
	arena_reset();
	uint64_t initial = 0;
	for (int i = 0; i < d.supercells; ++i) initial += world.supercells[i].bricks.empty() ? 0u : kStartingPool;
	if (int e = arena_reserve(std::max<uint64_t>(initial, 1), true)) return e;
	for (int i = 0; i < d.supercells; ++i) {
		HostSupercell& c = world.supercells[i];
		c.resident = 0;
		c.pool_capacity = 0;
		c.pool_base = 0;
		if (!c.bricks.empty()) {
			if (int e = region_alloc(kStartingPool, &c.pool_base)) return e;
			c.pool_capacity = kStartingPool;
		}
		bases[i] = c.pool_base;
		uint32_t* dst = &words[static_cast<size_t>(i) * kCellsPerSupercell];
		for (int j = 0; j < kCellsPerSupercell; ++j)
			dst[j] = (c.indices[j] & BM_BRICK_LOADED_BIT) ? (BM_BRICK_UNLOADED_BIT | (c.indices[j] & BM_BRICK_LOD_BITS)) : 0u;
	}
	BM_HIP(hipMemcpy(d_index_grid_, words.data(), words.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
	BM_HIP(hipMemcpy(d_pool_base_, bases.data(), bases.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
	for (int r = 0; r < 2; ++r) BM_HIP(hipMemset(d_load_count_[r], 0, sizeof(uint32_t)));
	ring_cur_ = 0;
	snapshot_pending_ = false;
	view_.load_queue = d_load_queue_[0];
	view_.load_queue_count = d_load_count_[0];
	resident_bricks_ = 0;
	staging_busy_ = false;
	failed_ = false;
	stream_batches_ = stream_host_ns_ = 0;
	upload_seq_ = 0;
	for (FrameStream& f : frame_streams_) f.upload_seen = 0;
	return 0;
}

int Scene::preload_all() {
	if (!on_device_) { set_error("scene not generated"); return BM_ESTATE; }
	BM_HIP(hipSetDevice(device_));
	BM_HIP(hipDeviceSynchronize());
	const WorldDims& d = world.dims;
	std::vector<uint32_t> words(static_cast<size_t>(d.supercells) * kCellsPerSupercell);
	std::vector<uint32_t> bases(d.supercells, 0u);
	// "all bricks pre-loaded" (BASELINE configs 1-2): every pool is its supercell's full host brick vector, exact fit, and
	// the device words are the host words (slot | loaded | lod, Scene.cpp:104)
	arena_reset();
	if (int e = arena_reserve(std::max<uint64_t>(total_bricks_, 1), true)) return e;
	for (int i = 0; i < d.supercells; ++i) {
		HostSupercell& c = world.supercells[i];
		c.pool_base = static_cast<uint32_t>(arena_top_);
		c.pool_capacity = static_cast<uint32_t>(c.bricks.size());
		arena_top_ += c.bricks.size();
		bases[i] = c.pool_base;
		std::memcpy(&words[static_cast<size_t>(i) * kCellsPerSupercell], c.indices.data(), kCellsPerSupercell * sizeof(uint32_t));
		c.resident = static_cast<uint32_t>(c.bricks.size());
		if (!c.bricks.empty())
			BM_HIP(hipMemcpy(d_arena_ + static_cast<size_t>(c.pool_base) * kBrickWords, c.bricks.data(), c.bricks.size() * sizeof(Brick), hipMemcpyHostToDevice));
	}
	pool_bricks_ = total_bricks_;
	BM_HIP(hipMemcpy(d_pool_base_, bases.data(), bases.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
	BM_HIP(hipMemcpyAsync(d_index_grid_, words.data(), words.size() * sizeof(uint32_t), hipMemcpyHostToDevice, load_stream_));
	for (int r = 0; r < 2; ++r) BM_HIP(hipMemsetAsync(d_load_count_[r], 0, sizeof(uint32_t), load_stream_));
	BM_HIP(hipStreamSynchronize(load_stream_));
	ring_cur_ = 0;
	snapshot_pending_ = false;
	view_.load_queue = d_load_queue_[0];
	view_.load_queue_count = d_load_count_[0];
	resident_bricks_ = total_bricks_;
	staging_busy_ = false;
	failed_ = false;
	stream_batches_ = stream_host_ns_ = 0;
	upload_seq_ = 0;
	for (FrameStream& f : frame_streams_) f.upload_seen = 0;
	return 0;
}

// ---------------------------------------------------------------- frames vs. uploads
// A frame on `stream` must see every brick batch queued so far: wait for the latest upload event unless this stream
// already has.  (Per stream, not per scene: with frames on several streams each of them has to be ordered.)
int Scene::frame_begin(hipStream_t stream) {
	if (failed_) { set_error("a streaming batch failed on this scene: call bm_scene_reset_residency / bm_scene_preload_all"); return BM_ESTATE; }
	FrameStream* fs = nullptr;
	bool created = false;
	for (FrameStream& f : frame_streams_) if (f.stream == stream) { fs = &f; break; }
	if (!fs) {
		created = true;
		if (frame_streams_.size() >= kMaxFrameStreams) {
			// retire the entry that has been idle longest (a caller that keeps creating streams): its last frame must have
			// finished before the entry -- and with it the ordering of the load stream behind that frame -- can go
			size_t lru = 0;
			for (size_t i = 1; i < frame_streams_.size(); ++i) if (frame_streams_[i].last_use < frame_streams_[lru].last_use) lru = i;
			BM_HIP(hipEventSynchronize(frame_streams_[lru].done));
			BM_HIP(hipEventDestroy(frame_streams_[lru].done));
			frame_streams_.erase(frame_streams_.begin() + static_cast<long>(lru));
		}
		FrameStream f;
		f.stream = stream;
		BM_HIP(hipEventCreateWithFlags(&f.done, hipEventDisableTiming));
		frame_streams_.push_back(f);
		fs = &frame_streams_.back();
	}
	// (a stream handle can be recycled by the runtime after its owner destroyed it: an entry that claims to have seen the
	// latest batch is only trusted once that batch has actually completed)
	// (the event is only queried for an entry that claims to be up to date and was not made by this call; a stream that has
	// waited is not asked to wait again, and only the query's own "not ready" status is cleared -- never a sticky error of the
	// caller's that happens to be pending on this thread)
	bool must_wait = fs->upload_seen < upload_seq_;
	if (!must_wait && upload_seq_ > 0 && !created) {
		const hipError_t q = hipEventQuery(ev_upload_);
		if (q == hipErrorNotReady) { (void)hipGetLastError(); must_wait = true; }
	}
	if (must_wait) {
		BM_HIP(hipStreamWaitEvent(stream, ev_upload_, 0)); // ev_upload_ is re-recorded behind every batch: waiting for it covers all earlier ones
		fs->upload_seen = upload_seq_;
	}
	fs->last_use = ++frame_seq_;
	return 0;
}

int Scene::frame_end(hipStream_t stream) {
	for (FrameStream& f : frame_streams_)
		if (f.stream == stream) { BM_HIP(hipEventRecord(f.done, stream)); return 0; }
	set_error("frame_end without frame_begin");
	return BM_ESTATE;
}

int Scene::wait_frames_on_host() {
	for (FrameStream& f : frame_streams_) BM_HIP(hipEventSynchronize(f.done));
	return 0;
}

int Scene::order_load_stream_behind_frames() {
	for (FrameStream& f : frame_streams_) BM_HIP(hipStreamWaitEvent(load_stream_, f.done, 0));
	return 0;
}

void Scene::drop_frame_streams() {
	for (FrameStream& f : frame_streams_)
		if (f.done) (void)hipEventDestroy(f.done);
	frame_streams_.clear();
}

// ---------------------------------------------------------------- streaming
// Stage the first `count` requests of a ring (positions already in its pinned mirror), copy them up and scatter
// them into the arena / index grid on the load stream (Scene.cpp:215-229 + the upload kernel, kernel.cu:141-151,412-413).
int Scene::service_ring(int ring, uint32_t count) {
	const auto t_host0 = std::chrono::steady_clock::now();
	const WorldDims& d = world.dims;
	const int* pos = h_positions_[ring];
	// ---- pass 1: nothing is mutated before every entry has been checked (the positions come back from device memory:
	// never index host arrays with an entry that cannot be a request)
	for (uint32_t i = 0; i < count; ++i) {
		const int px = pos[3 * i], py = pos[3 * i + 1], pz = pos[3 * i + 2];
		if (px < 0 || py < 0 || pz < 0 || px >= d.cells || py >= d.cells || pz >= d.cells_height) {
			// (the ring's counter is still set and the requested bits still stand: every later call would fail the same way while
			// frames went on as if nothing had happened -- mark the scene failed, recovery is a residency reset)
			set_error("brick request ring holds a position outside the world");
			failed_ = true;
			return BM_ESTATE;
		}
		const HostSupercell& c = world.supercells[d.supercell_id(px / kSupercell, py / kSupercell, pz / kSupercell)];
		const uint32_t word = c.indices[static_cast<uint32_t>((px % kSupercell) + (py % kSupercell) * kSupercell + (pz % kSupercell) * kSupercell * kSupercell)];
		if (!(word & BM_BRICK_LOADED_BIT) || (word & BM_BRICK_INDEX_BITS) >= c.bricks.size()) {
			set_error("brick request ring names an empty brick");
			failed_ = true;
			return BM_ESTATE;
		}
	}
	if (staging_busy_) { // the staging buffers of the previous upload are free again once its copies have run
		BM_HIP(hipEventSynchronize(ev_upload_));
		staging_busy_ = false;
	}
	// ---- pass 2: hand out slots, grow pools.  From here on host state changes entry by entry; the only thing that can
	// still go wrong is running out of device memory while the arena grows, and that leaves the scene marked as failed
	// (every later frame / batch is refused until the residency is reset) instead of half-updated and in use.
	uint32_t n_moves = 0;
	std::unordered_map<int, uint32_t> batch_first_resident, batch_move; // per supercell: bricks resident before this batch / its entry in h_moves_
	for (uint32_t i = 0; i < count; ++i) {
		const int px = pos[3 * i], py = pos[3 * i + 1], pz = pos[3 * i + 2];
		const int sci = d.supercell_id(px / kSupercell, py / kSupercell, pz / kSupercell);
		HostSupercell& c = world.supercells[sci];
		const uint32_t local = static_cast<uint32_t>((px % kSupercell) + (py % kSupercell) * kSupercell + (pz % kSupercell) * kSupercell * kSupercell);
		const uint32_t word = c.indices[local];
		std::memcpy(h_bricks_ + static_cast<size_t>(i) * kBrickWords, c.bricks[word & BM_BRICK_INDEX_BITS].data, sizeof(Brick));
		// slots are handed out in request order (gpu_index_highest++, Scene.cpp:224); a full pool doubles first
		// (Scene.cpp:231-251: 2^ceil(log2(highest + 1))) -- here it moves to a larger region of the arena
		const uint32_t resident_before_batch = batch_first_resident.emplace(sci, c.resident).first->second;
		if (c.resident >= c.pool_capacity) {
			const uint32_t grown = std::max<uint32_t>(kStartingPool, c.pool_capacity * 2u);
			uint32_t fresh = 0;
			if (int e = region_alloc(grown, &fresh)) { failed_ = true; return e; }
			// Only the bricks that were resident BEFORE this batch have to be copied (the batch's own bricks are scattered to
			// base + slot after the bases are published), and only once: a pool that grows twice in one batch moves from
			// the region it had when the batch began straight to the last one.
			auto mv = batch_move.find(sci);
			if (mv == batch_move.end()) {
				batch_move.emplace(sci, n_moves);
				h_moves_[n_moves++] = PoolMove{c.pool_base, fresh, resident_before_batch, static_cast<uint32_t>(sci)};
			} else {
				h_moves_[mv->second].dst = fresh;
			}
			if (c.pool_capacity > 0) region_free_deferred(c.pool_capacity, c.pool_base);
			c.pool_base = fresh;
			c.pool_capacity = grown;
		}
		h_indices_[i] = c.resident | BM_BRICK_LOADED_BIT | (word & BM_BRICK_LOD_BITS);
		c.resident++;
	}
	auto queue = [&]() -> int {
		BM_HIP(hipMemcpyAsync(d_bricks_queue_, h_bricks_, static_cast<size_t>(count) * sizeof(Brick), hipMemcpyHostToDevice, load_stream_));    // :228
		BM_HIP(hipMemcpyAsync(d_indices_queue_, h_indices_, static_cast<size_t>(count) * sizeof(uint32_t), hipMemcpyHostToDevice, load_stream_)); // :229
		// The scatter kernel rewrites index words that a frame still in flight may be reading and requesting through
		// (plain load + atomicOr): a word flipping to "loaded" between the two would be requested a second time; the move
		// kernel rewrites pool bases such a frame addresses bricks with.  In overlapped mode both therefore run behind every
		// frame in flight, whatever stream it is on; the copies above already overlap them.
		if (overlapped_) { if (int e = order_load_stream_behind_frames()) return e; }
		DeviceScene ring_view = view_;
		ring_view.load_queue = d_load_queue_[ring];
		ring_view.load_queue_count = d_load_count_[ring];
		if (n_moves > 0) { // grown pools: copy their bricks to the new regions and publish the new bases, ahead of the scatter
			BM_HIP(hipMemcpyAsync(d_moves_, h_moves_, static_cast<size_t>(n_moves) * sizeof(PoolMove), hipMemcpyHostToDevice, load_stream_));
			launch_pool_moves(d_moves_, n_moves, d_arena_, d_pool_base_, load_stream_);
			BM_HIP(hipGetLastError());
		}
		launch_upload(ring_view, d_bricks_queue_, d_indices_queue_, d_arena_, count, load_stream_); // kernel.cu:412
		BM_HIP(hipGetLastError());
		BM_HIP(hipMemsetAsync(d_load_count_[ring], 0, sizeof(uint32_t), load_stream_));             // kernel.cu:413
		BM_HIP(hipEventRecord(ev_upload_, load_stream_));
		return 0;
	};
	if (int e = queue()) { failed_ = true; return e; } // the host bookkeeping is ahead of the device: refuse to go on
	for (const auto& f : freed_this_batch_) free_regions_[f.first].push_back(f.second); // reusable by the NEXT batch
	freed_this_batch_.clear();
	staging_busy_ = true;
	upload_seq_++;
	resident_bricks_ += count;
	stream_batches_++;
	stream_host_ns_ += static_cast<uint64_t>(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_host0).count());
	return 0;
}

int Scene::process_load_queue(uint32_t* serviced) {
	if (serviced) *serviced = 0;
	if (!on_device_) { set_error("scene not generated"); return BM_ESTATE; }
	if (failed_) { set_error("a streaming batch failed on this scene: call bm_scene_reset_residency / bm_scene_preload_all"); return BM_ESTATE; }
	BM_HIP(hipSetDevice(device_));
	if (!overlapped_) {
		// ---- reference order (main.cpp:142-144): the frames that raised the requests have finished (kernel.cu:431), the host
		// reads the ring, stages, uploads; the next frame sees the bricks
		if (int e = wait_frames_on_host()) return e; // every stream a frame was issued on
		BM_HIP(hipMemcpyAsync(h_count_[0], d_load_count_[0], sizeof(uint32_t), hipMemcpyDeviceToHost, load_stream_)); // Scene.cpp:202
		BM_HIP(hipStreamSynchronize(load_stream_));
		const uint32_t count = std::min<uint32_t>(static_cast<uint32_t>(queue_cap_), *h_count_[0]);                   // Scene.cpp:203
		if (count == 0) return 0;
		BM_HIP(hipMemcpyAsync(h_positions_[0], d_load_queue_[0], static_cast<size_t>(count) * 3 * sizeof(int), hipMemcpyDeviceToHost, load_stream_)); // :209
		BM_HIP(hipStreamSynchronize(load_stream_));
		if (int e = service_ring(0, count)) return e;
		if (serviced) *serviced = count;
		return 0;
	}
	// ---- overlapped mode: never wait for the GPU.  (1) service the ring that was copied out by the previous call,
	// (2) start copying out the ring the last frame wrote, behind that frame, on the load stream, (3) hand the other
	// ring to the next frame.  A brick requested in frame k is resident from frame k+2 on (reference order: from k+1 on).
	if (snapshot_pending_) {
		BM_HIP(hipEventSynchronize(ev_snapshot_));
		snapshot_pending_ = false;
		const uint32_t count = std::min<uint32_t>(static_cast<uint32_t>(queue_cap_), *h_count_[ring_snapshot_]);
		if (count > 0) {
			if (int e = service_ring(ring_snapshot_, count)) return e;
			if (serviced) *serviced = count;
		}
	}
	if (int e = order_load_stream_behind_frames()) return e; // the ring is complete once every frame that may append to it has ended
	const int ring = ring_cur_;
	// the count is not known on the host without waiting for the frame: a small kernel copies count + the entries that exist into
	// the pinned (device-mapped) mirrors -- min(count, capacity) x 12 bytes over PCIe instead of the ring's whole capacity
	{
		int* dev_positions = nullptr;
		uint32_t* dev_count = nullptr;
		BM_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&dev_positions), h_positions_[ring], 0));
		BM_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&dev_count), h_count_[ring], 0));
		launch_snapshot_ring(d_load_queue_[ring], d_load_count_[ring], dev_positions, dev_count, static_cast<uint32_t>(queue_cap_), load_stream_);
		BM_HIP(hipGetLastError());
	}
	BM_HIP(hipEventRecord(ev_snapshot_, load_stream_));
	snapshot_pending_ = true;
	ring_snapshot_ = ring;
	ring_cur_ = ring ^ 1;
	view_.load_queue = d_load_queue_[ring_cur_];
	view_.load_queue_count = d_load_count_[ring_cur_];
	return 0;
}

int Scene::dump(const char* path) { // Scene.cpp:254-258
	std::ofstream file(path ? path : "dump.txt");
	if (!file) { set_error("cannot open dump file"); return BM_EINVAL; }
	for (const auto& c : world.supercells) file << c.resident << "\n";
	return 0;
}

int Scene::info(bm_scene_info* out) {
	if (!out) { set_error("null argument"); return BM_EINVAL; }
	const WorldDims& d = world.dims;
	std::memset(out, 0, sizeof *out);
	out->grid_size = d.grid_size; out->grid_height = d.grid_height;
	out->supergrid_xy = d.supergrid_xy; out->supergrid_z = d.supergrid_z; out->supercells = d.supercells;
	out->queue_capacity = queue_cap_;
	out->lod_distance_8x8x8 = lod8_; out->lod_distance_2x2x2 = lod2_;
	out->generated = world.generated ? 1 : 0;
	out->on_device = on_device_ ? 1 : 0;
	out->total_bricks = world.generated ? world.total_bricks() : 0;
	out->resident_bricks = resident_bricks_;
	out->index_bytes = on_device_ ? static_cast<uint64_t>(d.supercells) * kCellsPerSupercell * 4 : 0;
	out->brick_bytes = on_device_ ? arena_capacity_ * 64 : 0;
	out->pool_bytes = on_device_ ? pool_bricks_ * 64 : 0;
	out->cube_field_bytes = on_device_ ? cube_field_bytes_ : 0;
	out->arena_growths = arena_growths_;
	out->arena_copy_growths = arena_copy_growths_;
	out->arena_virtual = arena_virtual_ ? 1 : 0;
	out->failed = failed_ ? 1 : 0;
	out->stream_batches = stream_batches_;
	out->stream_host_ns = stream_host_ns_;
	return 0;
}

int Scene::device_indices(int supercell, uint32_t* out4096) {
	if (!on_device_) { set_error("scene not generated"); return BM_ESTATE; }
	if (supercell < 0 || supercell >= world.dims.supercells || !out4096) { set_error("bad supercell"); return BM_EINVAL; }
	BM_HIP(hipSetDevice(device_));
	BM_HIP(hipDeviceSynchronize());
	BM_HIP(hipMemcpy(out4096, d_index_grid_ + static_cast<size_t>(supercell) * kCellsPerSupercell, kCellsPerSupercell * 4, hipMemcpyDeviceToHost));
	return 0;
}

int Scene::device_brick(int supercell, uint32_t device_slot, uint32_t* out16) {
	if (!on_device_) { set_error("scene not generated"); return BM_ESTATE; }
	if (supercell < 0 || supercell >= world.dims.supercells || !out16 || device_slot >= world.supercells[supercell].resident) {
		set_error("bad supercell or slot");
		return BM_EINVAL;
	}
	BM_HIP(hipSetDevice(device_));
	BM_HIP(hipDeviceSynchronize());
	BM_HIP(hipMemcpy(out16, d_arena_ + (static_cast<size_t>(world.supercells[supercell].pool_base) + device_slot) * kBrickWords, sizeof(Brick), hipMemcpyDeviceToHost));
	return 0;
}

// ---------------------------------------------------------------- frame launch (launch_kernels, kernel.cu:366-439)
int Scene::render(const bm_camera* cam, const bm_frame_params* fp, float* accum, uint32_t* dbg, hipStream_t stream) {
	return render_frames(1, cam, fp, &accum, dbg ? &dbg : nullptr, stream);
}

// `count` consecutive frames -- the reference's per-frame loop (main.cpp:117-147: one launch_kernels call per frame) -- as ONE launch
// of the persistent kernel (trace.hip "FRAME RING"): frame i's constants, ticket counters and buffers are entry first + i of the
// scene's rings, and every wave walks from frame to frame by itself.
int Scene::render_frames(int count, const bm_camera* cams, const bm_frame_params* fps, float* const* accums, uint32_t* const* dbgs, hipStream_t stream) {
	if (!on_device_) { set_error("scene not generated"); return BM_ESTATE; }
	if (count < 1 || count > kMaxFramesPerLaunch) { set_error("bm_render_frames: 1 ... 256 frames per launch"); return BM_EINVAL; }
	if (!cams || !fps || !accums) { set_error("null argument"); return BM_EINVAL; }
	bool hit_records = false;
	for (int i = 0; i < count; ++i) {
		if (!accums[i]) { set_error("null accumulation buffer"); return BM_EINVAL; }
		hit_records = hit_records || (dbgs && dbgs[i]);
	}
	// ---- constants of every frame; what shapes the hand-out must be the same for all frames of a launch
	std::vector<FrameConstants> fcs(static_cast<size_t>(count));
	for (int i = 0; i < count; ++i) {
		if (int e = fill_frame_constants(cams + i, fps + i, &fcs[static_cast<size_t>(i)], hit_records)) return e;
		FrameConstants& f = fcs[static_cast<size_t>(i)];
		f.accum = accums[i];
		f.dbg = dbgs ? dbgs[i] : nullptr;
		f.frames_after = count - 1 - i;
		const FrameConstants& g = fcs[0];
		if (f.width != g.width || f.height != g.height || f.spp != g.spp || f.max_bounces != g.max_bounces || f.flags != g.flags || f.band_rows != g.band_rows ||
			f.shard_rank != g.shard_rank || f.shard_count != g.shard_count) {
			set_error("bm_render_frames: the frames of one launch must agree in width, height, spp, max_bounces, flags and shard (camera, sun, sample_base, base_frame and buffers may differ)");
			return BM_EINVAL;
		}
	}
	// In a launch of several frames a wave refills later: what argues for an early refill in a lone frame -- the paths started last are what
	// the frame's end waits for -- does not count when the next frame covers that end (ring of 20, kernel ms per frame: 24 idle lanes 0.7514,
	// 32: 0.7478, 36: 0.7481, 40: 0.7515; 1080p at 4 spp 2.937 / 2.905 / 2.894 / 2.899; profiles/r06_frame_ring.txt)
	if (count > 1)
		for (FrameConstants& f : fcs) f.refill_min = ring_refill_min(f.refill_min, f.helpers != 0, tuning().refill_min);
	const FrameConstants& fc = fcs[0];
	bool shared_digest = false; // ray-digest frames that all write ONE hit-record buffer (and one accumulation buffer)
	if (count > 1) {
		// Frames of a launch overlap in time.  Hit records are written with plain stores, and so are the pixels of frames that neither
		// run helper lanes nor (chunk, sample) items (read when a lane takes the pixel, written back when it is done): such frames
		// need buffers of their own.  Frames that add with float atomics may share one buffer, like consecutive frames of the
		// reference's accumulation (kernel.cu:319-322,341-343).
		const size_t pixels = static_cast<size_t>(fc.local_rows) * static_cast<size_t>(fc.width);
		const bool plain_pixels = !(fc.helpers || (fc.flags & BM_FLAG_SAMPLE_ITEMS));
		for (int i = 0; i < count; ++i)
			for (int k = 0; k < i; ++k) {
				const char *a = reinterpret_cast<const char*>(accums[i]), *b = reinterpret_cast<const char*>(accums[k]);
				if (plain_pixels && a < b + pixels * 16 && b < a + pixels * 16) {
					set_error("bm_render_frames: ordered frames of one launch need accumulation buffers of their own (they overlap in time and write pixels back with plain stores)");
					return BM_EINVAL;
				}
				const char *c = dbgs ? reinterpret_cast<const char*>(dbgs[i]) : nullptr, *d = dbgs ? reinterpret_cast<const char*>(dbgs[k]) : nullptr;
				if (c && d && c == d && a == b && (fc.flags & BM_FLAG_RAY_DIGEST)) { shared_digest = true; continue; } // (allowed for uniform launches: below)
				if (c && d && c < d + pixels * 32 && d < c + pixels * 32) {
					set_error("bm_render_frames: the frames of one launch need hit-record buffers of their own");
					return BM_EINVAL;
				}
			}
	}
	// ---- a UNIFORM launch?  Frames that differ only in sample_base and buffers, both stepping by constants (a resting camera: the
	// reference's progressive accumulation; bench.py's steps; a rank's batch into one allocation): lanes of consecutive frames may then
	// share a wave (trace.hip), because nothing a lane reads after it took its item depends on the frame any more.
	bool digest_ok = false;
	if (shared_digest) { // every frame names the same two buffers?
		digest_ok = true;
		for (int i = 0; i < count; ++i) digest_ok = digest_ok && dbgs[i] == dbgs[0] && accums[i] == accums[0];
	}
	if (count > 1 && (!hit_records || digest_ok)) {
		auto same_view = [&](const FrameConstants& a, const FrameConstants& b) {
			// everything up to `width` is the view, the sun and the sky (device_types.h); base_frame seeds the RNG
			return std::memcmp(&a, &b, offsetof(FrameConstants, width)) == 0 && a.base_frame == b.base_frame;
		};
		const long long sample_stride = static_cast<long long>(fcs[1].sample_base) - fcs[0].sample_base;
		const long long byte_stride = reinterpret_cast<const char*>(accums[1]) - reinterpret_cast<const char*>(accums[0]);
		const unsigned long long pixels = static_cast<unsigned long long>(fc.local_rows) * static_cast<unsigned long long>(fc.width);
		bool uniform = sample_stride >= 0 && sample_stride < (1 << 20) && byte_stride >= 0 && byte_stride % 16 == 0 &&
					   static_cast<unsigned long long>(byte_stride / 16) * static_cast<unsigned long long>(count - 1) + pixels < (1ull << 32) &&
					   static_cast<long long>(fcs[0].sample_base) + sample_stride * (count - 1) + fc.spp < (1ll << 31);
		for (int i = 1; i < count && uniform; ++i)
			uniform = same_view(fcs[static_cast<size_t>(i)], fcs[0]) && static_cast<long long>(fcs[static_cast<size_t>(i)].sample_base) == fcs[0].sample_base + sample_stride * i &&
					  reinterpret_cast<const char*>(accums[i]) == reinterpret_cast<const char*>(accums[0]) + byte_stride * i;
		if (uniform) {
			for (int i = 0; i < count; ++i) { // every entry reads like the first; the frame is an offset the lanes add themselves
				fcs[static_cast<size_t>(i)].sample_base = fcs[0].sample_base;
				fcs[static_cast<size_t>(i)].accum = fcs[0].accum;
			}
			fcs[0].ring_uniform = 1;
			fcs[0].ring_sample_stride = static_cast<int>(sample_stride);
			fcs[0].ring_pixel_stride = static_cast<uint32_t>(byte_stride / 16);
		}
		if (shared_digest && !uniform) digest_ok = false;
	}
	if (shared_digest && !digest_ok) {
		// One hit-record buffer for several frames is the digest of the WHOLE launch: its keys count samples from the first frame's
		// sample_base and its first-hit record is written once -- which only a uniform launch (one view, stepping sample_base) defines
		set_error("bm_render_frames: ray-digest frames may share one hit-record buffer only in a uniform launch (one view and sun, sample_base stepping by a constant, one accumulation buffer)");
		return BM_EINVAL;
	}
	{ // the kernel's hang guard is a 64-bit product (trace.hip round_budget): a launch for which it would wrap -- it would end before it has
	  // traced anything -- is refused (such a launch is weeks of GPU time anyway)
		const unsigned __int128 rounds = static_cast<unsigned __int128>(static_cast<unsigned long long>(fc.tiles_x) * static_cast<unsigned long long>(fc.tiles_y) * 16ull + 64ull) *
										 static_cast<unsigned long long>(fc.spp + 1) * static_cast<unsigned long long>(fc.max_bounces + 2) *
										 static_cast<unsigned long long>(2ll * world.dims.cells + world.dims.cells_height + 64) * static_cast<unsigned long long>(count);
		if (rounds >= (static_cast<unsigned __int128>(1) << 62)) {
			set_error("launch too large: tiles x samples x segments x frames overflows the kernel's round budget (render fewer samples or frames per launch)");
			return BM_EINVAL;
		}
	}
	BM_HIP(hipSetDevice(device_));
	// `stream` is used as given: nullptr is HIP's default stream (what the reference's <<<>>> launches use), which is
	// ordered with the caller's other default-stream work (e.g. torch's fill kernels on the accumulation buffer).
	if (int e = frame_begin(stream)) return e; // bricks uploaded on the load stream must be visible to these frames
	const bool instrumented = hit_records || (fc.flags & BM_FLAG_COUNTERS);
	const int slot = static_cast<int>(launches_ % kTimingRing);
	// the event pair of this slot is reused every kTimingRing launches: wait for the launch that used it last (almost always done)
	if (launches_ >= kTimingRing) BM_HIP(hipEventSynchronize(ev_stop_[slot]));
	// ---- `count` consecutive entries of the frame ring (constants + ticket counters).  An entry is free once the launch that used
	// it last has finished: launches older than the timing ring have been waited for when their event pair was recycled (above, in
	// their turn); a younger one is waited for here -- a caller that queues thousands of frames without a synchronisation would
	// otherwise overwrite constants whose copy has not run yet.
	int first = ring_next_;
	if (first + count > kFrameRing) first = 0;
	long long waited = -1;
	for (int e = first; e < first + count; ++e) {
		const long long owner = ring_owner_[e];
		if (owner >= 0 && owner != waited && launches_ - owner < kTimingRing) {
			BM_HIP(hipEventSynchronize(ev_stop_[owner % kTimingRing]));
			waited = owner;
		}
	}
	uint32_t* const work_counter = d_work_counter_ + static_cast<size_t>(first) * (kWorkCounterBytes / sizeof(uint32_t)); // one block per frame
	BM_HIP(hipMemsetAsync(work_counter, 0, kWorkCounterBytes * static_cast<size_t>(count), stream)); // ticket counters of the persistent kernel
	std::memcpy(h_frame_constants_ + first, fcs.data(), sizeof(FrameConstants) * static_cast<size_t>(count));
	BM_HIP(hipMemcpyAsync(d_frame_constants_ + first, h_frame_constants_ + first, sizeof(FrameConstants) * static_cast<size_t>(count), hipMemcpyHostToDevice, stream));
	BM_HIP(hipEventRecord(ev_start_[slot], stream));
#ifdef BM_PHASE_TIMING
	DeviceCounters* const counters_arg = d_counters_; // profiling build: the plain kernel reports its phase timers too
#else
	DeviceCounters* const counters_arg = (fc.flags & BM_FLAG_COUNTERS) ? d_counters_ : nullptr;
#endif
	launch_trace(view_, fc, d_frame_constants_ + first, counters_arg, work_counter, instrumented,
				 compute_units_, blocks_per_cu_[instrumented ? 1 : 0], stream);
	BM_HIP(hipGetLastError());
	BM_HIP(hipEventRecord(ev_stop_[slot], stream));
	if (int e = frame_end(stream)) return e; // what process_load_queue orders itself behind
	for (int e = first; e < first + count; ++e) ring_owner_[e] = launches_;
	ring_next_ = first + count;
	launches_++;
	return 0;
}

int Scene::begin_frame(hipStream_t stream, DeviceScene* view, DeviceCounters** counters) {
	if (!on_device_) { set_error("scene not generated"); return BM_ESTATE; }
	BM_HIP(hipSetDevice(device_));
	if (int e = frame_begin(stream)) return e; // bricks uploaded on the load stream must be visible to this frame
	if (view) *view = view_;
	if (counters) *counters = d_counters_;
	return 0;
}

void Scene::end_frame(hipStream_t stream) {
	other_frames_++;
	(void)frame_end(stream);
}

int Scene::resolve(const float* accum, float* out, long long n, hipStream_t stream) {
	if (!accum || !out || n < 0) { set_error("bad argument"); return BM_EINVAL; }
	BM_HIP(hipSetDevice(device_));
	launch_resolve(accum, out, n, stream);
	BM_HIP(hipGetLastError());
	return 0;
}

int Scene::synchronize() {
	BM_HIP(hipSetDevice(device_));
	BM_HIP(hipDeviceSynchronize());
	return 0;
}

int Scene::last_render_ms(float* ms) {
	if (!ms) { set_error("null argument"); return BM_EINVAL; }
	if (launches_ == 0) { set_error("no frame rendered yet"); return BM_ESTATE; }
	BM_HIP(hipSetDevice(device_));
	const int slot = static_cast<int>((launches_ - 1) % kTimingRing);
	BM_HIP(hipEventSynchronize(ev_stop_[slot]));
	BM_HIP(hipEventElapsedTime(ms, ev_start_[slot], ev_stop_[slot]));
	return 0;
}

int Scene::render_times(float* ms, int capacity, int* count) {
	if (!ms || !count || capacity <= 0) { set_error("bad argument"); return BM_EINVAL; }
	BM_HIP(hipSetDevice(device_));
	const long long have = std::min<long long>(launches_, kTimingRing);
	const long long n = std::min<long long>(have, capacity);
	for (long long k = 0; k < n; ++k) {
		const int slot = static_cast<int>((launches_ - n + k) % kTimingRing);
		BM_HIP(hipEventSynchronize(ev_stop_[slot]));
		BM_HIP(hipEventElapsedTime(&ms[k], ev_start_[slot], ev_stop_[slot]));
	}
	*count = static_cast<int>(n);
	return 0;
}

int Scene::counters_read(bm_counters* out) {
	if (!out) { set_error("null argument"); return BM_EINVAL; }
	BM_HIP(hipSetDevice(device_));
	BM_HIP(hipDeviceSynchronize());
	static_assert(sizeof(bm_counters) == sizeof(DeviceCounters::v), "counter blocks must match");
	BM_HIP(hipMemcpy(out, d_counters_, sizeof(bm_counters), hipMemcpyDeviceToHost));
	return 0;
}

int Scene::sched_stats_read(bm_sched_stats* out) {
	if (!out) { set_error("null argument"); return BM_EINVAL; }
	BM_HIP(hipSetDevice(device_));
	BM_HIP(hipDeviceSynchronize());
	static_assert(sizeof(bm_sched_stats) == sizeof(DeviceCounters::sched) + sizeof(DeviceCounters::cycles), "scheduler stat blocks must match");
	BM_HIP(hipMemcpy(out, reinterpret_cast<const char*>(d_counters_) + offsetof(DeviceCounters, sched), sizeof(bm_sched_stats), hipMemcpyDeviceToHost));
	return 0;
}

int Scene::sched_detail_read(uint64_t* out8) {
	if (!out8) { set_error("null argument"); return BM_EINVAL; }
	BM_HIP(hipSetDevice(device_));
	BM_HIP(hipDeviceSynchronize());
	BM_HIP(hipMemcpy(out8, reinterpret_cast<const char*>(d_counters_) + offsetof(DeviceCounters, detail), 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
	return 0;
}

int Scene::counters_reset() {
	BM_HIP(hipSetDevice(device_));
	BM_HIP(hipDeviceSynchronize());
	BM_HIP(hipMemset(d_counters_, 0, sizeof(DeviceCounters)));
	return 0;
}

} // namespace bm
