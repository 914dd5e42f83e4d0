// brickmap.hpp -- header-only C++ mirror of the reference's host interface for the path-trace hot path,
// on top of the C-ABI in brickmap.h.  Same names, argument meaning and error behaviour as the reference
// (file:line into the reference's src/):
//   Camera          camera.h:3-24, camera.cpp:48-54          (input handling needs a window: out of scope)
//   State           state.h:5-34                             (ray queues / GL interop are gone: paths live in registers)
//   Scene           Scene.h:7-44, Scene.cpp:29-258
//   launch_kernels  launch.h:6, kernel.cu:366-439            (fused per-pixel paths; or, with a Wavefront, the reference's queue schedule)
//   launch_frames   main.cpp:117-147                         (that many iterations of the frame loop as ONE launch: the frame ring)
//   hip(...)        assert_cuda.h:5, assert_cuda.cpp:3-13     (print, then exit(code): the reference's cuda() macro)
// A maintainer swaps `#include "launch.h"` + the CUDA sources for this header and links libbrickmap_hip.so;
// see INTEGRATION.md for the exact diff against src/main.cpp.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "brickmap.h"

namespace brickmap {

// assert_cuda.cpp:3-13: report and abort on any device error (the C-ABI itself only returns codes)
inline int bm_assert(int code, const char* file, int line, bool abort = true) {
	if (code != 0) {
		std::fprintf(stderr, "hip_assert: %s %s %d\n", bm_last_error_string(), file, line);
		if (abort) std::exit(code);
	}
	return code;
}
#define BM_CHECKED(call) ::brickmap::bm_assert((call), __FILE__, __LINE__, true)

struct vec2 { float x, y; };
struct vec3 { float x, y, z; };
struct vec4 { float x, y, z, w; };

// variables.h:37-38, variables.cpp:3-4
inline vec2 sun_position = {0.05f, 0.1f};
inline bool sun_position_changed = true;

struct Camera { // camera.h:3-24
	vec3 position = {512, 512, 300};
	vec3 direction = {1, 0, 0};
	vec3 up = {0, 0, 1};
	float focalDistance = 1;
	float lensRadius = 0.0f;
	double horizontal_angle = 0.0;
	double vertical_angle = 0.0;

	void update() { // camera.cpp:48-54
		vec3 d = {static_cast<float>(std::cos(vertical_angle) * std::sin(horizontal_angle)),
				  static_cast<float>(std::cos(vertical_angle) * std::cos(horizontal_angle)), static_cast<float>(std::sin(vertical_angle))};
		const float inv = 1.0f / std::sqrt((d.x * d.x + d.y * d.y) + d.z * d.z); // glm::normalize
		direction = {d.x * inv, d.y * inv, d.z * inv};
	}
};
inline Camera camera; // camera.h:24 `extern Camera camera;`

class Scene { // Scene.h:7-44
public:
	struct GPUScene { // passed by value to launch_kernels, like Scene::GPUScene (Scene.h:9-17)
		bm_scene* handle = nullptr;
	};
	GPUScene gpuScene;

	// world dimensions are constexpr in the reference (variables.h:7-8: 4096 x 4096 x 512 voxels)
	explicit Scene(int grid_size = 4096, int grid_height = 512, int device = 0) { BM_CHECKED(bm_scene_create(device, grid_size, grid_height, &gpuScene.handle)); }
	~Scene() { bm_scene_destroy(gpuScene.handle); }
	Scene(const Scene&) = delete;
	Scene& operator=(const Scene&) = delete;

	void generate_supercell(int start_x, int start_y, int start_z) { BM_CHECKED(bm_scene_generate_supercell(gpuScene.handle, start_x, start_y, start_z)); }
	void generate() { BM_CHECKED(bm_scene_generate(gpuScene.handle, static_cast<int>(std::thread::hardware_concurrency()))); }
	void process_load_queue() { BM_CHECKED(bm_scene_process_load_queue(gpuScene.handle, nullptr)); }
	void dump() { BM_CHECKED(bm_scene_dump(gpuScene.handle, "dump.txt")); }
	// BASELINE configs 1-2: everything resident up front (no counterpart in the reference)
	void preload_all() { BM_CHECKED(bm_scene_preload_all(gpuScene.handle)); }
};

// Which part of the frame this process renders (no counterpart in the reference, which is single-GPU: main.cpp:89 computes
// `multi_gpu` and never uses it): the frame's rows are dealt out in bands of `band_rows` rows, band b to rank b % count, and
// a rank's rows are packed in increasing y into its State's blit_buffer.  {0, 1} = the whole frame.
struct Shard {
	int rank = 0, count = 1;
	int band_rows = 8;  // rows per band: many bands per rank, so that every rank sees every part of the image (the job is as fast as its slowest rank)
};

struct State { // state.h:5-34 without the wavefront queues and the GL interop
	vec4* blit_buffer = nullptr; // device memory, float4 per pixel: rgb = radiance sum, a = terminated paths
	size_t screen_width, screen_height;
	int device;
	Shard shard;           // the rows blit_buffer holds (default: all of them)
	size_t local_rows = 0; // = screen_height for the whole frame

	State(size_t width, size_t height, int device_ = 0, Shard shard_ = {}) : screen_width(width), screen_height(height), device(device_), shard(shard_) { alloc(); }
	~State() { bm_buffer_free(device, blit_buffer); }
	void screen_resize(size_t width, size_t height) {
		screen_width = width;
		screen_height = height;
		BM_CHECKED(bm_buffer_free(device, blit_buffer));
		alloc();
	}

private:
	void alloc() {
		bm_frame_params fp{};
		fp.width = static_cast<int32_t>(screen_width); fp.height = static_cast<int32_t>(screen_height);
		fp.band_rows = shard.count > 1 ? shard.band_rows : fp.height; fp.shard_rank = shard.rank; fp.shard_count = shard.count;
		local_rows = static_cast<size_t>(bm_local_rows(&fp));
		void* p = nullptr;
		BM_CHECKED(bm_buffer_alloc(device, screen_width * (local_rows ? local_rows : 1) * sizeof(vec4), &p));
		BM_CHECKED(bm_buffer_zero(device, p, screen_width * (local_rows ? local_rows : 1) * sizeof(vec4), nullptr));
		blit_buffer = static_cast<vec4*>(p);
	}
};

namespace detail {
inline bm_camera camera_to_c() {
	bm_camera cam{};
	cam.position[0] = camera.position.x; cam.position[1] = camera.position.y; cam.position[2] = camera.position.z;
	cam.direction[0] = camera.direction.x; cam.direction[1] = camera.direction.y; cam.direction[2] = camera.direction.z;
	cam.up[0] = camera.up.x; cam.up[1] = camera.up.y; cam.up[2] = camera.up.z;
	cam.focal_distance = camera.focalDistance;
	cam.lens_radius = camera.lensRadius;
	return cam;
}
inline bm_frame_params frame_params(const State& state, int max_bounces) {
	bm_frame_params fp{};
	fp.width = static_cast<int32_t>(state.screen_width);
	fp.height = static_cast<int32_t>(state.screen_height);
	fp.spp = 1;
	fp.max_bounces = max_bounces;
	fp.base_frame = 1;
	fp.band_rows = state.shard.count > 1 ? state.shard.band_rows : fp.height;
	fp.shard_rank = state.shard.rank;
	fp.shard_count = state.shard.count;
	fp.sun_position[0] = sun_position.x;
	fp.sun_position[1] = sun_position.y;
	return fp;
}
inline bool camera_changed(const Camera& last) {
	return last.position.x != camera.position.x || last.position.y != camera.position.y || last.position.z != camera.position.z ||
		   last.direction.x != camera.direction.x || last.direction.y != camera.direction.y || last.direction.z != camera.direction.z ||
		   last.focalDistance != camera.focalDistance || last.lensRadius != camera.lensRadius;
}
} // namespace detail

namespace detail {
// the statics of launch_kernels (kernel.cu:369,387-403: frame counter, first call, last camera) shared by its two forms
struct FusedStatics {
	bool first_time = true;
	int sample_base = 0;
	Camera last;
};
inline FusedStatics& fused_statics() { static FusedStatics s; return s; }
// kernel.cu:387-403: a camera / sun change (or the first call) resets the accumulation; returns the frame parameters of the next call
inline bm_frame_params begin_frames(State& state, vec4* blit_buffer, int spp, int max_bounces) {
	FusedStatics& st = fused_statics();
	bool reset_buffer = st.first_time || camera_changed(st.last);
	st.first_time = false;
	if (sun_position_changed) {
		sun_position_changed = false;
		reset_buffer = true;
	}
	if (reset_buffer) {
		BM_CHECKED(bm_buffer_zero(state.device, blit_buffer, state.screen_width * state.local_rows * sizeof(vec4), nullptr));
		st.sample_base = 0;
	}
	bm_frame_params fp = frame_params(state, max_bounces);
	fp.spp = spp;
	fp.sample_base = st.sample_base;
	// a shard has few pixels and (usually) many samples: (4x4 chunk, sample) work items keep the persistent waves fed
	if (state.shard.count > 1) fp.flags |= BM_FLAG_SAMPLE_ITEMS;
	return fp;
}
} // namespace detail

// launch_kernels (launch.h:6, kernel.cu:366-439).  The cudaSurfaceObject_t and the three queue pointers of the
// reference signature are gone (no GL surface; paths live in registers); `spp` complete paths per pixel are
// traced per call instead of one bounce of every in-flight path.  As in the reference the call blocks until
// the frame is done (kernel.cu:431) and a camera / sun change resets the accumulation (kernel.cu:387-403).
inline int launch_kernels(State& state, vec4* blit_buffer, Scene::GPUScene gpuScene, int spp = 1, int max_bounces = 3) {
	const bm_camera cam = detail::camera_to_c();
	const bm_frame_params fp = detail::begin_frames(state, blit_buffer, spp, max_bounces);
	BM_CHECKED(bm_render_frame(gpuScene.handle, &cam, &fp, reinterpret_cast<float*>(blit_buffer), nullptr, nullptr));
	BM_CHECKED(bm_synchronize(gpuScene.handle));
	detail::fused_statics().sample_base += spp;
	detail::fused_statics().last = camera;
	return 0; // the reference always returns cudaSuccess (kernel.cu:438)
}

// `frames` consecutive iterations of the reference's frame loop (main.cpp:117-147: launch_kernels once per frame) for the current
// camera and sun as ONE launch of the persistent kernel (bm_render_frames, the "frame ring"): exactly the frames that many
// launch_kernels calls render -- each with its own sample offsets, ticket counters and rays, all accumulating into blit_buffer like
// consecutive frames of the reference (kernel.cu:319-322,341-343) -- but a wave that has run out of work in frame i starts on
// frame i+1 by itself, so the end of a frame is covered by the next one (1080p, 1 spp: 0.86 instead of 1.03 ms per frame).
// Blocks until the last frame is done.  A host whose camera moves every frame keeps calling launch_kernels.
inline int launch_frames(State& state, vec4* blit_buffer, Scene::GPUScene gpuScene, int frames, int spp = 1, int max_bounces = 3) {
	if (frames < 1) return 0;
	const bm_camera cam = detail::camera_to_c();
	const bm_frame_params first = detail::begin_frames(state, blit_buffer, spp, max_bounces);
	for (int done = 0; done < frames;) {
		const int n = frames - done < 256 ? frames - done : 256; // (bm_render_frames takes up to 256 frames per launch)
		std::vector<bm_camera> cams(static_cast<size_t>(n), cam);
		std::vector<bm_frame_params> fps(static_cast<size_t>(n), first);
		std::vector<float*> buffers(static_cast<size_t>(n), reinterpret_cast<float*>(blit_buffer));
		for (int i = 0; i < n; ++i) fps[static_cast<size_t>(i)].sample_base = first.sample_base + (done + i) * spp;
		BM_CHECKED(bm_render_frames(gpuScene.handle, n, cams.data(), fps.data(), buffers.data(), nullptr, nullptr));
		done += n;
	}
	BM_CHECKED(bm_synchronize(gpuScene.handle));
	detail::fused_statics().sample_base += frames * spp;
	detail::fused_statics().last = camera;
	return 0;
}

// ---- multi-GPU: one process (or thread) per GPU, every one with its own Scene replica, State shard and Comm.
// The per-frame loop of main.cpp:142-147 becomes, on every rank:
//     launch_kernels(state, state.blit_buffer, scene.gpuScene, spp);      // this rank's bands (state.shard)
//     scene.process_load_queue();
//     gather_frame(comm, state, frame_on_root);                           // ncclSend / ncclRecv + assembly, RCCL over xGMI
// INTEGRATION.md has the whole program.
struct Comm {
	bm_comm* handle = nullptr;
	int rank = 0, world = 1;
	// ncclGetUniqueId: called on ONE rank; hand the 128 bytes to the others (MPI_Bcast, a file, a socket)
	static void unique_id(unsigned char id[BM_COMM_ID_BYTES]) { BM_CHECKED(bm_comm_unique_id(id)); }
	Comm(int device, int rank_, int world_, const unsigned char id[BM_COMM_ID_BYTES]) : rank(rank_), world(world_) {
		BM_CHECKED(bm_comm_create(device, rank, world, id, &handle));
	}
	~Comm() { bm_comm_destroy(handle); }
	Comm(const Comm&) = delete;
	Comm& operator=(const Comm&) = delete;
	void barrier() { BM_CHECKED(bm_comm_barrier(handle, nullptr)); }
};

// The exchange of one frame: every rank's packed bands to `root`, assembled there into frame_on_root (height x width float4,
// device memory of the root; ignored on the other ranks).  Enqueued on the default stream behind the frame; returns at once.
inline void gather_frame(Comm& comm, const State& state, vec4* frame_on_root, int root = 0) {
	BM_CHECKED(bm_gather_frame(comm.handle, reinterpret_cast<const float*>(state.blit_buffer), reinterpret_cast<float*>(frame_on_root),
							   static_cast<int>(state.screen_height), static_cast<int>(state.screen_width), state.shard.count > 1 ? state.shard.band_rows : static_cast<int>(state.screen_height),
							   root, nullptr));
}

// The exchange of a BATCH of frames (what a rank rendered with launch_frames-style launches into one allocation of `count` packed
// shard buffers): one message per peer for the whole batch (bm_gather_frames).  packed = count x local_rows x width float4,
// frames_on_root = count x height x width float4 (ignored on the other ranks).
inline void gather_frames(Comm& comm, const State& state, const vec4* packed, vec4* frames_on_root, int count, int root = 0) {
	BM_CHECKED(bm_gather_frames(comm.handle, reinterpret_cast<const float*>(packed), reinterpret_cast<float*>(frames_on_root), count,
								static_cast<int>(state.screen_height), static_cast<int>(state.screen_width),
								state.shard.count > 1 ? state.shard.band_rows : static_cast<int>(state.screen_height), root, nullptr));
}

// Streams that run side by side (bm_probe_streams): for a host that pipelines frames over two streams with the exchange on a third.
// The HIP stream handles come back as void*; the caller releases them with bm_release_streams (or hipStreamDestroy).
inline std::vector<void*> probe_streams(int device, int count) {
	std::vector<void*> streams(static_cast<size_t>(count), nullptr);
	BM_CHECKED(bm_probe_streams(device, count, streams.data()));
	return streams;
}

// The reference's own schedule.  RayQueue* queue / queue2 and ShadowQueue* shadowQueue of the reference signature
// (state.h:19-21) and the statics / __device__ globals of kernel.cu:106-119,369 live in a Wavefront object; one call
// advances every path in flight by one segment, exactly like the reference's launch_kernels, and the buffer swap of
// main.cpp:146 happens inside.  `ray_queue_buffer_size` is variables.h:61.
struct Wavefront {
	bm_wavefront* handle = nullptr;
	explicit Wavefront(Scene::GPUScene gpuScene, uint32_t ray_queue_buffer_size = 2 * 1048576) {
		BM_CHECKED(bm_wavefront_create(gpuScene.handle, ray_queue_buffer_size, &handle));
	}
	~Wavefront() { bm_wavefront_destroy(handle); }
	Wavefront(const Wavefront&) = delete;
	Wavefront& operator=(const Wavefront&) = delete;
};

inline int launch_kernels(State& state, vec4* blit_buffer, Scene::GPUScene gpuScene, Wavefront& queues, int max_bounces = 3) {
	static Camera last;
	static bool first_time = true;
	bool reset_buffer = !first_time && detail::camera_changed(last); // kernel.cu:387
	first_time = false;
	if (sun_position_changed) { // kernel.cu:389-395
		sun_position_changed = false;
		reset_buffer = true;
	}
	if (reset_buffer) { // kernel.cu:397-403
		BM_CHECKED(bm_buffer_zero(state.device, blit_buffer, state.screen_width * state.screen_height * sizeof(vec4), nullptr));
		BM_CHECKED(bm_wavefront_reset(queues.handle));
	}
	const bm_camera cam = detail::camera_to_c();
	const bm_frame_params fp = detail::frame_params(state, max_bounces);
	BM_CHECKED(bm_wavefront_frame(queues.handle, &cam, &fp, reinterpret_cast<float*>(blit_buffer), nullptr));
	BM_CHECKED(bm_synchronize(gpuScene.handle)); // kernel.cu:431
	last = camera;
	return 0;
}

} // namespace brickmap
