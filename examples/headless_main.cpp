// examples/headless_main.cpp -- the reference's main loop (src/main.cpp:100-147) without the window:
// State + Scene + generate(), then per frame launch_kernels -> process_load_queue, finally a PPM of the
// resolved frame.  Build: see `make -C examples` (g++ on this file, linked against libbrickmap_hip.so).
//   usage: headless_main [grid_size grid_height width height frames out.ppm [wavefront | ring]]
// With `wavefront` the frames are rendered with the reference's own queue schedule (one segment per call); with `ring` the
// world is made resident first and all frames are ONE launch of the persistent kernel (launch_frames, the frame ring).
#include <cstdint>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "../include/brickmap.hpp"

using namespace brickmap;

int main(int argc, char** argv) {
	const int grid_size = argc > 1 ? std::atoi(argv[1]) : 1024, grid_height = argc > 2 ? std::atoi(argv[2]) : 1024;
	const size_t width = argc > 3 ? std::atoi(argv[3]) : 1920, height = argc > 4 ? std::atoi(argv[4]) : 1080;
	const int frames = argc > 5 ? std::atoi(argv[5]) : 64;
	const char* out = argc > 6 ? argv[6] : "frame.ppm";
	const bool wavefront = argc > 7 && std::string(argv[7]) == "wavefront", ring = argc > 7 && std::string(argv[7]) == "ring";

	State state(width, height);                 // main.cpp:102
	Scene scene(grid_size, grid_height);        // main.cpp:104
	scene.generate();                           // main.cpp:105 -- nothing resident yet: bricks stream in on demand
	camera.position = {grid_size / 2.f, grid_size / 8.f, 0.8f * grid_height};
	camera.horizontal_angle = 0.8;
	camera.vertical_angle = -0.5;
	camera.update();                            // main.cpp:140

	if (wavefront) {
		Wavefront queues(scene.gpuScene); // state.h:19-21: ray_buffer_work / ray_buffer_next / shadow_queue_buffer
		for (int frame = 0; frame < frames; ++frame) {
			launch_kernels(state, state.blit_buffer, scene.gpuScene, queues); // main.cpp:142
			scene.process_load_queue();                                         // main.cpp:144 (the swap of :146 is inside)
		}
	} else if (ring) {
		scene.preload_all();                                                // (bricks are serviced between launches: a ring wants them resident)
		launch_frames(state, state.blit_buffer, scene.gpuScene, frames);   // main.cpp:117-147, `frames` iterations in one launch
	} else {
		for (int frame = 0; frame < frames; ++frame) {
			launch_kernels(state, state.blit_buffer, scene.gpuScene); // main.cpp:142
			scene.process_load_queue();                                 // main.cpp:144
		}
	}

	// blit_onto_framebuffer (kernel.cu:348-364) into an offscreen buffer instead of the GL surface
	void* resolved = nullptr;
	BM_CHECKED(bm_buffer_alloc(0, width * height * sizeof(vec4), &resolved));
	BM_CHECKED(bm_resolve(scene.gpuScene.handle, reinterpret_cast<const float*>(state.blit_buffer), static_cast<float*>(resolved),
						  static_cast<int64_t>(width * height), nullptr));
	std::vector<vec4> host(width * height);
	BM_CHECKED(bm_buffer_read(0, host.data(), resolved, host.size() * sizeof(vec4)));
	std::ofstream f(out, std::ios::binary);
	f << "P6\n" << width << " " << height << "\n255\n";
	for (const vec4& c : host) {
		const unsigned char rgb[3] = {static_cast<unsigned char>(std::min(1.f, std::max(0.f, c.x)) * 255.f),
									  static_cast<unsigned char>(std::min(1.f, std::max(0.f, c.y)) * 255.f),
									  static_cast<unsigned char>(std::min(1.f, std::max(0.f, c.z)) * 255.f)};
		f.write(reinterpret_cast<const char*>(rgb), 3);
	}
	bm_scene_info info;
	BM_CHECKED(bm_scene_get_info(scene.gpuScene.handle, &info));
	std::cout << "wrote " << out << ": " << frames << " frames, " << info.resident_bricks << " of " << info.total_bricks << " bricks resident\n";
	bm_buffer_free(0, resolved);
	return 0;
}
