// examples/multi_gpu_main.cpp -- the reference's main loop (src/main.cpp:100-147) on N GPUs of one node, one process per GPU:
// every rank holds a full Scene replica, renders the interleaved 8-row bands of the frame that belong to it and hands them to
// rank 0 through the C-ABI's RCCL gather (bm_gather_frame: ncclSend / ncclRecv over xGMI + one assembly kernel).
//   usage: multi_gpu_main <rank> <world> <id_file> [grid_size grid_height width height frames spp out.ppm]
// Start one process per GPU with the same <id_file> (a path all of them can reach): rank 0 writes the 128-byte communicator
// id there, the others wait for it -- any other way of moving 128 bytes (MPI_Bcast, a socket) does as well.  E.g.
//   for r in 0 1 2 3 4 5 6 7; do ./multi_gpu_main $r 8 /tmp/bm_id & done; wait
// Rank r renders on GPU r.  Build: `make -C examples`.
#include <chrono>
#include <cstdint>
#include <fstream>
#include <iostream>
#include <thread>
#include <vector>

#include "../include/brickmap.hpp"

using namespace brickmap;

int main(int argc, char** argv) {
	if (argc < 4) { std::cerr << "usage: multi_gpu_main <rank> <world> <id_file> [grid_size grid_height width height frames spp out.ppm]\n"; return 2; }
	const int rank = std::atoi(argv[1]), world = std::atoi(argv[2]);
	const std::string id_file = argv[3];
	const int grid_size = argc > 4 ? std::atoi(argv[4]) : 1024, grid_height = argc > 5 ? std::atoi(argv[5]) : 1024;
	const size_t width = argc > 6 ? std::atoi(argv[6]) : 1920, height = argc > 7 ? std::atoi(argv[7]) : 1080;
	const int frames = argc > 8 ? std::atoi(argv[8]) : 16, spp = argc > 9 ? std::atoi(argv[9]) : 8;
	const char* out = argc > 10 ? argv[10] : "frame.ppm";
	const int device = rank; // one GPU per rank

	// the communicator id: made by rank 0, read by everybody else
	unsigned char id[BM_COMM_ID_BYTES];
	if (rank == 0) {
		Comm::unique_id(id);
		std::ofstream f(id_file + ".tmp", std::ios::binary);
		f.write(reinterpret_cast<const char*>(id), sizeof id);
		f.close();
		std::rename((id_file + ".tmp").c_str(), id_file.c_str());
	} else {
		for (;;) {
			std::ifstream f(id_file, std::ios::binary);
			if (f && f.read(reinterpret_cast<char*>(id), sizeof id)) break;
			std::this_thread::sleep_for(std::chrono::milliseconds(20));
		}
	}
	Comm comm(device, rank, world, id);

	State state(width, height, device, Shard{rank, world, 8}); // this rank's bands, packed (main.cpp:102)
	Scene scene(grid_size, grid_height, device);                // a full replica per GPU (main.cpp:104)
	scene.generate();                                           // main.cpp:105 -- bricks stream in on demand, per GPU
	camera.position = {grid_size / 2.f, grid_size / 8.f, 0.8f * grid_height};
	camera.horizontal_angle = 0.8;
	camera.vertical_angle = -0.5;
	camera.update();

	void* frame = nullptr; // the assembled frame, on the root only
	if (rank == 0) BM_CHECKED(bm_buffer_alloc(device, width * height * sizeof(vec4), &frame));
	comm.barrier();
	const auto t0 = std::chrono::steady_clock::now();
	for (int f = 0; f < frames; ++f) {
		launch_kernels(state, state.blit_buffer, scene.gpuScene, spp); // main.cpp:142: this rank's rows, `spp` samples each
		scene.process_load_queue();                                     // main.cpp:144
		gather_frame(comm, state, static_cast<vec4*>(frame));           // the one exchange of the frame
	}
	BM_CHECKED(bm_synchronize(scene.gpuScene.handle));
	comm.barrier();
	const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

	if (rank == 0) {
		void* resolved = nullptr;
		BM_CHECKED(bm_buffer_alloc(device, width * height * sizeof(vec4), &resolved));
		BM_CHECKED(bm_resolve(scene.gpuScene.handle, static_cast<const float*>(frame), static_cast<float*>(resolved), static_cast<int64_t>(width * height), nullptr));
		std::vector<vec4> host(width * height);
		BM_CHECKED(bm_buffer_read(device, host.data(), resolved, host.size() * sizeof(vec4)));
		std::ofstream f(out, std::ios::binary);
		f << "P6\n" << width << " " << height << "\n255\n";
		for (const vec4& c : host) {
			const unsigned char rgb[3] = {static_cast<unsigned char>(std::min(1.f, std::max(0.f, c.x)) * 255.f),
										  static_cast<unsigned char>(std::min(1.f, std::max(0.f, c.y)) * 255.f),
										  static_cast<unsigned char>(std::min(1.f, std::max(0.f, c.z)) * 255.f)};
			f.write(reinterpret_cast<const char*>(rgb), 3);
		}
		std::cout << "wrote " << out << ": " << world << " ranks, " << frames << " frames of " << spp << " spp in " << secs * 1e3 << " ms ("
				  << double(width) * height * spp * 4 * frames / secs / 1e6 << " nominal Mrays/s)\n";
		bm_buffer_free(device, resolved);
		bm_buffer_free(device, frame);
		std::remove(id_file.c_str());
	}
	return 0;
}
