// world.h -- host-side brickmap world: simplex-noise terrain -> supercells of 64-byte bricks.
// Product code (not the oracle).  Mirrors the host half of the reference's Scene
// (src/Scene.h:3-44, src/Scene.cpp:44-147), with the world dimensions made runtime.
#pragma once
#include <cstdint>
#include <vector>

namespace bm {

constexpr int kBrickSize = 8;        // variables.h:9
constexpr int kSupercell = 16;       // variables.h:11
constexpr int kBrickWords = 16;      // variables.h:20  (512 bits)
constexpr int kCellsPerSupercell = kSupercell * kSupercell * kSupercell;
constexpr int kColumnSpan = kSupercell * kBrickSize; // 128 voxels

struct Brick {
	uint32_t data[kBrickWords];
};
static_assert(sizeof(Brick) == 64, "a brick is one 64-byte record");

struct HostSupercell {               // Scene::Supercell, Scene.h:21-29 (host part)
	std::vector<uint32_t> indices;   // 4096 words: slot | loaded | lod<<12, 0 = empty brick
	std::vector<Brick> bricks;       // non-empty bricks in generation order
	uint32_t resident = 0;           // gpu_index_highest: next free slot of this supercell's pool
	// device pool of this supercell (Scene::Supercell gpu_count / gpu_index_highest, Scene.h:24-27), kept by Scene
	uint32_t pool_capacity = 0;      // bricks the pool can hold (0 = no pool yet)
	uint32_t pool_base = 0;          // first arena slot of the pool
};

struct WorldDims {
	int grid_size = 0, grid_height = 0;         // voxels
	int cells = 0, cells_height = 0;            // bricks
	int supergrid_xy = 0, supergrid_z = 0;      // supercells
	int supercells = 0;
	bool set(int grid_size_, int grid_height_);
	int supercell_id(int sx, int sy, int sz) const { return sx + sy * supergrid_xy + sz * supergrid_xy * supergrid_xy; }
};

// 2-D simplex noise + fBm exactly as the reference's terrain uses it (SimplexNoise.cpp:216-292,
// 435-450 with SimplexNoise(1,1,2,0.5), Scene.cpp:45,53).  Pure fp32, no contraction.
float simplex2(float x, float y);
float fbm2(int octaves, float x, float y);

class World {
public:
	WorldDims dims;
	std::vector<HostSupercell> supercells;
	bool generated = false;

	// terrain heights of one supercell column, 128x128, heights[x + 128*y] (Scene.cpp:47-58)
	void column_heights(int sx, int sy, float* heights) const;
	// Scene::generate_supercell (Scene.cpp:44-116)
	void generate_supercell(int sx, int sy, int sz);
	// CPU half of Scene::generate (Scene.cpp:118-147)
	void generate(int threads);
	uint64_t total_bricks() const;
	// Octant cube field for the GPU walk (device_types.h DeviceScene::cube_field): 8 planes of
	// (cells + 2)^2 * (cells_height + 2) bytes.  Plane o, cell c: edge (capped at 254) of the largest cube of empty
	// cells inside the grid that has c as its near corner and extends towards -x / -y / -z where bit 0 / 1 / 2 of o is
	// set, +x / +y / +z otherwise; 0 for a cell whose index word is non-zero, 255 for the border cells.
	void build_cube_field(std::vector<uint8_t>& field, int threads) const;

private:
	void build_supercell(int sx, int sy, int sz, const float* heights);
};

} // namespace bm
