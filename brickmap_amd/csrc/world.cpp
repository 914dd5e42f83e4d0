// world.cpp -- CPU world build (the reference keeps this on the CPU too: BASELINE config 1
// "SimplexNoise world built by Scene.cpp on CPU (plumbing, no GPU)").
//
// Same results as the reference generator (src/Scene.cpp:44-147 + src/SimplexNoise.cpp), but
// organised for large worlds: the heightfield of a supercell column is computed once and
// reused for every z layer, and bricks that lie wholly below / above the terrain are
// emitted without touching their 512 voxels.
#include "world.h"

#include <algorithm>
#include <atomic>
#include <cstring>
#include <thread>

namespace bm {

bool WorldDims::set(int grid_size_, int grid_height_) {
	if (grid_size_ <= 0 || grid_height_ <= 0 || grid_size_ % kColumnSpan || grid_height_ % kColumnSpan) return false;
	grid_size = grid_size_;
	grid_height = grid_height_;
	cells = grid_size / kBrickSize;
	cells_height = grid_height / kBrickSize;
	supergrid_xy = cells / kSupercell;
	supergrid_z = cells_height / kSupercell;
	supercells = supergrid_xy * supergrid_xy * supergrid_z;
	return true;
}

// ---------------------------------------------------------------- simplex noise
namespace {

// Ken Perlin's permutation (SimplexNoise.cpp:73-87); must be these exact 256 values.
const uint8_t kPerm[256] = {
	151, 160, 137, 91, 90, 15, 131, 13, 201, 95, 96, 53, 194, 233, 7, 225, 140, 36, 103, 30, 69, 142, 8, 99, 37, 240,
	21, 10, 23, 190, 6, 148, 247, 120, 234, 75, 0, 26, 197, 62, 94, 252, 219, 203, 117, 35, 11, 32, 57, 177, 33, 88,
	237, 149, 56, 87, 174, 20, 125, 136, 171, 168, 68, 175, 74, 165, 71, 134, 139, 48, 27, 166, 77, 146, 158, 231, 83,
	111, 229, 122, 60, 211, 133, 230, 220, 105, 92, 41, 55, 46, 245, 40, 244, 102, 143, 54, 65, 25, 63, 161, 1, 216,
	80, 73, 209, 76, 132, 187, 208, 89, 18, 169, 200, 196, 135, 130, 116, 188, 159, 86, 164, 100, 109, 198, 173, 186,
	3, 64, 52, 217, 226, 250, 124, 123, 5, 202, 38, 147, 118, 126, 255, 82, 85, 212, 207, 206, 59, 227, 47, 16, 58,
	17, 182, 189, 28, 42, 223, 183, 170, 213, 119, 248, 152, 2, 44, 154, 163, 70, 221, 153, 101, 155, 167, 43, 172, 9,
	129, 22, 39, 253, 19, 98, 108, 110, 79, 113, 224, 232, 178, 185, 112, 104, 218, 246, 97, 228, 251, 34, 242, 193,
	238, 210, 144, 12, 191, 179, 162, 241, 81, 51, 145, 235, 249, 14, 239, 107, 49, 192, 214, 31, 181, 199, 106, 157,
	184, 84, 204, 176, 115, 121, 50, 45, 127, 4, 150, 254, 138, 236, 205, 93, 222, 114, 67, 29, 24, 72, 243, 141, 128,
	195, 78, 66, 215, 61, 156, 180
};

inline int perm_at(int i) { return kPerm[static_cast<uint8_t>(i)]; }

inline int floor_to_int(float v) { // SimplexNoise.cpp:47-50
	const int t = static_cast<int>(v);
	return v < t ? t - 1 : t;
}

// one simplex corner: falloff^4 * gradient . offset.  The gradient pick keeps the upstream
// quirk (hash masked with 0x3F but compared against 4, SimplexNoise.cpp:144-149).
inline float corner(int hash, float dx, float dy) {
	float falloff = 0.5f - dx * dx - dy * dy;
	if (falloff < 0.0f) return 0.0f;
	const int h = hash & 0x3F;
	const float u = h < 4 ? dx : dy;
	const float v = h < 4 ? dy : dx;
	const float g = ((h & 1) ? -u : u) + ((h & 2) ? -2.0f * v : 2.0f * v);
	falloff *= falloff;
	return falloff * falloff * g;
}

} // namespace

float simplex2(float x, float y) { // SimplexNoise.cpp:216-292
	const float kSkew = 0.366025403f;   // (sqrt(3)-1)/2
	const float kUnskew = 0.211324865f; // (3-sqrt(3))/6
	const float skew = (x + y) * kSkew;
	const int ci = floor_to_int(x + skew);
	const int cj = floor_to_int(y + skew);
	const float unskew = static_cast<float>(ci + cj) * kUnskew;
	const float dx0 = x - (ci - unskew);
	const float dy0 = y - (cj - unskew);
	const int oi = dx0 > dy0 ? 1 : 0; // which of the two triangles of the cell
	const int oj = 1 - oi;
	const float dx1 = dx0 - oi + kUnskew;
	const float dy1 = dy0 - oj + kUnskew;
	const float dx2 = dx0 - 1.0f + 2.0f * kUnskew;
	const float dy2 = dy0 - 1.0f + 2.0f * kUnskew;
	const float n0 = corner(perm_at(ci + perm_at(cj)), dx0, dy0);
	const float n1 = corner(perm_at(ci + oi + perm_at(cj + oj)), dx1, dy1);
	const float n2 = corner(perm_at(ci + 1 + perm_at(cj + 1)), dx2, dy2);
	return 45.23065f * (n0 + n1 + n2);
}

float fbm2(int octaves, float x, float y) { // SimplexNoise.cpp:435-450, lacunarity 2, persistence 0.5
	float sum = 0.f, norm = 0.f, freq = 1.0f, amp = 1.0f;
	for (int o = 0; o < octaves; ++o) {
		sum += amp * simplex2(x * freq, y * freq);
		norm += amp;
		freq *= 2.0f;
		amp *= 0.5f;
	}
	return sum / norm;
}

// ---------------------------------------------------------------- terrain -> bricks
void World::column_heights(int sx, int sy, float* heights) const {
	const float half = dims.grid_height / 2.f;
	for (int y = 0; y < kColumnSpan; ++y) {
		const float fy = (sy * kColumnSpan + y) / 2048.f;
		for (int x = 0; x < kColumnSpan; ++x) {
			float h = fbm2(8, (sx * kColumnSpan + x) / 2048.f, fy);
			h *= half;
			h += half;
			heights[x + y * kColumnSpan] = h;
		}
	}
}

void World::build_supercell(int sx, int sy, int sz, const float* heights) {
	HostSupercell& cell = supercells[dims.supercell_id(sx, sy, sz)];
	cell.indices.assign(kCellsPerSupercell, 0u);
	cell.bricks.clear();
	cell.resident = 0;

	// per brick column: lowest / highest terrain height under its 8x8 footprint
	float lo[kSupercell * kSupercell], hi[kSupercell * kSupercell];
	for (int by = 0; by < kSupercell; ++by)
		for (int bx = 0; bx < kSupercell; ++bx) {
			float mn = heights[bx * kBrickSize + by * kBrickSize * kColumnSpan], mx = mn;
			for (int cy = 0; cy < kBrickSize; ++cy)
				for (int cx = 0; cx < kBrickSize; ++cx) {
					const float h = heights[bx * kBrickSize + cx + (by * kBrickSize + cy) * kColumnSpan];
					mn = std::min(mn, h);
					mx = std::max(mx, h);
				}
			lo[bx + by * kSupercell] = mn;
			hi[bx + by * kSupercell] = mx;
		}

	for (int bz = 0; bz < kSupercell; ++bz) {
		const int z0 = (sz * kSupercell + bz) * kBrickSize; // global z of the brick's lowest voxel layer
		for (int by = 0; by < kSupercell; ++by)
			for (int bx = 0; bx < kSupercell; ++bx) {
				const int col = bx + by * kSupercell;
				// voxel is solid iff z < height (int compared as float, Scene.cpp:90)
				if (!(static_cast<float>(z0) < hi[col])) continue; // every voxel at or above the terrain: empty brick
				Brick brick;
				uint32_t lod = 0;
				if (static_cast<float>(z0 + kBrickSize - 1) < lo[col]) {
					std::memset(brick.data, 0xFF, sizeof brick.data); // wholly under the terrain
					lod = 0xFFu;
				} else {
					std::memset(brick.data, 0, sizeof brick.data);
					for (int cy = 0; cy < kBrickSize; ++cy)
						for (int cx = 0; cx < kBrickSize; ++cx) {
							const float h = heights[bx * kBrickSize + cx + (by * kBrickSize + cy) * kColumnSpan];
							for (int cz = 0; cz < kBrickSize; ++cz) {
								if (!(static_cast<float>(z0 + cz) < h)) break; // z ascending: the rest is air
								const int bit = cx + cy * kBrickSize + cz * kBrickSize * kBrickSize; // Scene.cpp:91-93
								brick.data[bit >> 5] |= 1u << (bit & 31);
								lod |= 1u << (((cx & 4) >> 2) + ((cy & 4) >> 1) + (cz & 4)); // Scene.cpp:95
							}
						}
				}
				cell.bricks.push_back(brick);
				cell.indices[bx + by * kSupercell + bz * kSupercell * kSupercell] =
					static_cast<uint32_t>(cell.bricks.size() - 1) | 0x80000000u | (lod << 12); // Scene.cpp:104
			}
	}
}

void World::generate_supercell(int sx, int sy, int sz) {
	if (supercells.size() != static_cast<size_t>(dims.supercells)) supercells.resize(dims.supercells);
	std::vector<float> heights(kColumnSpan * kColumnSpan);
	column_heights(sx, sy, heights.data());
	build_supercell(sx, sy, sz, heights.data());
}

void World::generate(int threads) {
	supercells.clear();
	supercells.resize(dims.supercells);
	const int columns = dims.supergrid_xy * dims.supergrid_xy;
	threads = std::max(1, std::min(threads, columns));
	std::atomic<int> next{0};
	auto work = [&]() {
		std::vector<float> heights(kColumnSpan * kColumnSpan);
		for (;;) {
			const int c = next.fetch_add(1);
			if (c >= columns) return;
			const int sx = c % dims.supergrid_xy, sy = c / dims.supergrid_xy;
			column_heights(sx, sy, heights.data());
			for (int sz = 0; sz < dims.supergrid_z; ++sz) build_supercell(sx, sy, sz, heights.data());
		}
	};
	std::vector<std::thread> pool;
	for (int i = 1; i < threads; ++i) pool.emplace_back(work);
	work();
	for (auto& t : pool) t.join();
	generated = true;
}

uint64_t World::total_bricks() const {
	uint64_t n = 0;
	for (const auto& c : supercells) n += c.bricks.size();
	return n;
}

// Largest empty cube per cell and octant: the classic "maximal square" recurrence in 3-D.  A cube of edge n anchored at
// c exists iff c is empty and cubes of edge n - 1 are anchored at the 7 neighbours c + {0,1}^3 * dir, so
// E(c) = 1 + min over those neighbours, swept from the far end of the octant's direction.  The border reads as 0
// during the sweep (a cube never leaves the grid) and is stamped 255 afterwards.
void World::build_cube_field(std::vector<uint8_t>& field, int threads) const {
	const int X = dims.cells + 2, Z = dims.cells_height + 2;
	const size_t plane = static_cast<size_t>(X) * X * Z;
	field.assign(plane * 8, 0);
	std::vector<uint8_t> occupied(plane, 1); // border counts as occupied
	for (int sc = 0; sc < dims.supercells; ++sc) {
		const HostSupercell& c = supercells[sc];
		const int sx = sc % dims.supergrid_xy, sy = (sc / dims.supergrid_xy) % dims.supergrid_xy, sz = sc / (dims.supergrid_xy * dims.supergrid_xy);
		for (int lz = 0; lz < kSupercell; ++lz)
			for (int ly = 0; ly < kSupercell; ++ly) {
				uint8_t* row = &occupied[(static_cast<size_t>(sz * kSupercell + lz + 1) * X + (sy * kSupercell + ly + 1)) * X + sx * kSupercell + 1];
				const uint32_t* words = c.indices.empty() ? nullptr : &c.indices[ly * kSupercell + lz * kSupercell * kSupercell];
				for (int lx = 0; lx < kSupercell; ++lx) row[lx] = words && words[lx] ? 1 : 0;
			}
	}
	auto sweep = [&](int oct) {
		uint8_t* f = field.data() + plane * oct;
		const int dx = (oct & 1) ? -1 : 1, dy = (oct & 2) ? -1 : 1, dz = (oct & 4) ? -1 : 1;
		const ptrdiff_t ox = dx, oy = static_cast<ptrdiff_t>(dy) * X, oz = static_cast<ptrdiff_t>(dz) * X * X;
		for (int iz = 0; iz < dims.cells_height; ++iz) {
			const int z = dz > 0 ? dims.cells_height - iz : iz + 1; // bordered coordinate, far end first
			for (int iy = 0; iy < dims.cells; ++iy) {
				const int y = dy > 0 ? dims.cells - iy : iy + 1;
				const size_t row = (static_cast<size_t>(z) * X + y) * X;
				for (int ix = 0; ix < dims.cells; ++ix) {
					const int x = dx > 0 ? dims.cells - ix : ix + 1;
					const size_t i = row + x;
					if (occupied[i]) continue; // stays 0
					const uint8_t* n = f + i;
					uint8_t m = n[ox];
					m = std::min(m, n[oy]); m = std::min(m, n[ox + oy]);
					m = std::min(m, n[oz]); m = std::min(m, n[oz + ox]); m = std::min(m, n[oz + oy]); m = std::min(m, n[oz + oy + ox]);
					f[i] = static_cast<uint8_t>(std::min<int>(m, 253) + 1);
				}
			}
		}
		for (int z = 0; z < Z; ++z) // stamp the border
			for (int y = 0; y < X; ++y) {
				uint8_t* row = f + (static_cast<size_t>(z) * X + y) * X;
				if (z == 0 || z == Z - 1 || y == 0 || y == X - 1) std::memset(row, 255, X);
				else row[0] = row[X - 1] = 255;
			}
	};
	const int n_threads = std::max(1, std::min(threads, 8));
	std::vector<std::thread> pool;
	std::atomic<int> next{0};
	for (int t = 0; t < n_threads; ++t)
		pool.emplace_back([&] {
			for (int oct = next.fetch_add(1); oct < 8; oct = next.fetch_add(1)) sweep(oct);
		});
	for (auto& th : pool) th.join();
}

} // namespace bm
