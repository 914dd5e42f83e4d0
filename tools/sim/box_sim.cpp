// box_sim.cpp -- what anisotropic empty BOXES (nxy x nxy x nz per cell and octant) would buy over the isotropic cube field.
// Same rays and replay as walk_sim.cpp; the walk operation model is the round-2 kernel's (a jump ends at the box exit or at
// the end of the binade of the smallest tmax; boxes with every edge below JUMP_MIN are crossed by single moves).
// build: g++ -O2 -std=c++17 -ffp-contract=off -Ibrickmap_amd/csrc tools/sim/box_sim.cpp brickmap_amd/csrc/world.cpp -Loracle -l:liboracle.so -lpthread -o scratch/box_sim
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "world.h"

extern "C" {
void* orc_world_create(int, int);
void orc_world_generate(void*, int);
void orc_world_reset_device(void*, int);
void* orc_wavefront_create(unsigned, int);
void orc_wavefront_frame(void*, void*, const void* cam, int W, int H, float, float, float* accum);
void orc_wavefront_stats(const void*, unsigned* out6);
int orc_wavefront_read_queue(const void*, int which, unsigned first, unsigned count, void* out);
int orc_intersect_voxel(void* w, const float* origin, const float* direction, float* normal_io, float* distance_io, const int* campos, int* out4, uint64_t* index_loads);
void orc_camera_direction(double, double, float*);
}
struct OrcCamera { float position[3], direction[3], up[3], focal, lens; };
struct RayRec { float o[3], d[3], thr[3], n[3], dist; int id, bounces; unsigned pixel; };
struct ShadowRec { float o[3], d[3], c[3]; unsigned pixel; };
static inline uint32_t fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float gmin(float a, float b) { return (b < a) ? b : a; }
static inline float gmax(float a, float b) { return (a < b) ? b : a; }
static inline int isign(float x) { return (0.f < x) - (x < 0.f); }

static const int kSizes[] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 24, 32, 48, 64, 96, 128};
constexpr int kNS = 16;

int main(int argc, char** argv) {
	const int G = 1024, W = 1920, H = 1080, frames = 5, stride = argc > 1 ? atoi(argv[1]) : 8;
	const int JUMP_MIN = argc > 2 ? atoi(argv[2]) : 4;
	void* ow = orc_world_create(G, G);
	orc_world_generate(ow, 8);
	orc_world_reset_device(ow, 1);
	OrcCamera cam{};
	cam.position[0] = G / 2.f; cam.position[1] = G / 8.f; cam.position[2] = 0.8f * G;
	orc_camera_direction(0.8, -0.5, cam.direction);
	cam.up[2] = 1.f; cam.focal = 1.f;
	const unsigned Q = 2u * 1048576u;
	void* wf = orc_wavefront_create(Q, 3);
	std::vector<float> accum(size_t(W) * H * 4);
	unsigned st[6];
	for (int f = 0; f < frames; ++f) { orc_wavefront_frame(wf, ow, &cam, W, H, 0.05f, 0.1f, accum.data()); orc_wavefront_stats(wf, st); }
	std::vector<RayRec> ext(Q);
	orc_wavefront_read_queue(wf, 2, 0, Q, ext.data());
	std::vector<ShadowRec> shd(st[1]);
	orc_wavefront_read_queue(wf, 1, 0, st[1], shd.data());

	bm::World world;
	world.dims.set(G, G);
	world.generate(8);
	std::vector<uint8_t> field;
	world.build_cube_field(field, 8);
	const int cells = world.dims.cells, cells_h = world.dims.cells_height, cfx = cells + 2;
	const size_t plane = field.size() / 8;
	auto F = [&](int oct, int x, int y, int z) -> int { return field[oct * plane + (size_t(z + 1) * cfx + (y + 1)) * cfx + (x + 1)]; };
	// summed-area table of occupied cells
	const int sx1 = cells + 1, sz1 = cells_h + 1;
	std::vector<uint32_t> sat(size_t(sx1) * sx1 * sz1, 0);
	auto SAT = [&](int x, int y, int z) -> uint32_t& { return sat[(size_t(z) * sx1 + y) * sx1 + x]; };
	for (int z = 0; z < cells_h; ++z) for (int y = 0; y < cells; ++y) for (int x = 0; x < cells; ++x) {
		const uint32_t occ = F(0, x, y, z) == 0 ? 1u : 0u;
		SAT(x + 1, y + 1, z + 1) = occ + SAT(x, y + 1, z + 1) + SAT(x + 1, y, z + 1) + SAT(x + 1, y + 1, z) - SAT(x, y, z + 1) - SAT(x, y + 1, z) - SAT(x + 1, y, z) + SAT(x, y, z);
	}
	auto box_sum = [&](int x0, int x1, int y0, int y1, int z0, int z1) -> uint32_t { // inclusive ranges, inside the grid
		return SAT(x1 + 1, y1 + 1, z1 + 1) - SAT(x0, y1 + 1, z1 + 1) - SAT(x1 + 1, y0, z1 + 1) - SAT(x1 + 1, y1 + 1, z0) + SAT(x0, y0, z1 + 1) + SAT(x0, y1 + 1, z0) + SAT(x1 + 1, y0, z0) - SAT(x0, y0, z0);
	};
	auto box_empty = [&](int oct, int x, int y, int z, int a, int b) -> bool {
		int x0 = (oct & 1) ? x - a + 1 : x, x1 = (oct & 1) ? x : x + a - 1;
		int y0 = (oct & 2) ? y - a + 1 : y, y1 = (oct & 2) ? y : y + a - 1;
		int z0 = (oct & 4) ? z - b + 1 : z, z1 = (oct & 4) ? z : z + b - 1;
		if (x0 < 0 || y0 < 0 || z0 < 0 || x1 >= cells || y1 >= cells || z1 >= cells_h) return false;
		return box_sum(x0, x1, y0, y1, z0, z1) == 0;
	};
	// heuristics: 0 = isotropic cube (baseline), 1 = max volume, 2 = max a*b (area-ish), 3 = max min-steps over three probe directions,
	// 4 = exact (unquantised) variant of 3
	auto choose = [&](int mode, int oct, int x, int y, int z, int& a_out, int& b_out) {
		const int n = F(oct, x, y, z);
		if (mode == 0 || n == 0) { a_out = b_out = n; return; }
		double best = -1; int ba = n, bb = n;
		auto consider = [&](int a, int b) {
			double u;
			if (mode == 1) u = double(a) * a * b;
			else if (mode == 2) u = double(a) * b;
			else { // expected cells crossed ~ min(a/|dx|, a/|dy|, b/|dz|) * (|dx|+|dy|+|dz|) for probe directions
				const double probes[3][3] = {{0.577, 0.577, 0.577}, {0.7, 0.65, 0.3}, {0.35, 0.3, 0.89}};
				u = 0;
				for (auto& p : probes) u += std::min(std::min(a / p[0], a / p[1]), b / p[2]) * (p[0] + p[1] + p[2]);
			}
			if (u > best) { best = u; ba = a; bb = b; }
		};
		if (mode == 4) {
			int a = 254;
			for (int b = 1; b <= 254; ++b) { // a is non-increasing in b
				while (a >= 1 && !box_empty(oct, x, y, z, a, b)) --a;
				if (a < 1) break;
				consider(a, b);
			}
		} else {
			int ai = kNS - 1;
			for (int bi = 0; bi < kNS; ++bi) {
				while (ai >= 0 && !box_empty(oct, x, y, z, kSizes[ai], kSizes[bi])) --ai;
				if (ai < 0) break;
				consider(kSizes[ai], kSizes[bi]);
			}
		}
		a_out = ba; b_out = bb;
	};
	const int campos[3] = {int(cam.position[0] / 8.f), int(cam.position[1] / 8.f), int(cam.position[2] / 8.f)};
	constexpr int MODES = 5;
	struct Stats { uint64_t rays = 0, cells = 0, jumps[MODES] = {}, singles[MODES] = {}, binade[MODES] = {}; } S[3];
	auto run = [&](const float* o_in, const float* d_in, bool shadow, int cls) {
		float nrm[3] = {0, 0, 0}, dist = shadow ? 0.f : 1e20f;
		int out4[4];
		uint64_t loads = 0;
		orc_intersect_voxel(ow, o_in, d_in, nrm, &dist, campos, out4, &loads);
		if (loads == 0) return;
		Stats& s = S[cls];
		s.rays++; s.cells += loads;
		for (int mode = 0; mode < MODES; ++mode) {
			float ox = o_in[0], oy = o_in[1], oz = o_in[2];
			const float dx = d_in[0], dy = d_in[1], dz = d_in[2];
			const float gs = float(G), gh = float(G);
			float t1x = (0.f - ox) / dx, t1y = (0.f - oy) / dy, t1z = (0.f - oz) / dz, t2x = (gs - ox) / dx, t2y = (gs - oy) / dy, t2z = (gh - oz) / dz;
			float tminn = gmax(gmax(gmin(t1x, t2x), 0.f), gmax(gmin(t1y, t2y), gmin(t1z, t2z)));
			if (tminn > 0) {
				ox += dx * tminn; oy += dy * tminn; oz += dz * tminn;
				float cx = gs / 2.f - ox, cy = gs / 2.f - oy, cz = gh / 2.f - oz;
				float ax = fabsf(cx) * (1.f / (gs / gh)), ay = fabsf(cy) * (1.f / (gs / gh)), az = fabsf(cz) * 1.f;
				float m = gmax(ax, gmax(ay, az));
				float nx = float(isign(-cx)) * truncf(ax / m + 0.000001f), ny = float(isign(-cy)) * truncf(ay / m + 0.000001f), nz = float(isign(-cz)) * truncf(az / m + 0.000001f);
				ox -= nx * 0.001f; oy -= ny * 0.001f; oz -= nz * 0.001f;
			}
			ox /= 8.f; oy /= 8.f; oz /= 8.f;
			int px = int(ox), py = int(oy), pz = int(oz);
			const int sx = isign(dx), sy = isign(dy), sz = isign(dz);
			const float rx = dx == 0.f ? 0.f : 1.f / dx, ry = dy == 0.f ? 0.f : 1.f / dy, rz = dz == 0.f ? 0.f : 1.f / dz;
			float tx = dx != 0.f ? ((dx > 0 ? float(px + 1) : float(px)) - ox) * rx : 1000000.f;
			float ty = dy != 0.f ? ((dy > 0 ? float(py + 1) : float(py)) - oy) * ry : 1000000.f;
			float tz = dz != 0.f ? ((dz > 0 ? float(pz + 1) : float(pz)) - oz) * rz : 1000000.f;
			const float ddx = float(sx) * rx, ddy = float(sy) * ry, ddz = float(sz) * rz;
			const int oct = (dx < 0 ? 1 : 0) | (dy < 0 ? 2 : 0) | (dz < 0 ? 4 : 0);
			uint64_t visited = 1;
			bool in_op = false, single = false; int na = 0, nb = 0, c[3] = {0, 0, 0}, len = 0; uint32_t e0 = 0;
			while (visited < loads) {
				const bool inside = px >= 0 && py >= 0 && pz >= 0 && px < cells && py < cells && pz < cells_h;
				if (!inside) break;
				const float m = gmin(gmin(tx, ty), tz);
				const bool poss = fbits(m) - ((127u - 10u) << 23) < ((127u + 19u) << 23) - ((127u - 10u) << 23);
				if (!in_op) {
					int a, b;
					choose(mode, oct, px, py, pz, a, b);
					if (a == 0) a = b = 1; // a passed-through brick: one plain move
					na = a; nb = b; c[0] = c[1] = c[2] = 0; len = 0; e0 = fbits(m) & 0x7F800000u; in_op = true;
					single = !(poss && std::max(a, b) >= JUMP_MIN);
					if (single) s.singles[mode]++; else s.jumps[mode]++;
				}
				if (!single && len > 0 && fbits(m) >= e0 + (1u << 23)) { s.binade[mode]++; in_op = false; continue; }
				const bool mx = tx < ty && tx < tz, my = ty <= tx && ty < tz;
				int axis;
				if (mx) { px += sx; tx += ddx; axis = 0; } else if (my) { py += sy; ty += ddy; axis = 1; } else { pz += sz; tz += ddz; axis = 2; }
				visited++; c[axis]++; len++;
				if (single || c[axis] >= (axis == 2 ? nb : na)) in_op = false;
			}
		}
	};
	for (unsigned i = 0; i < Q; i += stride) run(ext[i].o, ext[i].d, false, ext[i].bounces == 0 ? 0 : 1);
	for (unsigned i = 0; i < st[1]; i += stride) run(shd[i].o, shd[i].d, true, 2);
	const char* names[3] = {"primary", "bounce", "shadow"};
	const char* mn[MODES] = {"cube", "box/volume", "box/a*b", "box/probes(q)", "box/probes(exact)"};
	uint64_t tr = 0, tj[MODES] = {}, ts[MODES] = {};
	for (int c = 0; c < 3; ++c) {
		const Stats& s = S[c];
		const double r = double(s.rays);
		printf("%-8s rays %8llu  cells/ray %6.1f\n", names[c], (unsigned long long)s.rays, s.cells / r);
		for (int m = 0; m < MODES; ++m) {
			printf("   %-18s jumps/ray %.2f (binade stops %.2f) singles/ray %.2f total %.2f\n", mn[m], s.jumps[m] / r, s.binade[m] / r, s.singles[m] / r, (s.jumps[m] + s.singles[m]) / r);
			tj[m] += s.jumps[m]; ts[m] += s.singles[m];
		}
		tr += s.rays;
	}
	printf("all rays:\n");
	for (int m = 0; m < MODES; ++m) printf("   %-18s jumps/ray %.2f singles/ray %.2f total %.2f\n", mn[m], tj[m] / double(tr), ts[m] / double(tr), (tj[m] + ts[m]) / double(tr));
	return 0;
}
