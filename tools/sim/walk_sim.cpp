// walk_sim.cpp -- CPU model of the brick-grid walk of traverse.h on real rays (analysis tool, not product, not a test).
// Rays: the oracle's wavefront schedule at steady state on bench config 2 (extend rays of one frame + its shadow rays).
// For every ray the reference's cell sequence is replayed one cell at a time (exact fp32 stepping, voxel.cuh:249-258) for as
// many cells as the oracle says the reference visits; the walk operations the kernel would need are counted on the way:
//   policy A (round-2 kernel): an operation = one dda_jump (ends at cube exit OR at the end of the binade of the smallest
//            tmax) or one single move, each followed by a cube-field lookup;
//   policy B: a jump continues through binade ends with the residual cube budget (iterations counted separately);
//   policy C: no binade limit at all (the geometric lower bound for isotropic cubes).
// build: g++ -O2 -std=c++17 -ffp-contract=off -Ibrickmap_amd/csrc tools/sim/walk_sim.cpp brickmap_amd/csrc/world.cpp -Loracle -l:liboracle.so -lpthread -o scratch/walk_sim
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

struct Stats {
	uint64_t rays = 0, cells = 0;
	uint64_t opsA_jump = 0, opsA_single = 0, endA_cube = 0, endA_binade = 0, opsA_small = 0;
	uint64_t opsB = 0, itersB = 0, opsB_single = 0;
	uint64_t opsC = 0, opsC_single = 0;
	uint64_t cands = 0;
	uint64_t hist_len[16] = {}; // policy A jump length (cells) histogram, log2 buckets
	uint64_t binade_stop_budget_left[8] = {}; // residual budget at a binade stop: 1,2,3-4,5-8,...
};

int main(int argc, char** argv) {
	const int G = 1024, W = 1920, H = 1080, frames = argc > 1 ? atoi(argv[1]) : 5, stride = argc > 2 ? atoi(argv[2]) : 4;
	const int JUMP_MIN = argc > 3 ? atoi(argv[3]) : 4;
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
	for (int f = 0; f < frames; ++f) {
		orc_wavefront_frame(wf, ow, &cam, W, H, 0.05f, 0.1f, accum.data());
		orc_wavefront_stats(wf, st);
		fprintf(stderr, "frame %d: survivors %u shadow %u generated %u\n", f + 1, st[0], st[1], st[4]);
	}
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
	const int campos[3] = {int(cam.position[0] / 8.f), int(cam.position[1] / 8.f), int(cam.position[2] / 8.f)};

	Stats S[3]; // 0: primary (bounces 0), 1: bounce rays, 2: shadow rays
	auto run = [&](const float* o_in, const float* d_in, bool shadow, int cls) {
		float nrm[3] = {0, 0, 0}, dist = shadow ? 0.f : 1e20f;
		int out4[4];
		uint64_t loads = 0;
		orc_intersect_voxel(ow, o_in, d_in, nrm, &dist, campos, out4, &loads);
		if (loads == 0) return;
		Stats& s = S[cls];
		s.rays++; s.cells += loads;
		// setup (voxel.cuh:136-189)
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
		uint64_t visited = 1; // the start cell
		auto step = [&]() { // one reference move; returns axis
			const bool mx = tx < ty && tx < tz, my = ty <= tx && ty < tz;
			if (mx) { px += sx; tx += ddx; return 0; }
			if (my) { py += sy; ty += ddy; return 1; }
			pz += sz; tz += ddz; return 2;
		};
		auto possible = [&]() { float m = gmin(gmin(tx, ty), tz); return fbits(m) - ((127u - 10u) << 23) < ((127u + 19u) << 23) - ((127u - 10u) << 23); };
		// ---- one pass over the ray per policy would need three replays; instead replay once with three sets of op bookkeeping
		// policy A state
		int a_n = 0, a_c[3] = {0, 0, 0}; uint32_t a_e = 0; bool a_in = false, a_single = false; int a_len = 0;
		// policy B state
		int b_n = 0, b_c[3] = {0, 0, 0}; uint32_t b_e = 0; bool b_in = false;
		// policy C state
		int c_n = 0, c_c[3] = {0, 0, 0}; bool c_in = false;
		auto inside = [&]() { return px >= 0 && py >= 0 && pz >= 0 && px < cells && py < cells && pz < cells_h; };
		while (visited < loads) {
			const int v = inside() ? F(oct, px, py, pz) : 255;
			if (v == 255) break;
			const bool poss = possible();
			const float m = gmin(gmin(tx, ty), tz);
			const uint32_t e_now = fbits(m) & 0x7F800000u;
			// --- policy A: start a new op if none in progress
			if (!a_in) {
				if (v == 0) s.cands++;
				a_n = v == 0 ? 1 : v; a_c[0] = a_c[1] = a_c[2] = 0; a_e = e_now; a_in = true; a_len = 0;
				a_single = !(poss && v >= JUMP_MIN);
				if (a_single) { if (poss && v >= 1) s.opsA_small++; s.opsA_single++; } else s.opsA_jump++;
			}
			if (!b_in) {
				b_n = v == 0 ? 1 : v; b_c[0] = b_c[1] = b_c[2] = 0; b_e = e_now; b_in = true;
				if (!(poss && v >= JUMP_MIN)) { s.opsB_single++; b_n = 1; } else { s.opsB++; s.itersB++; }
			} else if (e_now != b_e && fbits(m) >= b_e + (1u << 23)) { // binade of the smallest tmax ended: another iteration of the same op
				s.itersB++; b_e = e_now;
			}
			if (!c_in) {
				c_n = v == 0 ? 1 : v; c_c[0] = c_c[1] = c_c[2] = 0; c_in = true;
				if (!(poss && v >= JUMP_MIN)) { s.opsC_single++; c_n = 1; } else s.opsC++;
			}
			// does policy A's jump stop here because of the binade?  (the move about to be made has tmax >= 2^(e+1))
			if (a_in && !a_single && a_len > 0 && fbits(m) >= a_e + (1u << 23)) {
				s.endA_binade++;
				int left = a_n - std::max(a_c[0], std::max(a_c[1], a_c[2]));
				int b = 0; while ((1 << b) < left && b < 7) ++b; s.binade_stop_budget_left[b]++;
				int lb = 0; while ((2 << lb) <= a_len && lb < 15) ++lb; s.hist_len[lb]++;
				a_in = false;
				continue; // re-evaluate this cell as the start of a new op
			}
			const int ax = step();
			visited++;
			a_c[ax]++; a_len++; b_c[ax]++; c_c[ax]++;
			if (a_single || a_c[ax] >= a_n) {
				if (!a_single) { s.endA_cube++; int lb = 0; while ((2 << lb) <= a_len && lb < 15) ++lb; s.hist_len[lb]++; }
				a_in = false;
			}
			if (b_c[ax] >= b_n) b_in = false;
			if (c_c[ax] >= c_n) c_in = false;
		}
	};
	for (unsigned i = 0; i < Q; i += stride) run(ext[i].o, ext[i].d, false, ext[i].bounces == 0 ? 0 : 1);
	for (unsigned i = 0; i < st[1]; i += stride) run(shd[i].o, shd[i].d, true, 2);
	const char* names[3] = {"primary", "bounce", "shadow"};
	for (int c = 0; c < 3; ++c) {
		const Stats& s = S[c];
		const double r = double(s.rays);
		printf("%-8s rays %8llu  cells/ray %6.1f  candidates/ray %.2f\n", names[c], (unsigned long long)s.rays, s.cells / r, s.cands / r);
		printf("   A (kernel r02): jumps/ray %.2f (end: cube %.2f, binade %.2f)  singles/ray %.2f (of which cube<%d but jumpable: %.2f)  total ops %.2f\n",
			   s.opsA_jump / r, s.endA_cube / r, s.endA_binade / r, s.opsA_single / r, JUMP_MIN, s.opsA_small / r, (s.opsA_jump + s.opsA_single) / r);
		printf("   B (through binades, residual budget): jump ops/ray %.2f  binade iterations/ray %.2f  singles/ray %.2f  total ops %.2f\n",
			   s.opsB / r, s.itersB / r, s.opsB_single / r, (s.opsB + s.opsB_single) / r);
		printf("   C (no binade rule): jumps/ray %.2f singles/ray %.2f total %.2f\n", s.opsC / r, s.opsC_single / r, (s.opsC + s.opsC_single) / r);
		printf("   A jump length hist (cells, log2 buckets 1,2-3,4-7,...):");
		for (int k = 0; k < 10; ++k) printf(" %.2f", s.hist_len[k] / r);
		printf("\n   residual budget at binade stops (1,2,3-4,5-8,9-16,..):");
		for (int k = 0; k < 8; ++k) printf(" %.2f", s.binade_stop_budget_left[k] / r);
		printf("\n");
	}
	return 0;
}
