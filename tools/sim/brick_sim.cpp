// brick_sim.cpp -- how many 8^3 brick walks a conservative pre-test against the index word's 2^3 LoD byte would remove, and
// what that does to the longest walk of a candidate pass (analysis tool; rays as in walk_sim.cpp).
// Pre-test: the chord of the ray through the brick (entry point .. exit point), its bounding box grown by EPS voxels; if no
// occupied 4^3 sub-block (LoD byte, Scene.cpp:95) overlaps the box, the exact voxel walk cannot find a voxel.
// build: g++ -O2 -std=c++17 -ffp-contract=off tools/sim/brick_sim.cpp -Loracle -l:liboracle.so -lpthread -Wl,-rpath,$PWD/oracle -o scratch/brick_sim
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
extern "C" {
void* orc_world_create(int, int);
void orc_world_generate(void*, int);
void orc_world_reset_device(void*, int);
void* orc_wavefront_create(unsigned, int);
void orc_wavefront_frame(void*, void*, const void* cam, int W, int H, float, float, float* accum);
void orc_camera_direction(double, double, float*);
typedef void (*probe_t)(const float*, const float*, uint32_t, const uint32_t*, int, unsigned);
void orc_set_brick_probe(probe_t);
}
struct OrcCamera { float position[3], direction[3], up[3], focal, lens; };
static uint64_t tests = 0, hits = 0, steps_all = 0, filtered = 0, filtered_steps = 0, wrong = 0, hist[32] = {}, hist_f[32] = {};
static uint64_t filt2 = 0, filt2_steps = 0, wrong2 = 0;
static std::vector<uint8_t> trip_plain, trip_filt; // per test: steps (0 if filtered) for the wave-max model
static bool collecting = false;
static const float EPS = 0.01f;
static void probe(const float* o, const float* d, uint32_t word, const uint32_t* brick, int hit, unsigned steps) {
	if (!collecting) return;
	tests++; hits += hit; steps_all += steps; hist[std::min(steps, 31u)]++;
	const uint32_t byte = (word >> 12) & 0xFFu;
	// exit parameter of the chord
	float t_exit = 1e30f;
	for (int a = 0; a < 3; ++a) {
		if (d[a] > 0.f) t_exit = std::min(t_exit, (8.f - o[a]) / d[a]);
		else if (d[a] < 0.f) t_exit = std::min(t_exit, (0.f - o[a]) / d[a]);
	}
	t_exit = std::max(t_exit, 0.f);
	uint32_t mask = 0;
	int lo[3], hi[3];
	for (int a = 0; a < 3; ++a) {
		const float p1 = o[a] + d[a] * t_exit;
		const float mn = std::min(o[a], p1) - EPS, mx = std::max(o[a], p1) + EPS;
		lo[a] = mn >= 4.f ? 1 : 0;
		hi[a] = mx >= 4.f ? 1 : 0;
	}
	for (int z = lo[2]; z <= hi[2]; ++z) for (int y = lo[1]; y <= hi[1]; ++y) for (int x = lo[0]; x <= hi[0]; ++x) mask |= 1u << (x + 2 * y + 4 * z);
	const bool f = (byte & mask) == 0;
	if (f) { filtered++; filtered_steps += steps; if (hit) wrong++; } else hist_f[std::min(steps, 31u)]++;
	// variant 2: per-z-slab boxes (the chord cut at z = 4): tighter for steep chords
	bool f2 = f;
	if (!f && d[2] != 0.f) {
		const float tz = (4.f - o[2]) / d[2];
		uint32_t m2 = 0;
		auto add_box = [&](float ta, float tb) {
			if (tb < ta) return;
			int l[3], h[3];
			for (int a = 0; a < 3; ++a) {
				const float pa = o[a] + d[a] * ta, pb = o[a] + d[a] * tb;
				l[a] = std::min(pa, pb) - EPS >= 4.f ? 1 : 0;
				h[a] = std::max(pa, pb) + EPS >= 4.f ? 1 : 0;
			}
			for (int z = l[2]; z <= h[2]; ++z) for (int y = l[1]; y <= h[1]; ++y) for (int x = l[0]; x <= h[0]; ++x) m2 |= 1u << (x + 2 * y + 4 * z);
		};
		if (tz > 0.f && tz < t_exit) { add_box(0.f, tz); add_box(tz, t_exit); } else add_box(0.f, t_exit);
		f2 = (byte & m2) == 0;
	}
	if (f2) { filt2++; filt2_steps += steps; if (hit) wrong2++; }
	trip_plain.push_back((uint8_t)std::min(steps, 255u));
	trip_filt.push_back(f2 ? 0 : (uint8_t)std::min(steps, 255u));
	(void)brick;
}
int main() {
	const int G = 1024, W = 1920, H = 1080;
	void* ow = orc_world_create(G, G);
	orc_world_generate(ow, 8);
	orc_world_reset_device(ow, 1);
	OrcCamera cam{};
	cam.position[0] = G / 2.f; cam.position[1] = G / 8.f; cam.position[2] = 0.8f * G;
	orc_camera_direction(0.8, -0.5, cam.direction);
	cam.up[2] = 1.f; cam.focal = 1.f;
	void* wf = orc_wavefront_create(2u * 1048576u, 3);
	std::vector<float> accum(size_t(W) * H * 4);
	orc_set_brick_probe(probe);
	for (int f = 0; f < 5; ++f) { collecting = f == 4; orc_wavefront_frame(wf, ow, &cam, W, H, 0.05f, 0.1f, accum.data()); }
	printf("brick tests %llu, hits %llu (%.1f %%), voxel steps %.2f per test\n", (unsigned long long)tests, (unsigned long long)hits, 100.0 * hits / tests, double(steps_all) / tests);
	printf("bbox pre-test: removes %llu tests (%.1f %% of all, %.1f %% of the pass-throughs), %.1f %% of the voxel steps; WRONG (a hit filtered): %llu\n", (unsigned long long)filtered,
		   100.0 * filtered / tests, 100.0 * filtered / (tests - hits), 100.0 * filtered_steps / steps_all, (unsigned long long)wrong);
	printf("two-slab pre-test: removes %llu tests (%.1f %% of all, %.1f %% of the pass-throughs), %.1f %% of the voxel steps; WRONG: %llu\n", (unsigned long long)filt2,
		   100.0 * filt2 / tests, 100.0 * filt2 / (tests - hits), 100.0 * filt2_steps / steps_all, (unsigned long long)wrong2);
	printf("steps histogram all   :"); for (int i = 0; i < 24; ++i) printf(" %.1f", 100.0 * hist[i] / tests); printf("\n");
	printf("steps histogram kept  :"); for (int i = 0; i < 24; ++i) printf(" %.1f", 100.0 * hist_f[i] / tests); printf("\n");
	// wave-max model: groups of 30 consecutive tests (what a candidate pass holds); loop length = the longest walk in the group
	for (int lanes : {16, 30, 48}) {
		double mp = 0, mf = 0; size_t groups = 0;
		for (size_t i = 0; i + lanes <= trip_plain.size(); i += lanes) {
			int a = 0, b = 0;
			for (int k = 0; k < lanes; ++k) { a = std::max<int>(a, trip_plain[i + k]); b = std::max<int>(b, trip_filt[i + k]); }
			mp += a; mf += b; groups++;
		}
		printf("longest walk in groups of %d tests: %.2f -> %.2f with the two-slab pre-test\n", lanes, mp / groups, mf / groups);
	}
	return 0;
}
