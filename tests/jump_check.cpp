// jump_check.cpp -- CPU replay test of brickmap_amd/csrc/jump.h (compiled and run by tests/test_jump.py).
// For random rays and random points along their walk, dda_jump() must land on exactly the state that the reference's
// one-cell-at-a-time stepping (src/voxel.cuh:249-258) reaches after the same number of steps: identical tmax bit
// patterns, identical per-axis step counts, identical final axis.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../brickmap_amd/csrc/jump.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() {
	rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
	return static_cast<uint32_t>(rng_state >> 32);
}
static float rndf() { return (rnd() >> 8) * (1.0f / 16777216.0f); }

struct Dda {
	float t[3], d[3];
	// one reference step (voxel.cuh:249-258); returns the axis
	int step() {
		const bool mx = t[0] < t[1] && t[0] < t[2];
		const bool my = t[1] <= t[0] && t[1] < t[2];
		const int a = mx ? 0 : (my ? 1 : 2);
		t[a] = t[a] + d[a];
		return a;
	}
};

static uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

int main(int argc, char** argv) {
	const long rays = argc > 1 ? std::atol(argv[1]) : 200000;
	long jumps = 0, exits = 0, steps_total = 0, skipped_pre = 0, failures = 0, multi = 0, multi_iters = 0;
	long len_hist[10] = {0};
	for (long ray = 0; ray < rays; ++ray) {
		// direction: mostly random unit vectors, some near-axis / near-diagonal / power-of-two-slope ones (ties)
		float dir[3];
		const uint32_t kind = rnd() % 8;
		for (int i = 0; i < 3; ++i) dir[i] = rndf() * 2.f - 1.f;
		if (kind == 0) dir[rnd() % 3] *= 1e-3f;
		if (kind == 1) { dir[0] = 1.f; dir[1] = 0.5f; dir[2] = 0.25f; }          // exact binary slopes: tdelta ties
		if (kind == 2) { dir[0] = dir[1]; }                                       // equal components: tmax ties
		if (kind == 3) { dir[rnd() % 3] = 0.f; }                                  // zero component: the 1e6 sentinel
		if (kind == 4) { dir[0] = std::ldexp(1.f, -static_cast<int>(rnd() % 12)); dir[1] = std::ldexp(3.f, -static_cast<int>(rnd() % 12)); }
		float len = std::sqrt((dir[0] * dir[0] + dir[1] * dir[1]) + dir[2] * dir[2]);
		if (len == 0.f) continue;
		if (kind != 1 && kind != 4) for (int i = 0; i < 3; ++i) dir[i] *= 1.0f / len;
		float o[3];
		for (int i = 0; i < 3; ++i) o[i] = (kind == 5 ? static_cast<float>(rnd() % 100) : rndf() * 100.f) + (kind == 6 ? 0.5f : 0.f);
		Dda s;
		for (int i = 0; i < 3; ++i) { // voxel.cuh:166-187
			const int p = static_cast<int>(o[i]);
			const float cb = dir[i] > 0.f ? static_cast<float>(p + 1) : static_cast<float>(p);
			const float rdinv = dir[i] == 0.f ? 0.f : 1.f / dir[i];
			const float sgn = static_cast<float>((0.f < dir[i]) - (dir[i] < 0.f));
			s.t[i] = dir[i] != 0.f ? (cb - o[i]) * rdinv : 1000000.f;
			s.d[i] = sgn * rdinv;
		}
		// walk; at random points try a jump with a random cube edge
		const int walk = 1 + static_cast<int>(rnd() % (ray % 8 == 0 ? 3000 : 600)); // up to 3 x 1026 cells: the longest walk of a supported world
		for (int k = 0; k < walk; ++k) {
			if (rnd() % 4 == 0) {
				if (!bm::jump_possible(s.t[0], s.t[1], s.t[2])) { skipped_pre++; }
				else {
					const uint32_t nsel = rnd() % 4;
					const uint32_t n = nsel == 0 ? 1 + rnd() % 4 : (nsel == 1 ? 1 + rnd() % 32 : 1 + rnd() % 254);
					float jt[3] = {s.t[0], s.t[1], s.t[2]};
					uint32_t c[3];
					int last = -1;
					// |direction| as the fused kernel passes it, or recovered from tdelta as the queue kernels do (perturbed by an ulp like a hardware reciprocal)
					float inv[3];
					const bool from_delta = rnd() % 2 == 0;
					for (int i = 0; i < 3; ++i) {
						inv[i] = from_delta ? (s.d[i] > 0.f ? 1.0f / s.d[i] : 0.f) : std::fabs(dir[i]);
						if (from_delta && inv[i] > 0.f) inv[i] = std::nextafter(inv[i], (rnd() & 1) ? 0.f : 2.f * inv[i]);
					}
					const bool exited = bm::dda_jump(jt[0], jt[1], jt[2], s.d[0], s.d[1], s.d[2], inv[0], inv[1], inv[2], n, c[0], c[1], c[2], last);
					Dda r = s;
					uint32_t rc[3] = {0, 0, 0};
					int rlast = -1;
					const uint32_t total = c[0] + c[1] + c[2];
					bool ok = total >= 1 && total < 100000;
					for (uint32_t q = 0; ok && q < total; ++q) { rlast = r.step(); rc[rlast]++; }
					for (int i = 0; i < 3; ++i) ok = ok && rc[i] == c[i] && bits(r.t[i]) == bits(jt[i]) && c[i] <= n;
					if (exited) ok = ok && rlast == last && (c[last] == n || true);
					// a jump must not step past the cube: at most one axis reaches n, and only as the final step
					int at_n = 0;
					for (int i = 0; i < 3; ++i) at_n += c[i] == n;
					ok = ok && (at_n == 0 || (exited && at_n == 1 && c[last] == n));
					if (!ok) {
						if (failures < 10)
							std::fprintf(stderr, "MISMATCH ray %ld n %u t=(%a %a %a) d=(%a %a %a) jump c=(%u %u %u) last %d exited %d -> t=(%a %a %a); replay c=(%u %u %u) last %d t=(%a %a %a)\n",
										 ray, n, s.t[0], s.t[1], s.t[2], s.d[0], s.d[1], s.d[2], c[0], c[1], c[2], last, exited, jt[0], jt[1], jt[2], rc[0], rc[1], rc[2], rlast,
										 r.t[0], r.t[1], r.t[2]);
						failures++;
					}
					// ---- a jump that stopped at a binade end may go on with what is left of the budget on every axis (dda_jump3,
					// traverse.h field_jump with BM_JUMP_BINADES > 1): up to three more binades, then the same comparison
					if (ok && !exited && rnd() % 2 == 0) {
						uint32_t b[3] = {n - c[0], n - c[1], n - c[2]}, tc[3] = {c[0], c[1], c[2]};
						bool ex2 = false;
						int last2 = -1;
						for (int it = 0; it < 3 && !ex2 && bm::jump_possible(jt[0], jt[1], jt[2]); ++it) {
							uint32_t a[3];
							ex2 = bm::dda_jump3(jt[0], jt[1], jt[2], s.d[0], s.d[1], s.d[2], inv[0], inv[1], inv[2], b[0], b[1], b[2], a[0], a[1], a[2], last2);
							for (int i = 0; i < 3; ++i) { tc[i] += a[i]; ok = ok && a[i] <= b[i]; b[i] -= a[i]; }
							multi_iters++;
						}
						Dda r2 = s;
						uint32_t rc2[3] = {0, 0, 0};
						int rlast2 = -1;
						const uint32_t total2 = tc[0] + tc[1] + tc[2];
						for (uint32_t q = 0; ok && q < total2; ++q) { rlast2 = r2.step(); rc2[rlast2]++; }
						int at_n2 = 0;
						for (int i = 0; i < 3; ++i) { ok = ok && rc2[i] == tc[i] && bits(r2.t[i]) == bits(jt[i]) && tc[i] <= n; at_n2 += tc[i] == n; }
						if (ex2) ok = ok && rlast2 == last2;
						ok = ok && (at_n2 == 0 || (ex2 && at_n2 == 1 && tc[last2] == n)); // never past the cube
						if (!ok) {
							if (failures < 10) std::fprintf(stderr, "MISMATCH (continued jump) ray %ld n %u c=(%u %u %u) replay=(%u %u %u)\n", ray, n, tc[0], tc[1], tc[2], rc2[0], rc2[1], rc2[2]);
							failures++;
						}
						multi++;
					}
					jumps++;
					exits += exited;
					steps_total += total;
					int l = 0;
					while ((1u << l) < total && l < 9) l++;
					len_hist[l]++;
				}
			}
			s.step();
		}
	}
	std::printf("jumps %ld exits %ld steps %ld (%.1f per jump) precondition-skips %ld failures %ld\n", jumps, exits, steps_total, jumps ? double(steps_total) / jumps : 0.0,
				skipped_pre, failures);
	std::printf("continued jumps %ld (%ld further binades)\n", multi, multi_iters);
	std::printf("jump length histogram (<=1,2,4,...):");
	for (int i = 0; i < 10; ++i) std::printf(" %ld", len_hist[i]);
	std::printf("\n");
	return failures ? 1 : 0;
}
