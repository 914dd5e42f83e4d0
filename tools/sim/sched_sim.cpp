// sched_sim.cpp -- CPU model of the fused kernel's in-wave pass scheduler (trace.hip) on the real paths of bench config 2
// (analysis tool, not product, not a test).  Question it answers: how full do the passes get, and what
// does the frame cost in issue cycles, when a lane owns K paths instead of one ("K-slot lanes": a pass serves a lane if ANY of
// its K paths wants it; the path state lives outside the register file), at the occupancy the LDS then allows?
//
// Input: every ray of every path of the frame, in path order, from the oracle's canonical per-pixel render (orc_set_ray_probe),
// and the outcome of every 8^3 test (orc_set_brick_probe: cells walked, hit).  The walk itself is REPLAYED with the product's
// own arithmetic: the octant cube field of world.cpp, dda_jump of jump.h, single moves -- so jump passes, single moves and
// candidates occur exactly as in the kernel, under the kernel's policy (quorums, jump bursts, refill rule, ticket order).
// Machine model: 256 CUs x 4 SIMDs; each SIMD issues for one of its W resident waves at a time; a pass costs its wave-level
// VALU instruction count x CPI cycles of the SIMD's issue time, after which the wave waits LAT cycles per dependent global load
// the pass ends in (field byte; index word + brick) while the SIMD serves its other waves.  Frame time = the slowest SIMD.
//
// build: g++ -O2 -std=c++17 -ffp-contract=off -Ibrickmap_amd/csrc tools/sim/sched_sim.cpp brickmap_amd/csrc/world.cpp -Loracle -l:liboracle.so -Wl,-rpath,$PWD/oracle -lpthread -o scratch/sched_sim
// run:   scratch/sched_sim [key=value ...]   keys: K W qB qC lat cpi ovJ ovB ovC ovS sched refillmin tiles(1/n of the tile rows+cols) policy
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <queue>
#include <string>
#include <vector>

#include "jump.h"
#include "world.h"

extern "C" {
void* orc_world_create(int, int);
void orc_world_generate(void*, int);
void orc_world_reset_device(void*, int);
void orc_camera_direction(double, double, float*);
typedef void (*orc_brick_probe_t)(const float*, const float*, uint32_t, const uint32_t*, int, unsigned);
void orc_set_brick_probe(orc_brick_probe_t);
typedef void (*orc_ray_probe_t)(unsigned, int, int, const float*, const float*);
void orc_set_ray_probe(orc_ray_probe_t);
struct OrcCamera { float position[3], direction[3], up[3], focal, lens; };
struct OrcFrame { int width, height, spp, sample_base, max_bounces; unsigned base_frame; int primary_only, band_rows, shard_rank, shard_count; float sun_x, sun_y; };
double orc_render(void*, const OrcCamera*, const OrcFrame*, float*, uint32_t*, void*, int);
}

struct RayRec { float o[3], d[3]; uint32_t cand_first; uint16_t cand_count; uint8_t kind; };
struct CandRec { uint8_t steps, hit; };
static std::vector<RayRec> g_rays;
static std::vector<CandRec> g_cands;
static std::vector<uint32_t> g_path_first, g_path_count; // per pixel
static std::vector<uint32_t> g_tile_perm; // hand-out position -> 16x16-pixel tile (order=...: tiles sorted by the cost of their paths)
static void ray_probe(unsigned pixel, int, int kind, const float* o, const float* d) {
	RayRec r;
	memcpy(r.o, o, 12); memcpy(r.d, d, 12);
	r.cand_first = (uint32_t)g_cands.size(); r.cand_count = 0; r.kind = (uint8_t)kind;
	if (g_path_count[pixel] == 0) g_path_first[pixel] = (uint32_t)g_rays.size();
	g_path_count[pixel]++;
	g_rays.push_back(r);
}
static void brick_probe(const float*, const float*, uint32_t, const uint32_t*, int hit, unsigned steps) {
	g_cands.push_back(CandRec{(uint8_t)std::min(steps, 255u), (uint8_t)hit});
	g_rays.back().cand_count++;
}

static inline float gmin(float a, float b) { return (b < a) ? b : a; }
static inline float gmax(float a, float b) { return (a < b) ? b : a; }
static inline int isign(float x) { return (0.f < x) - (x < 0.f); }

enum { S_IDLE = 0, S_NEED = 1, S_OUTER = 2, S_CAND = 3, S_JUMP = 4 };

struct Params {
	int K = 1, W = 5;
	double qB = 0.5, qC = 0.25;         // quorums (fraction of live lanes)
	double lat = 600, cpi = 3.7;        // cycles per dependent global load; SIMD cycles per wave-level VALU instruction
	double cJ = 150, cS = 35, cB = 250, cBstep = 22, cC = 650, cSched = 40, cRefill = 110;
	double ovJ = 0, ovS = 0, ovB = 0, ovC = 0, schedMul = 1.0; // K-slot overheads (instructions per pass)
	int refill_min = 16, tiles = 1, jump_min = 4, jump_passes = 6, steps_per_round = 4;
	int policy = 0; // 0 = the kernel's quorum policy, 1 = greedy (most lanes per cost), 2 = quorum on slots instead of lanes
	int drain_merge = 0;
	int pool = 0;   // > 0: the waves of a workgroup share ONE pool of slots (K = layers per column); value = waves per workgroup
	int minfill = 0; // pool mode: a wave that finds fewer lanes than this for every pass type sleeps while other waves hold slots
	double beta = 0; // > 0: staged shutdown -- slot layer k is refilled only while the pixels left exceed k * beta * (lanes of the machine)
	int nwaves = 0;
	int twolaunch = 0; // 1: the hand-out order's first `split_tiles` tiles are one launch, the rest a second one whose workgroups start as the first's waves retire
	int split_tiles = 0;
	int split = 0; // 1: prepass launch (primary rays only, no shading), 2: the path launch that starts from the prepass's hit records
	double cGen = 320, cRec = 40; // prepass: primary ray + set-up; writing the hit record
	int rep = 1; // every pixel is handed out `rep` times (a steady-state / multi-sample frame: the drain is amortised)
	int spill = 0; // > 0: drain compaction through a global queue -- a wave out of tickets with <= spill live lanes writes its paths' records to the queue and retires;
	               // waves out of tickets with >= refill_min idle lanes take records from it; records nobody took are a follow-up launch's (modelled as a late pull)
	int spill_keep = 0; // waves still running below which nobody spills any more (the tail is latency, not issue)
	double cSpill = 120, cPull = 140;
	double refillLat = 1.5; // load latencies a refill stalls its wave for (the atomic's round trip); 0 models tickets fetched ahead of time
	double wB = 1.0, wC = 1.0; // policy 2: run the pass type with the largest (lanes * weight); the walk has weight 1
	int offload = 0; // round 5: 1 = a shade pass hands a path's SHADOW ray to an idle lane of the wave (if there is one) and the owner goes straight on
	                 // with its bounce ray (or ends): shadow and bounce rays walk side by side instead of one after the other.  K = 1 only.
	int reserve = 0;  // offload: a refill leaves this many idle lanes unfilled (kept for shadow rays)
	double cOff = 30; // instructions the exchange adds to a shade pass in which at least one ray is handed over
};

struct Slot {
	int st = S_IDLE;
	uint32_t next_ray = 0, ray_end = 0;
	const RayRec* ray = nullptr;
	uint32_t cand_i = 0;
	float tx, ty, tz, dx, dy, dz, ix, iy, iz;
	int px, py, pz, sx, sy, sz, oct;
	uint32_t cube = 0; bool nojump = false;
	bool busy = false; // claimed by a wave: invisible to the scans of the others until that wave's next round
	bool helper = false; // offload mode: this lane traces a shadow ray on behalf of another lane's path
};

struct World {
	bm::World w;
	std::vector<uint8_t> field;
	int cells, cells_h, cfx; size_t plane; float gs, gh;
	int F(int oct, int x, int y, int z) const {
		if (x < -1 || y < -1 || z < -1 || x > cells || y > cells || z > cells_h) return 255;
		return field[oct * plane + (size_t(z + 1) * cfx + (y + 1)) * cfx + (x + 1)];
	}
};
static World g_w;
static std::vector<uint8_t> g_coarse; // per octant and 4x4x4 block of (bordered coordinate + 15) >> 2: min of the fine bytes
static int g_cb = 0; static size_t g_cplane = 0;
static uint64_t g_lookups = 0, g_coarse_hits[4] = {0, 0, 0, 0}; // lookups answered by the coarse level for thresholds 2, 4, 8, 16

static int lookup(Slot& s, const Params& P) {
	const int v = g_w.F(s.oct, s.px, s.py, s.pz);
	if (!g_coarse.empty() && v != 255) {
		const int bx = (s.px + 16) >> 2, by = (s.py + 16) >> 2, bz = (s.pz + 16) >> 2;
		const int c = g_coarse[s.oct * g_cplane + (size_t(bz - 3) * g_cb + (by - 3)) * g_cb + (bx - 3)];
		g_lookups++;
		const int th[4] = {2, 4, 8, 16};
		for (int k = 0; k < 4; ++k) g_coarse_hits[k] += c >= th[k];
	}
	const bool possible = bm::jump_possible(s.tx, s.ty, s.tz);
	s.cube = (uint32_t)v; s.nojump = !possible;
	int st = (v >= P.jump_min && possible) ? S_JUMP : S_OUTER;
	if (v == 0) st = S_CAND;
	if (v == 255) st = S_NEED;
	return st;
}
static int setup(Slot& s, const RayRec& r, const Params& P) { // traverse.h ray_setup
	float ox = r.o[0], oy = r.o[1], oz = r.o[2];
	const float dx = r.d[0], dy = r.d[1], dz = r.d[2];
	const float gs = g_w.gs, gh = g_w.gh;
	float tminn = 0.f;
	const bool inside = ox > 0.f && ox < gs && oy > 0.f && oy < gs && oz > 0.f && oz < gh && (dx != 0.f || dy != 0.f || dz != 0.f) && dx == dx && dy == dy && dz == dz;
	if (!inside) {
		const float t1x = (0.f - ox) / dx, t1y = (0.f - oy) / dy, t1z = (0.f - oz) / dz, t2x = (gs - ox) / dx, t2y = (gs - oy) / dy, t2z = (gh - oz) / dz;
		tminn = gmax(gmax(gmin(t1x, t2x), 0.f), gmax(gmin(t1y, t2y), gmin(t1z, t2z)));
		if (!(gmin(gmax(t1x, t2x), gmin(gmax(t1y, t2y), gmax(t1z, t2z))) > tminn)) return S_NEED;
	}
	if (tminn > 0) {
		ox += dx * tminn; oy += dy * tminn; oz += dz * tminn;
		const float cx = gs / 2.f - ox, cy = gs / 2.f - oy, cz = gh / 2.f - oz;
		const float ax = fabsf(cx) * (1.f / (gs / gh)), ay = fabsf(cy) * (1.f / (gs / gh)), az = fabsf(cz) * 1.f;
		const float m = gmax(ax, gmax(ay, az));
		const float nx = float(isign(-cx)) * truncf(ax / m + 0.000001f), ny = float(isign(-cy)) * truncf(ay / m + 0.000001f), nz = float(isign(-cz)) * truncf(az / m + 0.000001f);
		ox -= nx * 0.001f; oy -= ny * 0.001f; oz -= nz * 0.001f;
	}
	ox /= 8.f; oy /= 8.f; oz /= 8.f;
	s.px = int(ox); s.py = int(oy); s.pz = int(oz);
	if (s.px < 0 || s.px >= g_w.cells || s.py < 0 || s.py >= g_w.cells || s.pz < 0 || s.pz >= g_w.cells_h) return S_NEED;
	s.sx = isign(dx); s.sy = isign(dy); s.sz = isign(dz);
	const float rx = dx == 0.f ? 0.f : 1.f / dx, ry = dy == 0.f ? 0.f : 1.f / dy, rz = dz == 0.f ? 0.f : 1.f / dz;
	s.tx = dx != 0.f ? ((dx > 0 ? float(s.px + 1) : float(s.px)) - ox) * rx : 1000000.f;
	s.ty = dy != 0.f ? ((dy > 0 ? float(s.py + 1) : float(s.py)) - oy) * ry : 1000000.f;
	s.tz = dz != 0.f ? ((dz > 0 ? float(s.pz + 1) : float(s.pz)) - oz) * rz : 1000000.f;
	s.dx = float(s.sx) * rx; s.dy = float(s.sy) * ry; s.dz = float(s.sz) * rz;
	s.ix = fabsf(dx); s.iy = fabsf(dy); s.iz = fabsf(dz);
	s.oct = (dx < 0 ? 1 : 0) | (dy < 0 ? 2 : 0) | (dz < 0 ? 4 : 0);
	s.cand_i = 0;
	return lookup(s, P);
}
static int single_step(Slot& s, const Params& P) {
	const bool mx = s.tx < s.ty && s.tx < s.tz, my = s.ty <= s.tx && s.ty < s.tz;
	if (mx) { s.px += s.sx; s.tx += s.dx; }
	else if (my) { s.py += s.sy; s.ty += s.dy; }
	else { s.pz += s.sz; s.tz += s.dz; }
	return lookup(s, P);
}
static int jump(Slot& s, const Params& P) {
	if (s.nojump) return single_step(s, P);
	uint32_t cx, cy, cz; int axis;
	bm::dda_jump(s.tx, s.ty, s.tz, s.dx, s.dy, s.dz, s.ix, s.iy, s.iz, s.cube & 0xFFu, cx, cy, cz, axis);
	s.px += int(cx) * s.sx; s.py += int(cy) * s.sy; s.pz += int(cz) * s.sz;
	return lookup(s, P);
}

struct Stats {
	double runs[5] = {}, lanes[5] = {}, instr[5] = {}; // J, S, B, C, sched(+refill)
	double brick_loop = 0, brick_passes = 0;
};

struct Seg { double issue, lat; };

struct Pool { std::vector<Slot> slots; }; // 64 * K, slot k of column l at [l * K + k]
struct Wave {
	Pool* pool = nullptr;
	std::vector<Slot*> held;
	std::vector<Seg> pending; size_t pend_i = 0;
	double ready = 0;
	int my_counter = 0, counters_done = 0;
	bool work_left = true, done = false;
	std::vector<uint32_t> mailbox; // offload=2: shadow rays waiting for an idle lane of this wave (ray indices)
	bool second = false; // two-launch model: this wave slot now runs a workgroup of the second launch
	double t_dry = -1, t_end = 0;
};

struct Sim {
	Params P;
	int W_img, H_img, tiles_x, tiles_y;
	uint32_t total_chunks;
	uint32_t counters[8] = {};
	uint32_t counters2[8] = {};
	Stats st;
	uint64_t paths_done = 0, rays_done = 0;
	std::vector<std::vector<Slot>> queue; // spill mode: published pages of path records (one page per spill)
	bool final_launch = false; // the follow-up launch: takes pages, spills nothing
	int waves_running = 0;
	uint64_t spilled = 0, pulled = 0, late = 0, spill_events = 0;
	uint64_t offloaded = 0;

	// one scheduler round of a wave: applies the functional effects now, returns the issue / latency segments it costs
	void round(Wave& w, double now) {
		const int K = P.K;
		std::vector<Slot>& slots = w.pool->slots;
		auto release = [&]() { for (Slot* s : w.held) s->busy = false; w.held.clear(); };
		auto claim = [&](Slot* s) { s->busy = true; w.held.push_back(s); };
		release();
		auto add = [&](int kind, double instr, double nlat) {
			st.instr[kind] += instr;
			w.pending.push_back(Seg{instr * P.cpi, nlat * P.lat});
		};
		// ---- refill
		int nI = 0, k_eff = K;
		if (P.beta > 0 && w.work_left) {
			const uint32_t total_groups0 = (total_chunks + 3u) >> 2;
			const double total_items = double(total_groups0) * 16.0 * P.rep;
			double handed = 0; for (int c = 0; c < 8; ++c) handed += counters[c];
			const double pixels_left = std::max(0.0, total_items - handed) * 4.0;
			k_eff = std::min(K, 1 + int(pixels_left / (P.beta * 64.0 * P.nwaves)));
		}
		for (int i = 0; i < (int)slots.size(); ++i) nI += (slots[i].st == S_IDLE && !slots[i].busy && (i % K) < k_eff);
		double sched_instr = P.cSched * P.schedMul;
		if (w.work_left && nI >= P.refill_min && !(P.offload == 2 && !w.mailbox.empty())) {
			if (P.pool && nI > 64) nI = 64; // one idle slot per column and refill
			const int want = std::max(1, (nI - P.reserve) / 4);
			const uint32_t total_groups = (total_chunks + 3u) >> 2;
			const uint32_t my_groups = total_groups > (uint32_t)w.my_counter ? (total_groups - w.my_counter + 7u) / 8u : 0u;
			const uint32_t my_tickets1 = my_groups * 4u * 4u;
			uint32_t my_tickets = my_tickets1 * (uint32_t)P.rep;
			uint32_t first_ticket = 0;
			if (P.twolaunch) {
				// tickets of this counter whose tile comes before split_tiles in the hand-out order: groups g = j * 8 + counter with (4 g) >> 4 < split_tiles
				const uint32_t split_groups = (uint32_t)P.split_tiles * 4u; // 4 groups of 4 chunks per tile
				const uint32_t ga = split_groups > (uint32_t)w.my_counter ? (split_groups - w.my_counter + 7u) / 8u : 0u;
				const uint32_t ta = std::min(ga, my_groups) * 16u;
				if (!w.second) my_tickets = ta; else first_ticket = ta;
			}
			uint32_t* ctr = w.second ? counters2 : counters;
			const uint32_t base = first_ticket + ctr[w.my_counter];
			ctr[w.my_counter] += (uint32_t)want;
			const uint32_t counter_now = (uint32_t)w.my_counter;
			if (base + want >= my_tickets) {
				w.my_counter = (w.my_counter + 1) % 8;
				if (++w.counters_done >= 8) { w.work_left = false; w.t_dry = now; }
			}
			int rank = 0;
			for (int si = 0; si < (int)slots.size(); ++si) {
				Slot& s = slots[si];
				if (s.st != S_IDLE || s.busy || (si % K) >= k_eff) continue;
				if (rank < want * 4) {
					const uint32_t item = base + (uint32_t)(rank / 4);
					const uint32_t item1 = item % my_tickets1;
					const uint32_t ticket = item1 / 4u, part = item1 % 4u;
					const uint32_t chunk = ((ticket >> 2) * 8u + counter_now) * 4u + (ticket & 3u);
					if (item < my_tickets && chunk < total_chunks) {
						const uint32_t tile0 = chunk >> 4, k = chunk & 15u;
						const uint32_t tile = g_tile_perm.empty() ? tile0 : g_tile_perm[tile0];
						const int tile_x = int(tile % (uint32_t)tiles_x), tile_y = int(tile / (uint32_t)tiles_x);
						const int cx = int((k & 1u) | ((k >> 1) & 2u)), cy = int(((k >> 1) & 1u) | ((k >> 2) & 2u));
						const uint32_t q = part * 4u + ((uint32_t)rank % 4u);
						const int x = tile_x * 16 + cx * 4 + int(q & 3u), y = tile_y * 16 + cy * 4 + int(q >> 2);
						if (x < W_img && y < H_img) {
							const uint32_t p = (uint32_t)y * W_img + x;
							if (g_path_count[p]) { // (pixels outside the sampled tiles have no path: the lane stays idle)
								s.st = S_NEED; s.next_ray = g_path_first[p]; s.ray_end = s.next_ray + g_path_count[p]; s.ray = nullptr;
								if (P.split == 1) s.ray_end = s.next_ray + 1;   // prepass: the primary ray only
								if (P.split == 2) { s.next_ray += 1; }          // path launch: the primary ray has been traced
								if (P.pool) claim(&s);
							}
						}
					}
				}
				rank++;
			}
			sched_instr += P.cRefill;
			add(4, sched_instr, P.refillLat); // the atomic's round trip
			sched_instr = 0;
			if (P.pool) return; // pool mode: a refill is a round of its own (the claimed slots are published at the next one)
		}
		if (P.spill && !w.work_left) {
			int nlive = 0, nidle = 0;
			for (auto& sl : slots) { nlive += sl.st != S_IDLE; nidle += sl.st == S_IDLE; }
			if (!final_launch && nlive > 0 && nlive <= P.spill && waves_running > P.spill_keep) {
				queue.emplace_back();
				for (auto& sl : slots) if (sl.st != S_IDLE) { queue.back().push_back(sl); sl.st = S_IDLE; sl.ray = nullptr; }
				spilled += nlive; spill_events++;
				add(4, P.cSpill, 1.0);
				waves_running--;
				w.done = true; return;
			}
			if (nidle >= P.spill && !queue.empty()) { // any page fits
				std::vector<Slot> page = std::move(queue.front()); queue.erase(queue.begin());
				for (auto& sl : slots) if (sl.st == S_IDLE && !page.empty()) { sl = page.back(); page.pop_back(); }
				pulled++; if (final_launch) late++;
				add(4, P.cPull, 2.5); // claim (atomic round trip), then the records
				return;
			}
			if (nlive == 0) { waves_running--; w.done = true; return; }
		}
		// ---- counts: columns in which ANY available slot wants the pass
		int nJ = 0, nO = 0, nB = 0, nC = 0, live = 0, busy_live = 0;
		for (int l = 0; l < 64; ++l) {
			bool j = false, o = false, b = false, c = false;
			for (int k = 0; k < K; ++k) {
				const Slot& sl = slots[l * K + k];
				if (sl.busy) { busy_live += sl.st != S_IDLE; continue; }
				const int s = sl.st;
				j |= s == S_JUMP; o |= s == S_OUTER; b |= s == S_CAND; c |= s == S_NEED;
			}
			nJ += j; nO += (o && !j); nB += b; nC += c; live += (j || o || b || c);
		}
		if (P.offload == 2 && !w.mailbox.empty()) { // idle lanes with mail waiting want a shade pass
			int idle_n = 0; for (auto& sl : slots) idle_n += (sl.st == S_IDLE && !sl.busy);
			const int take = std::min<int>(idle_n, (int)w.mailbox.size());
			nC += take; live += take;
			if (take == 0 && live == 0) live = 1; // (cannot happen: mail exists only while its senders' paths or other lanes live; guard)
		}
		const int nA = nJ + nO;
		if (live == 0 || (P.pool && busy_live > 0 && std::max(nA, std::max(nB, nC)) < P.minfill)) {
			if (live == 0 && busy_live == 0 && !w.work_left) {
				if (P.twolaunch && !w.second) { // the first launch's workgroup retires; one of the second launch takes its place
					w.second = true; w.work_left = true; w.counters_done = 0;
					add(4, 50, 8.0); // dispatch of a new workgroup: a few microseconds
					return;
				}
				w.done = true; return;
			}
			if (P.pool) { add(4, 12, 1.0); return; } // nothing (worth) running: s_sleep, look again later
			add(4, sched_instr, 0); return;
		}
		int phase;
		if (P.policy == 1) {
			// greedy: lanes served per instruction; the walk's value is per jump pass
			const double vA = nA / P.cJ, vB = nB / (P.cB + P.cBstep * 14), vC = nC / P.cC;
			phase = vC >= vA && vC >= vB ? 2 : (vB >= vA ? 1 : 0);
			if (phase == 0 && nA == 0) phase = nC >= nB ? 2 : 1;
		} else if (P.policy == 2) {
			const double vA = nA, vB = nB * P.wB, vC = nC * P.wC;
			phase = vC >= vA && vC >= vB ? 2 : (vB >= vA ? 1 : 0);
		} else {
			const int quorum = (int)std::ceil(live * P.qB), quorum_shade = (int)std::ceil(live * P.qC);
			if (nC >= quorum_shade) phase = 2;
			else if (nB >= quorum) phase = 1;
			else if (nA > 0) phase = 0;
			else phase = nC >= nB ? 2 : 1;
		}
		if (sched_instr > 0) add(4, sched_instr, P.pool ? 0.3 : 0); // pool mode: the scan's and the claim's LDS round trips
		auto pick = [&](int l, auto pred) -> Slot* {
			for (int k = 0; k < K; ++k) { Slot& s = slots[l * K + k]; if (!s.busy && pred(s.st)) return &s; }
			return nullptr;
		};
		if (phase == 2) {
			int n = 0, handed = 0;
			if (P.offload == 2) {
				for (int l2 = 0; l2 < 64 && !w.mailbox.empty(); ++l2) {
					Slot& sl = slots[l2];
					if (sl.st != S_IDLE || sl.busy) continue;
					sl.next_ray = w.mailbox.back(); w.mailbox.pop_back();
					sl.ray_end = sl.next_ray + 1; sl.ray = &g_rays[sl.next_ray++]; sl.helper = true;
					sl.st = setup(sl, *sl.ray, P);
					claim(&sl); n++;
				}
			}
			for (int l = 0; l < 64; ++l) {
				Slot* s = pick(l, [](int st) { return st == S_NEED; });
				if (!s) continue;
				n++;
				claim(s);
				if (s->ray) rays_done++;
				if (P.offload && K == 1 && s->next_ray < s->ray_end && g_rays[s->next_ray].kind == 1) {
					// the next ray is a shadow ray: an idle lane of the wave takes it, the owner continues with what follows
					Slot* idle = nullptr;
					for (int l2 = 0; l2 < 64 && !idle; ++l2) if (slots[l2].st == S_IDLE && !slots[l2].busy) idle = &slots[l2];
					if (!idle && P.offload == 2) { w.mailbox.push_back(s->next_ray); s->next_ray++; handed++; }
					if (idle) {
						idle->next_ray = s->next_ray; idle->ray_end = s->next_ray + 1; idle->ray = &g_rays[idle->next_ray++];
						idle->helper = true;
						idle->st = setup(*idle, *idle->ray, P);
						if (idle->st == S_IDLE) idle->st = S_NEED;
						claim(idle);
						s->next_ray++;
						handed++; n++;
					}
				}
				if (s->next_ray < s->ray_end) {
					s->ray = &g_rays[s->next_ray++];
					s->st = setup(*s, *s->ray, P);
				} else {
					s->st = S_IDLE; s->ray = nullptr; if (!s->helper) paths_done++; s->helper = false;
				}
			}
			offloaded += handed;
			st.runs[3]++; st.lanes[3] += n;
			add(3, (P.split == 1 ? P.cGen + P.cRec : P.cC + (P.split == 2 ? 90 : 0)) + P.ovC + (handed ? P.cOff : 0), 1);
		} else if (phase == 1) {
			int n = 0, longest = 0;
			for (int l = 0; l < 64; ++l) {
				Slot* s = pick(l, [](int st) { return st == S_CAND; });
				if (!s) continue;
				n++;
				claim(s);
				CandRec c{1, 0};
				if (s->cand_i < s->ray->cand_count) c = g_cands[s->ray->cand_first + s->cand_i];
				s->cand_i++;
				longest = std::max(longest, (int)c.steps);
				if (c.hit) s->st = S_NEED;
				else { s->st = S_OUTER; s->cube = 0; s->nojump = !bm::jump_possible(s->tx, s->ty, s->tz); }
			}
			st.runs[2]++; st.lanes[2] += n; st.brick_loop += longest; st.brick_passes++;
			add(2, P.cB + P.cBstep * longest + P.ovB, 2);
		} else {
			if (nJ * 4 >= nO) {
				int walkers = nA;
				for (int pass = 0; pass < P.jump_passes; ++pass) {
					int n = 0, still = 0;
					if (pass > 0) release(); // (the wave re-picks among its columns' slots for every pass of a burst)
					for (int l = 0; l < 64; ++l) {
						Slot* s = pick(l, [](int st) { return st == S_JUMP; });
						if (!s) s = pick(l, [](int st) { return st == S_OUTER; });
						if (!s) continue;
						n++;
						s->st = jump(*s, P);
						claim(s);
					}
					for (int l = 0; l < 64; ++l) {
						bool any = false;
						for (int k = 0; k < K && !any; ++k) { const Slot& sl = slots[l * K + k]; any = (sl.st == S_JUMP || sl.st == S_OUTER) && (!sl.busy || std::find(w.held.begin(), w.held.end(), &sl) != w.held.end()); }
						still += any;
					}
					st.runs[0]++; st.lanes[0] += n;
					add(0, P.cJ + P.ovJ + (pass > 0 && P.pool ? P.cSched * P.schedMul : 0), 1);
					if (still * 4 < walkers || still == 0) break;
					walkers = still;
				}
			} else {
				// single moves: the K-slot kernel loads the slot once per round (overhead once), then steps it
				std::vector<Slot*> sel(64, nullptr);
				for (int l = 0; l < 64; ++l) { sel[l] = pick(l, [](int st) { return st == S_OUTER; }); if (sel[l]) claim(sel[l]); }
				for (int k = 0; k < P.steps_per_round; ++k) {
					int n = 0;
					for (int l = 0; l < 64; ++l) {
						Slot* s = sel[l];
						if (!s || s->st != S_OUTER) continue;
						n++;
						s->st = single_step(*s, P);
					}
					st.runs[1]++; st.lanes[1] += n;
					add(1, P.cS + (k == 0 ? P.ovS : 0), 1);
				}
			}
		}
	}
};

int main(int argc, char** argv) {
	Params P;
	std::map<std::string, double*> dk = {{"qB", &P.qB}, {"qC", &P.qC}, {"lat", &P.lat}, {"cpi", &P.cpi}, {"ovJ", &P.ovJ}, {"ovS", &P.ovS}, {"ovB", &P.ovB}, {"ovC", &P.ovC},
										 {"sched", &P.schedMul}, {"cJ", &P.cJ}, {"cS", &P.cS}, {"cB", &P.cB}, {"cBstep", &P.cBstep}, {"cC", &P.cC}, {"cSched", &P.cSched}, {"cRefill", &P.cRefill}, {"wB", &P.wB}, {"wC", &P.wC}, {"beta", &P.beta}, {"cSpill", &P.cSpill}, {"cPull", &P.cPull}, {"refillLat", &P.refillLat}, {"cOff", &P.cOff}};
	std::map<std::string, int*> ik = {{"K", &P.K}, {"W", &P.W}, {"refillmin", &P.refill_min}, {"tiles", &P.tiles}, {"policy", &P.policy}, {"jumpmin", &P.jump_min},
									  {"jumppasses", &P.jump_passes}, {"steps", &P.steps_per_round}, {"rep", &P.rep}, {"pool", &P.pool}, {"minfill", &P.minfill}, {"split", &P.split}, {"twolaunch", &P.twolaunch}, {"spill", &P.spill}, {"spillkeep", &P.spill_keep}, {"splittiles", &P.split_tiles}, {"offload", &P.offload}, {"reserve", &P.reserve}};
	std::vector<std::string> sweeps;
	for (int i = 1; i < argc; ++i) {
		std::string a = argv[i];
		const size_t eq = a.find('=');
		if (eq == std::string::npos) { fprintf(stderr, "bad arg %s\n", argv[i]); return 1; }
		const std::string k = a.substr(0, eq), v = a.substr(eq + 1);
		if (k == "sweep") { sweeps.push_back(v); continue; }
		if (dk.count(k)) *dk[k] = atof(v.c_str());
		else if (ik.count(k)) *ik[k] = atoi(v.c_str());
		else { fprintf(stderr, "unknown key %s\n", k.c_str()); return 1; }
	}
	const int G = 1024, W = 1920, H = 1080;
	// ---- the frame's rays from the oracle
	void* ow = orc_world_create(G, G);
	orc_world_generate(ow, 8);
	orc_world_reset_device(ow, 1);
	OrcCamera cam{};
	cam.position[0] = G / 2.f; cam.position[1] = G / 8.f; cam.position[2] = 0.8f * G;
	orc_camera_direction(0.8, -0.5, cam.direction);
	cam.up[2] = 1.f; cam.focal = 1.f;
	OrcFrame f{};
	f.width = W; f.height = H; f.spp = 1; f.sample_base = 0; f.max_bounces = 3; f.base_frame = 1; f.band_rows = H; f.shard_rank = 0; f.shard_count = 1; f.sun_x = 0.05f; f.sun_y = 0.1f;
	g_path_first.assign(size_t(W) * H, 0); g_path_count.assign(size_t(W) * H, 0);
	std::vector<float> accum(size_t(W) * H * 4, 0.f);
	orc_set_ray_probe(ray_probe); orc_set_brick_probe(brick_probe);
	if (P.tiles > 1) {
		// sample: render only every tiles-th band of 16 rows (whole rows keep the ticket order's neighbourhoods intact)
		f.band_rows = 16; f.shard_count = P.tiles; f.shard_rank = 0;
	}
	orc_render(ow, &cam, &f, accum.data(), nullptr, nullptr, 1);
	orc_set_ray_probe(nullptr); orc_set_brick_probe(nullptr);
	size_t npaths = 0; for (auto c : g_path_count) npaths += c > 0;
	fprintf(stderr, "paths %zu rays %zu candidates %zu\n", npaths, g_rays.size(), g_cands.size());

	g_w.w.dims.set(G, G);
	g_w.w.generate(8);
	g_w.w.build_cube_field(g_w.field, 8);
	g_w.cells = g_w.w.dims.cells; g_w.cells_h = g_w.w.dims.cells_height; g_w.cfx = g_w.cells + 2; g_w.plane = g_w.field.size() / 8; g_w.gs = float(G); g_w.gh = float(G);

	if (getenv("BM_SIM_COARSE")) {
		g_cb = g_w.cells / 4 + 2; const int cbz = g_w.cells_h / 4 + 2;
		g_cplane = size_t(g_cb) * g_cb * cbz;
		g_coarse.assign(g_cplane * 8, 255);
		for (int o = 0; o < 8; ++o)
			for (int z = -1; z <= g_w.cells_h; ++z) for (int y = -1; y <= g_w.cells; ++y) for (int x = -1; x <= g_w.cells; ++x) {
				uint8_t& c = g_coarse[o * g_cplane + (size_t(((z + 16) >> 2) - 3) * g_cb + (((y + 16) >> 2) - 3)) * g_cb + (((x + 16) >> 2) - 3)];
				c = std::min<uint8_t>(c, (uint8_t)g_w.F(o, x, y, z));
			}
	}
	if (const char* ord = getenv("BM_SIM_ORDER")) {
		// hand-out order experiments: "lpt" = tiles by the number of rays of their paths, most expensive first (a list scheduler's
		// order); "skylast" = the normal order, but tiles whose every path is a single ray (sky) go to the end; "bottomup"
		const int tx = (W + 15) / 16, ty = (H + 15) / 16;
		std::vector<std::pair<double, uint32_t>> cost(size_t(tx) * ty);
		for (int t = 0; t < tx * ty; ++t) {
			double c = 0; int n = 0;
			for (int y = (t / tx) * 16; y < std::min(H, (t / tx) * 16 + 16); ++y) for (int x = (t % tx) * 16; x < std::min(W, (t % tx) * 16 + 16); ++x) { c += g_path_count[size_t(y) * W + x]; n++; }
			cost[t] = {n ? c / n : 0.0, uint32_t(t)};
		}
		g_tile_perm.resize(cost.size());
		if (!strcmp(ord, "lpt")) { std::stable_sort(cost.begin(), cost.end(), [](auto& a, auto& b) { return a.first > b.first; }); for (size_t i = 0; i < cost.size(); ++i) g_tile_perm[i] = cost[i].second; }
		else if (!strcmp(ord, "skylast")) { size_t k = 0; for (auto& c : cost) if (c.first > 1.0) g_tile_perm[k++] = c.second; P.split_tiles = (int)k; for (auto& c : cost) if (c.first <= 1.0) g_tile_perm[k++] = c.second; }
		else if (!strncmp(ord, "skytail:", 8)) { // the usual order, but the first K all-sky tiles of the sweep are handed out LAST: the paths that start last are one ray long
			const size_t K = (size_t)atoi(ord + 8); size_t k = 0, moved = 0; std::vector<uint32_t> tail;
			for (auto& c : cost) { if (c.first <= 1.0 && moved < K) { tail.push_back(c.second); moved++; } else g_tile_perm[k++] = c.second; }
			for (auto t : tail) g_tile_perm[k++] = t;
		}
		else if (!strcmp(ord, "bottomup")) { for (size_t i = 0; i < cost.size(); ++i) g_tile_perm[i] = uint32_t(cost.size() - 1 - i); }
		else if (!strcmp(ord, "skyfirst_terrain_lpt")) { size_t k = 0; for (auto& c : cost) if (c.first <= 1.0) g_tile_perm[k++] = c.second; std::vector<std::pair<double, uint32_t>> rest; for (auto& c : cost) if (c.first > 1.0) rest.push_back(c); std::stable_sort(rest.begin(), rest.end(), [](auto& a, auto& b) { return a.first > b.first; }); for (auto& c : rest) g_tile_perm[k++] = c.second; }
		else g_tile_perm.clear();
	}
	auto run = [&](const Params& P) {
		Sim sim; sim.P = P; sim.P.nwaves = P.W * (1024 / (P.tiles > 1 ? P.tiles : 1)); sim.W_img = W; sim.H_img = H; sim.tiles_x = (W + 15) / 16; sim.tiles_y = (H + 15) / 16;
		sim.total_chunks = uint32_t(sim.tiles_x) * sim.tiles_y * 16u;
		const int nsimd = 1024 / (P.tiles > 1 ? P.tiles : 1); // sampled frames run on a proportionally smaller machine
		std::vector<std::vector<Wave>> simd(nsimd);
		std::vector<Pool*> pools;
		// workgroups of 4 waves (pool mode: P.pool waves), wave j of a workgroup on SIMD j % 4 of its CU; wave index -> ticket counter as in the kernel
		const int ncu = nsimd / 4, per_wg = P.pool ? P.pool : 4, wg_per_cu = (P.W * 4) / per_wg;
		int block = 0;
		for (int g = 0; g < wg_per_cu; ++g)
			for (int cu = 0; cu < ncu; ++cu, ++block) {
				Pool* shared = nullptr;
				if (P.pool) { shared = new Pool; shared->slots.resize(64 * P.K); pools.push_back(shared); }
				for (int j = 0; j < per_wg; ++j) {
					Wave w;
					if (shared) w.pool = shared; else { w.pool = new Pool; w.pool->slots.resize(64 * P.K); pools.push_back(w.pool); }
					w.my_counter = (block * per_wg + j) % 8;
					simd[cu * 4 + (j % 4)].push_back(std::move(w));
				}
			}
		sim.waves_running = nsimd * P.W;
		struct Ev { double t; int s; bool operator<(const Ev& o) const { return t > o.t; } };
		std::vector<double> tnow(nsimd, 0.0);
		double t_last = 0, busy = 0, drain_sum = 0, life_sum = 0;
		auto run_launch = [&](double t0) {
			std::priority_queue<Ev> pq;
			for (int s = 0; s < nsimd; ++s) { tnow[s] = t0; pq.push(Ev{t0, s}); }
			while (!pq.empty()) {
				const Ev e = pq.top(); pq.pop();
				auto& waves = simd[e.s];
				// pick the ready wave with the earliest ready time
				Wave* best = nullptr;
				for (auto& w : waves) if (!w.done && (!best || w.ready < best->ready)) best = &w;
				if (!best) continue;
				double t = std::max(tnow[e.s], best->ready);
				if (best->pend_i >= best->pending.size()) {
					best->pending.clear(); best->pend_i = 0;
					sim.round(*best, t);
					if (best->done) { best->t_end = t; t_last = std::max(t_last, t); life_sum += t - t0; if (best->t_dry >= 0) drain_sum += t - best->t_dry; pq.push(Ev{t, e.s}); continue; }
				}
				if (best->pend_i < best->pending.size()) {
					const Seg sg = best->pending[best->pend_i++];
					t += sg.issue; busy += sg.issue;
					best->ready = t + sg.lat;
				}
				tnow[e.s] = t;
				pq.push(Ev{t, e.s});
			}
		};
		run_launch(0.0);
		const double t_first = t_last;
		size_t left_pages = sim.queue.size();
		if (P.spill && !sim.queue.empty()) { // the follow-up launch: the same grid, no tickets, takes what nobody took
			sim.final_launch = true;
			for (auto& ws : simd) for (auto& w : ws) { w.done = false; w.work_left = false; w.t_dry = t_last + 12000; w.ready = t_last + 12000; w.pending.clear(); w.pend_i = 0; }
			sim.waves_running = nsimd * P.W;
			run_launch(t_last + 12000); // ~5 us between the launches
		}
		const Stats& S = sim.st;
		const double scale = P.tiles > 1 ? P.tiles : 1; // report per full frame
		const double total_instr = S.instr[0] + S.instr[1] + S.instr[2] + S.instr[3] + S.instr[4];
		const char* nm[4] = {"J", "S", "B", "C"};
		printf("K=%d W=%d qB=%.2f qC=%.2f lat=%.0f policy=%d ov(J,S,B,C)=%.0f,%.0f,%.0f,%.0f  paths %llu rays %llu\n", P.K, P.W, P.qB, P.qC, P.lat, P.policy, P.ovJ, P.ovS, P.ovB, P.ovC,
			   (unsigned long long)sim.paths_done, (unsigned long long)sim.rays_done);
		double lane_instr = 0;
		for (int k = 0; k < 4; ++k) {
			printf("   %s passes %.3fM at %.1f lanes, %.1fM instr\n", nm[k], S.runs[k] * scale / 1e6, S.lanes[k] / std::max(1.0, S.runs[k]), S.instr[k] * scale / 1e6);
			lane_instr += S.instr[k] / std::max(1.0, S.runs[k]) * S.lanes[k];
		}
		printf("   brick loop %.1f cells;  sched+refill %.1fM instr;  total %.3fG wave instr, lanes/instr %.1f\n", S.brick_loop / std::max(1.0, S.brick_passes), S.instr[4] * scale / 1e6, total_instr * scale / 1e9,
			   lane_instr / (total_instr - S.instr[4]));
		if (P.offload) printf("   offload: %.3fM shadow rays handed to idle lanes\n", sim.offloaded * scale / 1e6);
		if (P.spill) printf("   spill <= %d lanes (keep %d waves): %llu records in %llu spills, %llu taken in-kernel, %zu pages left to the follow-up launch (first launch ends at %.3f ms)\n", P.spill, P.spill_keep, (unsigned long long)sim.spilled, (unsigned long long)sim.spill_events, (unsigned long long)(sim.pulled - sim.late), left_pages, t_first / 2.4e6);
		if (g_lookups) printf("   coarse level (min over 4x4x4 blocks): of %.1fM field lookups, block-min >= 2 / 4 / 8 / 16: %.1f %% / %.1f %% / %.1f %% / %.1f %%\n", g_lookups / 1e6,
							  100.0 * g_coarse_hits[0] / g_lookups, 100.0 * g_coarse_hits[1] / g_lookups, 100.0 * g_coarse_hits[2] / g_lookups, 100.0 * g_coarse_hits[3] / g_lookups);
		printf("   frame %.3f ms (slowest SIMD at 2.4 GHz), SIMD issue busy %.1f %%, drain %.1f %% of wave lifetime\n", t_last / 2.4e6, 100.0 * busy / (t_last * nsimd), 100.0 * drain_sum / life_sum);
		fflush(stdout);
	};
	if (sweeps.empty()) run(P);
	for (const std::string& sw : sweeps) { // sweep=K:W:qB:qC:ovJ:ovS:ovB:ovC[:sched]
		Params Q = P;
		double v[9] = {double(Q.K), double(Q.W), Q.qB, Q.qC, Q.ovJ, Q.ovS, Q.ovB, Q.ovC, Q.schedMul};
		int n = 0; size_t pos = 0;
		while (n < 9 && pos <= sw.size()) { size_t c = sw.find(':', pos); if (c == std::string::npos) c = sw.size(); if (c > pos) v[n] = atof(sw.substr(pos, c - pos).c_str()); n++; pos = c + 1; }
		Q.K = int(v[0]); Q.W = int(v[1]); Q.qB = v[2]; Q.qC = v[3]; Q.ovJ = v[4]; Q.ovS = v[5]; Q.ovB = v[6]; Q.ovC = v[7]; Q.schedMul = v[8];
		run(Q);
	}
	return 0;
}
