// trace.hip -- gfx950 path-trace kernels for the brickmap (hand-written HIP, wave64).
//
// One thread per pixel, one complete path per (pixel, sample): the reference's four wavefront
// kernels primary_rays / extend / shade / connect (src/kernel.cu:154-346) are fused into one
// launch that keeps the 64-byte RayQueue record (variables.h:43-52) in registers, and the
// brick-grid DDA of src/voxel.cuh:135-261 walks a flat supercell-blocked index grid instead
// of the reference's pointer table.  Per-ray arithmetic follows the reference operation for
// operation (IEEE fp32, no contraction; see DESIGN.md "Numeric contract") so that hits are
// bit-identical to the CPU oracle.
//
// Launch geometry: 256-thread workgroups = 16x16 pixel tiles, each wave an 8x8 pixel block
// (coherent primary rays).  Workgroup b runs on XCD b%8 (observed dispatch rule), so tile
// columns are dealt to XCDs in contiguous vertical stripes: every XCD's private 4 MiB L2 then
// caches one wedge of the view frustum, and every XCD sees the same sky/terrain mix.
#include <hip/hip_runtime.h>

#include "detmath.h"
#include "device_types.h"

namespace bm {

namespace {

constexpr float kPi = 3.1415926535897932f;       // variables.h:3
constexpr float kEpsilon = 0.001f;               // variables.h:22
constexpr uint32_t kIndexBits = 0xFFFu;          // variables.h:29-33
constexpr uint32_t kLodBits = 0xFF000u;
constexpr uint32_t kLoadedBit = 0x80000000u;
constexpr uint32_t kUnloadedBit = 0x40000000u;
constexpr uint32_t kRequestedBit = 0x20000000u;

struct f3 {
	float x, y, z;
};
__device__ __forceinline__ f3 mk(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ f3 operator/(f3 a, f3 b) { return mk(a.x / b.x, a.y / b.y, a.z / b.z); }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f3 operator/(f3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
// GLM forms: min(a,b) = (b<a)?b:a, max(a,b) = (a<b)?b:a, sign(x) = (0<x)-(x<0)
__device__ __forceinline__ float gmin(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float gmax(float a, float b) { return (a < b) ? b : a; }
__device__ __forceinline__ int isign(float x) { return (0.f < x) - (x < 0.f); }
__device__ __forceinline__ float dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ f3 cross(f3 x, f3 y) {
	return mk(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
__device__ __forceinline__ f3 normalize(f3 v) { return v * (1.0f / sqrtf(dot(v, v))); }
__device__ __forceinline__ f3 ld3(const float* p) { return mk(p[0], p[1], p[2]); }

// ---- RNG (kernel.cu:19-37)
__device__ __forceinline__ uint32_t random_int(uint32_t& seed) {
	seed ^= seed << 13;
	seed ^= seed >> 17;
	seed ^= seed << 5;
	return seed;
}
__device__ __forceinline__ float random_float(uint32_t& seed) { return random_int(seed) * 2.3283064365387e-10f; }
__device__ __forceinline__ float random_float2(uint32_t& seed) { return (random_int(seed) >> 16) / 65535.0f; }

// per-thread traversal counters (instrumented variant only)
struct Tally {
	uint32_t index_loads = 0, brick_tests = 0, byte_tests = 0, voxel_steps = 0, extend_rays = 0, shadow_rays = 0, requests = 0, paths = 0;
};
struct HitInfo {
	int level = 0, brick_id = -1, sub_id = 0;
};

// ---- 8^3 bitmask DDA (voxel.cuh:79-133) and 2^3 LoD DDA (voxel.cuh:26-77): one body, N = 8 or 2.
// `brick` holds the 64-byte brick (N == 8); `byte` is the LoD mask from the index word (N == 2).
// The brick is fetched once, as four 16-byte loads in flight together, and staged in LDS: a z-slice of the
// brick is exactly one 64-bit word (bit x + 8y), re-read only when the walk changes z.
struct BrickRegs {
	uint4 q0, q1, q2, q3;
};

// Brick staging in LDS: the 64-byte bitmask of the brick under test is written to the workgroup's LDS once and the
// walk re-reads one 8-byte z-slice whenever it changes z.  Slice z of thread t lives at lds_brick[z * 256 + t]:
// consecutive lanes hit consecutive 8-byte slots, so the eight stores and the per-z-move loads are free of bank
// conflicts whatever z each lane wants (a 14-instruction register select tree per z-move measured 5 % slower).
__device__ __forceinline__ void brick_to_lds(unsigned long long* lds_brick, const BrickRegs& b) {
	const int t = threadIdx.x;
	lds_brick[0 * 256 + t] = static_cast<unsigned long long>(b.q0.x) | (static_cast<unsigned long long>(b.q0.y) << 32);
	lds_brick[1 * 256 + t] = static_cast<unsigned long long>(b.q0.z) | (static_cast<unsigned long long>(b.q0.w) << 32);
	lds_brick[2 * 256 + t] = static_cast<unsigned long long>(b.q1.x) | (static_cast<unsigned long long>(b.q1.y) << 32);
	lds_brick[3 * 256 + t] = static_cast<unsigned long long>(b.q1.z) | (static_cast<unsigned long long>(b.q1.w) << 32);
	lds_brick[4 * 256 + t] = static_cast<unsigned long long>(b.q2.x) | (static_cast<unsigned long long>(b.q2.y) << 32);
	lds_brick[5 * 256 + t] = static_cast<unsigned long long>(b.q2.z) | (static_cast<unsigned long long>(b.q2.w) << 32);
	lds_brick[6 * 256 + t] = static_cast<unsigned long long>(b.q3.x) | (static_cast<unsigned long long>(b.q3.y) << 32);
	lds_brick[7 * 256 + t] = static_cast<unsigned long long>(b.q3.z) | (static_cast<unsigned long long>(b.q3.w) << 32);
}

template <int N, bool DBG>
__device__ __forceinline__ bool intersect_grid(f3 origin, f3 dir, int sx, int sy, int sz, float dx, float dy, float dz, f3& normal, float& distance,
											   const BrickRegs& brick, uint32_t byte, int& sub_id, Tally& tally, unsigned long long* lds_brick = nullptr) {
	int px = static_cast<int>(origin.x), py = static_cast<int>(origin.y), pz = static_cast<int>(origin.z);
	const float cbx = dir.x > 0.f ? static_cast<float>(px + 1) : static_cast<float>(px);
	const float cby = dir.y > 0.f ? static_cast<float>(py + 1) : static_cast<float>(py);
	const float cbz = dir.z > 0.f ? static_cast<float>(pz + 1) : static_cast<float>(pz);
	// rdinv = sign * |1/d| exactly (0 for d == 0)
	const float rx = static_cast<float>(sx) * dx, ry = static_cast<float>(sy) * dy, rz = static_cast<float>(sz) * dz;
	float tx = dir.x != 0.f ? (cbx - origin.x) * rx : 1000000.f;
	float ty = dir.y != 0.f ? (cby - origin.y) * ry : 1000000.f;
	float tz = dir.z != 0.f ? (cbz - origin.z) * rz : 1000000.f;
	px %= N; py %= N; pz %= N;
	distance = 0.f;
	int axis = -1;
	// "& 7" / "& 63" only define what the reference leaves undefined (a negative start cell); no effect otherwise
	if (N == 8) brick_to_lds(lds_brick, brick);
	unsigned long long slice = N == 8 ? lds_brick[(pz & 7) * 256 + threadIdx.x] : 0ull;
	// at most 3N-2 cells lie on a line through an N^3 block; the bound only guards against NaN input
	bool found = false;
	for (int guard = 0; guard < 3 * N + 2; ++guard) {
		if (DBG) tally.voxel_steps++;
		bool solid;
		if (N == 8) solid = (slice >> ((px + py * 8) & 63)) & 1ull;
		else solid = (byte >> ((px + py * 2 + pz * 4) & 31)) & 1u;
		if (solid) { found = true; break; } // resolved after the loop, once, for all lanes that hit
		// select-style move (voxel.cuh:122-130); `t += mask ? delta : 0` is `tmax += mask * tdelta` for finite deltas
		const bool mx = tx < ty && tx < tz;
		const bool my = ty <= tx && ty < tz; // mx implies !my
		const bool mz = !(mx || my);
		axis = mx ? 0 : (my ? 1 : 2);
		px += mx ? sx : 0;
		py += my ? sy : 0;
		pz += mz ? sz : 0;
		const int c = mx ? px : (my ? py : pz);
		const int s_sel = mx ? sx : (my ? sy : sz);
		if (c == (s_sel > 0 ? N : -1)) break; // left the block
		tx += mx ? dx : 0.f;
		ty += my ? dy : 0.f;
		tz += mz ? dz : 0.f;
		if (N == 8 && mz) slice = lds_brick[pz * 256 + threadIdx.x];
	}
	if (!found) return false;
	if (axis > -1) { // voxel.cuh:114-118; a hit in the very first cell keeps distance 0 and the caller's normal
		normal = mk(axis == 0 ? -static_cast<float>(sx) : 0.f, axis == 1 ? -static_cast<float>(sy) : 0.f, axis == 2 ? -static_cast<float>(sz) : 0.f);
		distance = axis == 0 ? tx - dx : (axis == 1 ? ty - dy : tz - dz);
	}
	sub_id = px + py * N + pz * N * N;
	return true;
}

// ---- brick-grid DDA (voxel.cuh:135-261), split into the three pieces the wave scheduler interleaves
//
// The reference reads one 32-bit index word per visited cell (two dependent loads through its pointer
// table).  ~96 % of those words are zero (air), so the walk consults a two-level occupancy summary kept
// in registers -- a 64-bit mask of the current 4x4x4-brick block and a 64-bit mask of the current
// supercell's blocks (DeviceScene) -- and touches the index grid only at cells known to be non-empty.
// Masks are re-read only when the walk crosses a block / supercell boundary, and because the world edge
// is a supercell boundary the reference's per-step exit test (voxel.cuh:256) moves into that rare path too.
// The per-cell arithmetic (axis choice, tmax accumulation) is the reference's, step for step.
struct RayState {
	f3 o, d;            // origin (brick units once set up) and direction
	float tx, ty, tz;   // tmax
	float dx, dy, dz;   // tdelta = |1/d|
	int px, py, pz;     // current brick cell
	int sx, sy, sz;     // step signs
	float tminn;
	f3 n;               // normal carried in/out of the traversal (voxel.cuh:135 `normal`)
	int axis;           // axis of the last move, -1 before the first
	unsigned long long coarse, fine;
	uint32_t block_base; // arena slot of the current block's first brick
	int sci;
	float distance;     // result
	bool hit;
};

enum : int { ST_NEED = 0, ST_OUTER = 1, ST_CAND = 2, ST_FIN = 3 };

__device__ __forceinline__ void load_super(const DeviceScene& sc, RayState& r) {
	r.sci = (r.px >> 4) + (r.py >> 4) * sc.sg_xy + (r.pz >> 4) * sc.sg_xy2;
	const uint2 rec = *reinterpret_cast<const uint2*>(sc.super_info + r.sci);
	r.coarse = static_cast<unsigned long long>(rec.x) | (static_cast<unsigned long long>(rec.y) << 32);
}
__device__ __forceinline__ void load_block(const DeviceScene& sc, RayState& r) {
	const int bi = ((r.px >> 2) & 3) + (((r.py >> 2) & 3) << 2) + (((r.pz >> 2) & 3) << 4);
	r.fine = 0ull;
	if ((r.coarse >> bi) & 1ull) {
		const uint4 rec = *reinterpret_cast<const uint4*>(sc.block_info + (static_cast<size_t>(r.sci) << 6) + bi);
		r.fine = static_cast<unsigned long long>(rec.x) | (static_cast<unsigned long long>(rec.y) << 32);
		r.block_base = rec.z;
	}
}
__device__ __forceinline__ bool cell_occupied(const RayState& r) {
	const int ci = (r.px & 3) + ((r.py & 3) << 2) + ((r.pz & 3) << 4);
	return (r.fine >> ci) & 1ull;
}

// voxel.cuh:136-189: clip against the world box, move onto it, set up the Amanatides-Woo state.
// Returns the lane's next state: ST_OUTER / ST_CAND, or ST_NEED with r.hit = false when the ray misses the box.
template <bool DBG>
__device__ __forceinline__ int ray_setup(const DeviceScene& sc, f3 origin, const f3 dir, RayState& r, Tally& tally) {
	r.hit = false;
	r.d = dir;
	// intersect_aabb_branchless2 (voxel.cuh:13-24).  For an origin strictly inside the box every slab entry time is
	// negative and every exit time positive, so the reference's result is exactly (true, tmin = 0): the six IEEE
	// divisions are only needed for rays that start on or outside the boundary.
	float tminn = 0.f;
	const bool inside = origin.x > 0.f && origin.x < sc.grid_size_f && origin.y > 0.f && origin.y < sc.grid_size_f && origin.z > 0.f &&
						origin.z < sc.grid_height_f && (dir.x != 0.f || dir.y != 0.f || dir.z != 0.f) &&
						dir.x == dir.x && dir.y == dir.y && dir.z == dir.z; // NaN directions (bounce off a zero normal) take the full test
	if (!inside) {
		const f3 t1 = (mk(0.f, 0.f, 0.f) - origin) / dir;
		const f3 t2 = (mk(sc.grid_size_f, sc.grid_size_f, sc.grid_height_f) - origin) / dir;
		const f3 tMin = mk(gmin(t1.x, t2.x), gmin(t1.y, t2.y), gmin(t1.z, t2.z));
		const f3 tMax = mk(gmax(t1.x, t2.x), gmax(t1.y, t2.y), gmax(t1.z, t2.z));
		tminn = gmax(gmax(tMin.x, 0.f), gmax(tMin.y, tMin.z));
		if (!(gmin(tMax.x, gmin(tMax.y, tMax.z)) > tminn)) return ST_NEED;
	}
	r.tminn = tminn;
	if (tminn > 0) { // move the ray onto the box and derive the entry-face normal (voxel.cuh:142-155)
		origin = origin + dir * tminn;
		const float gs = sc.grid_size_f, gh = sc.grid_height_f;
		const f3 scale = mk(1.f / (gs / gh), 1.f / (gs / gh), 1.f / (gh / gh));
		const f3 center = mk(gs / 2.f, gs / 2.f, gh / 2.f);
		const f3 d = center - origin;
		f3 to_center = mk(fabsf(d.x), fabsf(d.y), fabsf(d.z)) * scale;
		const f3 e = origin - center;
		const f3 signs = mk(static_cast<float>(isign(e.x)), static_cast<float>(isign(e.y)), static_cast<float>(isign(e.z)));
		to_center = to_center / gmax(to_center.x, gmax(to_center.y, to_center.z));
		r.n = signs * mk(truncf(to_center.x + 0.000001f), truncf(to_center.y + 0.000001f), truncf(to_center.z + 0.000001f));
		origin = origin - r.n * kEpsilon;
	}
	origin = origin / 8.f;
	r.o = origin;
	r.px = static_cast<int>(origin.x); r.py = static_cast<int>(origin.y); r.pz = static_cast<int>(origin.z);
	const int cells = sc.cells, cells_h = sc.cells_height;
	if (r.px < 0 || r.px >= cells || r.py < 0 || r.py >= cells || r.pz < 0 || r.pz >= cells_h) return ST_NEED;
	const float cbx = dir.x > 0.f ? static_cast<float>(r.px + 1) : static_cast<float>(r.px);
	const float cby = dir.y > 0.f ? static_cast<float>(r.py + 1) : static_cast<float>(r.py);
	const float cbz = dir.z > 0.f ? static_cast<float>(r.pz + 1) : static_cast<float>(r.pz);
	r.sx = isign(dir.x); r.sy = isign(dir.y); r.sz = isign(dir.z);
	const float rx = dir.x == 0.0f ? 0.0f : 1.f / dir.x;
	const float ry = dir.y == 0.0f ? 0.0f : 1.f / dir.y;
	const float rz = dir.z == 0.0f ? 0.0f : 1.f / dir.z;
	r.tx = dir.x != 0.f ? (cbx - origin.x) * rx : 1000000.f;
	r.ty = dir.y != 0.f ? (cby - origin.y) * ry : 1000000.f;
	r.tz = dir.z != 0.f ? (cbz - origin.z) * rz : 1000000.f;
	r.dx = static_cast<float>(r.sx) * rx; r.dy = static_cast<float>(r.sy) * ry; r.dz = static_cast<float>(r.sz) * rz;
	r.axis = -1;
	load_super(sc, r);
	load_block(sc, r);
	if (DBG) tally.index_loads++; // one per visited cell = the reference's index loads (algorithmic count)
	return cell_occupied(r) ? ST_CAND : ST_OUTER;
}

// voxel.cuh:249-258: one Amanatides-Woo move to the next cell.  Returns the next state.
// Written select-style (no per-axis branches): the only divergent region is the block / supercell
// boundary crossing.  `t += mask ? delta : 0` is the reference's `tmax += mask * tdelta` for finite deltas.
template <bool DBG>
__device__ __forceinline__ int outer_step(const DeviceScene& sc, RayState& r, Tally& tally) {
	// work on scalar copies: selects between struct members would otherwise pin the struct in scratch memory
	const float tx = r.tx, ty = r.ty, tz = r.tz;
	const int sx = r.sx, sy = r.sy, sz = r.sz;
	const bool mx = tx < ty && tx < tz;
	const bool my = ty <= tx && ty < tz; // mx implies !my
	const bool mz = !(mx || my);
	const int npx = r.px + (mx ? sx : 0);
	const int npy = r.py + (my ? sy : 0);
	const int npz = r.pz + (mz ? sz : 0);
	r.px = npx; r.py = npy; r.pz = npz;
	r.tx = tx + (mx ? r.dx : 0.f);
	r.ty = ty + (my ? r.dy : 0.f);
	r.tz = tz + (mz ? r.dz : 0.f);
	r.axis = mx ? 0 : (my ? 1 : 2);
	const int s_sel = mx ? sx : (my ? sy : sz);
	const int c = mx ? npx : (my ? npy : npz);
	// a move along -axis crosses a 4- / 16-aligned boundary when the NEW coordinate + 1 is aligned
	const int ce = c - (s_sel >> 31);
	if ((ce & 3) == 0) {
		if ((ce & 15) == 0) {
			// supercell boundary; the world edge is one of them, so the exit test (voxel.cuh:256) lives here
			const int lim = mz ? sc.cells_height : sc.cells;
			if (c == (s_sel > 0 ? lim : -1)) return ST_NEED; // left the grid: miss (r.hit stays false)
			load_super(sc, r);
		}
		load_block(sc, r);
	}
	if (DBG) tally.index_loads++;
	return cell_occupied(r) ? ST_CAND : ST_OUTER;
}

// voxel.cuh:200-247: the current cell holds a non-empty brick -- read its index word and resolve it.
template <bool DBG>
__device__ __forceinline__ int process_candidate(const DeviceScene& sc, const int* campos, RayState& r, HitInfo& info, Tally& tally,
													 unsigned long long* lds_brick) {
	const int px = r.px, py = r.py, pz = r.pz;
	// inside the grid 0 <= pos < cells, so >>4 and &15 equal the reference's signed /16 and %16
	const uint32_t flat = (static_cast<uint32_t>(r.sci) << 12) + static_cast<uint32_t>((px & 15) + ((py & 15) << 4) + ((pz & 15) << 8));
	// Home slot of this brick: block base + rank of its bit in the block mask.  It does not depend on the index
	// word, so the 64-byte brick read is issued together with the index-word read instead of behind it (every
	// non-empty cell owns its slot whether or not the brick is resident, so the read is always in bounds).
	const int ci = (px & 3) + ((py & 3) << 2) + ((pz & 3) << 4);
	const uint32_t slot = r.block_base + static_cast<uint32_t>(__popcll(r.fine & ((1ull << ci) - 1ull)));
	const uint4* bq = reinterpret_cast<const uint4*>(sc.brick_arena + (static_cast<size_t>(slot) << 4));
	const uint32_t index = sc.index_grid[flat];
	BrickRegs brick;
	brick.q0 = bq[0]; brick.q1 = bq[1]; brick.q2 = bq[2]; brick.q3 = bq[3];
	float new_distance = 0.f;
	if (r.axis != -1) {
		r.n = mk(0.f, 0.f, 0.f);
		if (r.axis == 0) { r.n.x = -static_cast<float>(r.sx); new_distance = r.tx - r.dx; }
		else if (r.axis == 1) { r.n.y = -static_cast<float>(r.sy); new_distance = r.ty - r.dy; }
		else { r.n.z = -static_cast<float>(r.sz); new_distance = r.tz - r.dz; }
	}
	const int ddx = campos[0] - px, ddy = campos[1] - py, ddz = campos[2] - pz;
	const int lod2 = ddx * ddx + ddy * ddy + ddz * ddz;
	float sub_distance = 0.f;
	if (DBG) info.brick_id = px + py * sc.cells + pz * sc.cells * sc.cells;
	if (lod2 > sc.lod_distance_8x8x8) {
		r.distance = new_distance * 8.f + r.tminn;
		if (DBG) { info.level = 0; info.sub_id = 0; }
		r.hit = true;
		return ST_NEED;
	} else if (lod2 > sc.lod_distance_2x2x2) {
		if (DBG) tally.byte_tests++;
		int sub = 0;
		const f3 o2 = (r.o + r.d * new_distance) * 2.f - r.n * 0.2f * kEpsilon;
		if (intersect_grid<2, DBG>(o2, r.d, r.sx, r.sy, r.sz, r.dx, r.dy, r.dz, r.n, sub_distance, brick, (index & kLodBits) >> 12, sub, tally)) {
			r.distance = new_distance * 8.f + sub_distance * 4.f + r.tminn;
			if (DBG) { info.level = 1; info.sub_id = sub; }
			r.hit = true;
			return ST_NEED;
		}
	} else if (index & kLoadedBit) {
		if (DBG) tally.brick_tests++;
		int sub = 0;
		const f3 o8 = (r.o + r.d * new_distance) * 8.f - r.n * kEpsilon;
		if (intersect_grid<8, DBG>(o8, r.d, r.sx, r.sy, r.sz, r.dx, r.dy, r.dz, r.n, sub_distance, brick, 0u, sub, tally, lds_brick)) {
			r.distance = new_distance * 8.f + sub_distance + r.tminn;
			if (DBG) { info.level = 2; info.sub_id = sub; }
			r.hit = true;
			return ST_NEED;
		}
	} else if (index & kUnloadedBit) {
		// brick-request protocol (voxel.cuh:228-245): 32-bit atomics on the index word and the ring counter
		const uint32_t old = atomicOr(&sc.index_grid[flat], kRequestedBit);
		if (!(old & kRequestedBit)) {
			const uint32_t load_index = atomicAdd(sc.load_queue_count, 1u);
			if (load_index < sc.queue_cap) {
				int* q = sc.load_queue + 3 * static_cast<size_t>(load_index);
				q[0] = px; q[1] = py; q[2] = pz;
				if (DBG) tally.requests++;
			} else {
				atomicAnd(&sc.index_grid[flat], ~kRequestedBit);
			}
		}
		r.distance = new_distance * 8.f + r.tminn;
		if (DBG) { info.level = 3; info.sub_id = 0; }
		r.hit = true;
		return ST_NEED;
	}
	return ST_OUTER; // nothing solid along the ray inside this brick: keep walking
}

// ---- sky model (sunsky.cu:10-161); view-independent terms arrive precomputed in FrameConstants.
// Split so that lanes shading a sun sample (sun()) and lanes shading a miss (sky() / sunsky()) share the
// extinction term.  RayleighPhase / hgPhase (sunsky.cu:10-12,20-22) contain double literals in the
// reference; they are evaluated in fp32 here (difference ~1e-6 relative, inside the 1e-4 radiance bar).
struct SkyView {
	f3 Fex;           // combined extinction factor
	float cosViewSun;
};
__device__ __forceinline__ SkyView sky_view(const FrameConstants& fc, f3 viewDir) {
	SkyView o;
	o.cosViewSun = dot(viewDir, ld3(fc.sun_direction));
	const float cosUpView = dot(mk(0.f, 0.f, 1.f), viewDir);
	const float zenith = gmax(0.0f, cosUpView);
	const float rayleighLen = 8.4E3f / zenith;
	const float mieLen = 1.25E3f / zenith;
	const f3 a = ld3(fc.rayleigh) * rayleighLen + ld3(fc.mie) * mieLen;
	o.Fex = mk(expf(-a.x), expf(-a.y), expf(-a.z));
	return o;
}
// in-scattered sky light: `sky` of sunsky.cu:109-111 (before the 0.01 / SkyFactor scaling)
__device__ __forceinline__ f3 sky_scatter(const FrameConstants& fc, const SkyView& v) {
	const float c = v.cosViewSun;
	const float rayleighPhase = (3.0f / (16.0f * kPi)) * (1.0f + c * c);
	const float g = 0.80f, g2 = 0.80f * 0.80f;
	const float base = 1.0f - 2.0f * g * c + g2;
	const float hg = (1.0f / (4.0f * kPi)) * ((1.0f - g2) / (base * sqrtf(base)));
	const f3 light = ld3(fc.rayleigh) * rayleighPhase + ld3(fc.mie) * hg;
	const f3 somethingElse = (light / ld3(fc.total)) * fc.sunE;
	const f3 sky = somethingElse * mk(1.0f - v.Fex.x, 1.0f - v.Fex.y, 1.0f - v.Fex.z);
	const f3 q = somethingElse * v.Fex;
	const f3 p = mk(sqrtf(q.x), sqrtf(q.y), sqrtf(q.z)); // pow(x, 0.5)
	const float a = fc.mixf;
	return sky * mk(1.0f * (1.0f - a) + p.x * a, 1.0f * (1.0f - a) + p.y * a, 1.0f * (1.0f - a) + p.z * a);
}
__device__ __forceinline__ f3 sun_from_view(const FrameConstants& fc, const SkyView& v) { // sun(), sunsky.cu:32-74
	// quirk kept: `sunAngularDiameterCos < (cosViewSunAngle ? 1.0 : 0.0)` tests cos != 0
	const float sundisk = static_cast<double>(fc.sun_angular_cos) < (v.cosViewSun != 0.0f ? 1.0 : 0.0) ? 1.0f : 0.0f;
	return ((v.Fex * (fc.sunE * 19000.0f)) * sundisk) * 0.01f;
}
__device__ __forceinline__ f3 sky_from_view(const FrameConstants& fc, const SkyView& v) { // sky(), sunsky.cu:76-114
	return sky_scatter(fc, v) * (1.f * 0.01f);
}
__device__ __forceinline__ f3 sunsky_from_view(const FrameConstants& fc, const SkyView& v) { // sunsky(), sunsky.cu:116-161
	const f3 sky = sky_scatter(fc, v);
	const float e0 = fc.sun_angular_cos, e1 = fc.sun_angular_cos + 0.00002f;
	const float s = gmin(gmax((v.cosViewSun - e0) / (e1 - e0), 0.0f), 1.0f);
	const float sundisk = s * s * (3.0f - 2.0f * s);
	const f3 sun = ((v.Fex * (fc.sunE * 19000.0f)) * sundisk) * 1E-5f;
	return (sun + sky) * 0.01f;
}
__device__ __forceinline__ f3 sun_radiance(const FrameConstants& fc, f3 viewDir) { return sun_from_view(fc, sky_view(fc, viewDir)); }
__device__ __forceinline__ f3 sky_radiance(const FrameConstants& fc, f3 viewDir) { return sky_from_view(fc, sky_view(fc, viewDir)); }
__device__ __forceinline__ f3 sunsky_radiance(const FrameConstants& fc, f3 viewDir) {
	if (fc.sun_angular_cos == 1.0f) return mk(1.0f, 0.0f, 0.0f); // sunsky.cu:121-123
	return sunsky_from_view(fc, sky_view(fc, viewDir));
}

// getConeSample(sunDirection, extent, seed) (sunsky.cu:163-183).  Its orthonormal frame depends only on the
// sun direction, so normalize(dir), o1 and o2 arrive precomputed (same fp32 operations, done once on the host).
__device__ __forceinline__ f3 cone_sample(const FrameConstants& fc, uint32_t& seed) {
	const f3 dir = ld3(fc.cone_dir), o1 = ld3(fc.cone_o1), o2 = ld3(fc.cone_o2);
	float rx = random_float2(seed);
	float ry = random_float2(seed);
	rx = rx * 2.f * kPi;
	ry = 1.0f - ry * fc.cone_extent;
	const float oneminus = sqrtf(1.0f - ry * ry);
	float s, c;
	det_sincos(rx, s, c);
	return (o1 * (c * oneminus) + o2 * (s * oneminus)) + dir * ry;
}

// hit-record hashing, identical to oracle.c (hmix / pack_normal)
__device__ __forceinline__ uint32_t hmix(uint32_t h, uint32_t v) {
	h ^= v;
	h *= 16777619u;
	h ^= h >> 15;
	return h;
}
__device__ __forceinline__ uint32_t pack_normal(f3 n) {
	const float c[3] = {n.x, n.y, n.z};
	uint32_t r = 0;
	for (int i = 0; i < 3; i++) {
		const uint32_t code = c[i] == 0.0f ? 0u : (c[i] == 1.0f ? 1u : (c[i] == -1.0f ? 2u : 3u));
		r |= code << (2 * i);
	}
	return r;
}

} // namespace

// Persistent-wave path tracer.
//
// Work distribution: the shard's pixels are cut into 4x4-pixel chunks (ordered so that four consecutive
// chunks form an 8x8 block and sixteen a 16x16 tile).  Waves are persistent: whenever 16 or more of a
// wave's lanes have no pixel, the wave takes that many chunks from a global counter (one atomic per
// refill) and hands one pixel to each idle lane.  A lane traces ALL samples of its pixel, in order, before
// it takes another one, so each pixel's accumulation order is fixed (sample by sample, event by event).
//
// Path state machine per lane:
//   GEN -> [extend ray] -> EXT_DONE (shade) -> [shadow ray] -> SHD_DONE (connect) -> BOUNCE -> [extend ray] ...
// A wave interleaves three kinds of work, each run only when enough lanes want it (or nothing else can
// run), so that the expensive, rarely-needed code never executes for a handful of lanes:
//   phase A  one brick-grid DDA move            lanes in ST_OUTER   (cheap, most of the work)
//   phase B  index word + 8^3 / 2^3 bitmask DDA  lanes in ST_CAND    (expensive, ~2.5 per ray)
//   phase C  shade / next primary ray + setup    lanes in ST_NEED    (expensive, once per extend ray)
//   phase D  connect + stored bounce ray setup   lanes in ST_CONN    (cheap, once per shadow ray)
// Scheduling changes only WHEN a lane's operations happen, never their operands, so results are
// identical to the reference's per-ray functions run one ray at a time (the oracle).
enum : int { P_GEN = 0, P_EXT_DONE = 1, P_SHD_DONE = 2, P_BOUNCE = 3 };
enum : int { ST_IDLE = 4, ST_CONN = 5 };

#ifndef BM_WAVES_PER_SIMD
#define BM_WAVES_PER_SIMD 4
#endif
#ifndef BM_WORK_COUNTERS
#define BM_WORK_COUNTERS 8
#endif
#ifndef BM_QUORUM_DIV
#define BM_QUORUM_DIV 4
#endif
#ifndef BM_QUORUM_CONN_DIV
#define BM_QUORUM_CONN_DIV 8
#endif
#ifndef BM_STEPS_PER_ROUND
#define BM_STEPS_PER_ROUND 8
#endif
// -DBM_PHASE_TIMING: profiling build in which the plain kernel also reports the scheduler statistics
#ifdef BM_PHASE_TIMING
#define BM_TIMED true
#else
#define BM_TIMED DBG
#endif
template <bool DBG>
// (the instrumented variant carries hit records and counters: it gets the registers instead of the occupancy)
__global__ __launch_bounds__(256, DBG ? 2 : BM_WAVES_PER_SIMD) void trace_paths(const DeviceScene sc, const FrameConstants* __restrict__ fcp, float4* __restrict__ accum,
												  uint32_t* __restrict__ dbg, DeviceCounters* __restrict__ counters,
												  uint32_t* __restrict__ work_counter) {
	// the per-frame constants live in device memory (not in the kernel-argument registers): they are read with scalar
	// loads where they are needed, which keeps the scalar register file free for the scheduler loop
	const FrameConstants& fc = *fcp;
	__shared__ unsigned long long lds_brick[8 * 256]; // 16 KiB: one 64-byte brick per thread (brick_to_lds)
	const int lane = threadIdx.x & 63;
	const uint32_t W = static_cast<uint32_t>(fc.width), H = static_cast<uint32_t>(fc.height);
	const uint32_t total_chunks = static_cast<uint32_t>(fc.tiles_x) * static_cast<uint32_t>(fc.tiles_y) * 16u;

	// per-pixel state
	uint32_t xy = 0;          // x | y << 16
	uint32_t p = 0;           // global pixel index y*W + x
	uint32_t local_pixel = 0; // index into this shard's packed buffers
	Tally tally;
	HitInfo info;
	uint32_t d0 = 0, d1 = 0, d2 = 0xFFFFFFFFu, d3 = 0, hseg = 2166136261u, hsh = 2166136261u, next = 0, nsh = 0, loads0 = 0;

	RayState r;
	r.hit = false;
	r.n = mk(0.f, 0.f, 0.f);
	int state = ST_IDLE;
	int pstate = P_GEN;
	int s = 0;               // sample being traced
	int bounces = 0;
	bool shadow = false;     // kind of the ray in flight
	bool terminated = false; // path ends after its pending shadow ray
	f3 hitp = mk(0.f, 0.f, 0.f);   // surface point: shadow-ray origin and next extend origin
	f3 pn = mk(0.f, 0.f, 0.f);     // surface normal of the path (RayQueue::normal)
	f3 scolor = mk(0.f, 0.f, 0.f); // ShadowQueue::color
	f3 bdir = mk(0.f, 0.f, 0.f);   // next bounce direction, drawn in shade, used once the shadow ray is done

	bool work_left = true;
	constexpr uint32_t kCounters = BM_WORK_COUNTERS, kCounterStride = 32; // one 128-byte line per counter
	int my_counter = static_cast<int>((blockIdx.x * 4u + (threadIdx.x >> 6)) % kCounters);
	int counters_done = 0;
	// hang guard only (NaN directions): no wave needs more scheduler rounds than this
	long long rounds_left = (static_cast<long long>(total_chunks) + 64) * (static_cast<long long>(fc.spp) + 1) * (fc.max_bounces + 2) *
							(2ll * sc.cells + sc.cells_height + 64);
	uint32_t runsA = 0, lanesA = 0, runsB = 0, lanesB = 0, runsC = 0, lanesC = 0, runsD = 0, lanesD = 0; // wave-uniform scheduler statistics

	unsigned long long cycA = 0, cycB = 0, cycC = 0, cycD = 0;
	const unsigned long long t_begin = BM_TIMED ? __builtin_amdgcn_s_memtime() : 0ull;

	for (;;) {
		// ---- refill: hand pixels to idle lanes, 16 (one 4x4 chunk) at a time
		const unsigned long long idle = __ballot(state == ST_IDLE);
		const int nI = __popcll(idle);
		if (work_left && nI >= 16) {
			// One global word serves only ~90 returning atomics per microsecond chip-wide, and a refill stalls the whole
			// wave until its atomic returns; with thousands of waves on one counter that queue is tens of microseconds
			// long.  The chunk sequence is therefore dealt to kCounters interleaved counters (8x8-pixel groups of four
			// chunks, group g on counter g % kCounters, each counter on its own cache line): every counter still sweeps
			// the image top-down, so concurrently running waves keep working on neighbouring rows of the image.
			const int want = nI >> 4;
			uint32_t base = 0;
			if (lane == 0) base = atomicAdd(work_counter + my_counter * kCounterStride, static_cast<uint32_t>(want));
			base = __builtin_amdgcn_readfirstlane(base);
			const uint32_t total_groups = (total_chunks + 3u) >> 2;
			const uint32_t my_groups = total_groups > static_cast<uint32_t>(my_counter)
										   ? (total_groups - static_cast<uint32_t>(my_counter) + kCounters - 1u) / kCounters : 0u;
			const uint32_t my_tickets = my_groups * 4u;
			const uint32_t counter_now = static_cast<uint32_t>(my_counter);
			if (base + want >= my_tickets) { // this counter is used up: move to the next one (helping out), or finish
				my_counter = (my_counter + 1) % static_cast<int>(kCounters);
				if (++counters_done >= static_cast<int>(kCounters)) work_left = false;
			}
			const int rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(idle >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(idle), 0u));
			if (state == ST_IDLE && rank < want * 16) {
				const uint32_t ticket = base + static_cast<uint32_t>(rank >> 4);
				const uint32_t chunk = ((ticket >> 2) * kCounters + counter_now) * 4u + (ticket & 3u);
				if (ticket < my_tickets && chunk < total_chunks) {
					const uint32_t tile = chunk >> 4, k = chunk & 15u;
					const int tile_x = static_cast<int>(tile % static_cast<uint32_t>(fc.tiles_x));
					const int tile_y = static_cast<int>(tile / static_cast<uint32_t>(fc.tiles_x));
					const int cx = static_cast<int>((k & 1u) | ((k >> 1) & 2u)), cy = static_cast<int>(((k >> 1) & 1u) | ((k >> 2) & 2u));
					const int x = tile_x * 16 + cx * 4 + (rank & 3);
					const int ly = tile_y * 16 + cy * 4 + ((rank >> 2) & 3); // row inside this shard's packed buffer
					const int y = ((ly / fc.band_rows) * fc.shard_count + fc.shard_rank) * fc.band_rows + ly % fc.band_rows;
					if (x < fc.width && ly < fc.local_rows && y < fc.height) {
						p = static_cast<uint32_t>(y) * W + static_cast<uint32_t>(x);
						local_pixel = static_cast<uint32_t>(ly) * W + static_cast<uint32_t>(x);
						xy = static_cast<uint32_t>(x) | (static_cast<uint32_t>(y) << 16);
						s = 0;
						pstate = P_GEN;
						state = ST_NEED;
						if (DBG) {
							d0 = 0; d1 = 0; d2 = 0xFFFFFFFFu; d3 = 0; hseg = 2166136261u; hsh = 2166136261u; next = 0; nsh = 0;
							loads0 = tally.index_loads;
						}
					}
				}
			}
		}
		const int nA = __popcll(__ballot(state == ST_OUTER));
		const int nB = __popcll(__ballot(state == ST_CAND));
		const int nC = __popcll(__ballot(state == ST_NEED));
		const int nD = __popcll(__ballot(state == ST_CONN));
		const int live = nA + nB + nC + nD;
		if (live == 0) {
			if (!work_left || --rounds_left < 0) break;
			continue; // everything idle but chunks remain (only pixels outside the image were handed out)
		}
		if (--rounds_left < 0) break;
		// Policy: an expensive phase runs once a quarter of the live lanes wait for it, the cheap connect phase
		// once an eighth does; otherwise the DDA keeps moving.  With no lane left in the DDA the largest group runs.
		const int quorum = (live + BM_QUORUM_DIV - 1) / BM_QUORUM_DIV, quorum_conn = (live + BM_QUORUM_CONN_DIV - 1) / BM_QUORUM_CONN_DIV;
		int phase; // 0 = A (DDA moves), 1 = B (candidates), 2 = C (shade / generate), 3 = D (connect)
		if (nC >= quorum) phase = 2;
		else if (nB >= quorum) phase = 1;
		else if (nD >= quorum_conn) phase = 3;
		else if (nA > 0) phase = 0;
		else phase = (nC >= nB && nC >= nD) ? 2 : (nB >= nD ? 1 : 3);

		const unsigned long long t_phase = BM_TIMED ? __builtin_amdgcn_s_memtime() : 0ull;
		if (phase == 2) {
			if (BM_TIMED) { runsC++; lanesC += nC; }
			// ================= phase C: shade the finished extend ray / generate the next primary ray, then set the new ray up
			if (state == ST_NEED) {
				bool need_setup = false;
				f3 ro = mk(0.f, 0.f, 0.f), rd = mk(0.f, 0.f, 0.f);
				if (pstate == P_EXT_DONE) {
					// ---- extend finished (kernel.cu:226-238); `hit <=> distance < VERY_FAR`
					const bool is_hit = r.hit;
					pn = r.n; // extend writes RayQueue::normal in place (also clobbered on the way to a miss; unused then)
					if (DBG) {
						tally.extend_rays++;
						next++;
						if (s == 0 && bounces == 0) {
							d0 = is_hit ? __float_as_uint(r.distance) : 0u;
							d1 = is_hit ? (pack_normal(pn) | (1u << 8) | (static_cast<uint32_t>(info.level) << 12)) : 0u;
							d2 = is_hit ? static_cast<uint32_t>(info.brick_id) : 0xFFFFFFFFu;
							d3 = is_hit ? static_cast<uint32_t>(info.sub_id) : 0u;
						}
						hseg = hmix(hseg, static_cast<uint32_t>(is_hit));
						if (is_hit) {
							hseg = hmix(hseg, __float_as_uint(r.distance));
							hseg = hmix(hseg, pack_normal(pn) | (static_cast<uint32_t>(info.level) << 12));
							hseg = hmix(hseg, static_cast<uint32_t>(info.brick_id));
							hseg = hmix(hseg, static_cast<uint32_t>(info.sub_id));
						}
					}
					const bool primary_only = fc.flags & 1u; // BM_FLAG_PRIMARY_ONLY
					// direction whose sky terms are needed: the ray itself on a miss, the sun sample on a hit
					f3 view = r.d; // RayQueue::direction of the extend ray that just finished
					f3 miss_color = mk(0.f, 0.f, 0.f);
					float sunLight = 0.f;
					bool cast = false;
					if (is_hit && !primary_only) {
						// ---- shade, hit branch (kernel.cu:255-302); frame = base_frame + bounce, queue slot = slot.
						// throughput is identically (1,1,1) (kernel.cu:261,271) and is not carried.
						const uint32_t frame = fc.base_frame + static_cast<uint32_t>(bounces);
						const uint32_t slot = p + static_cast<uint32_t>(fc.sample_base + s) * W * H;
						uint32_t sseed = (frame * p * 147565741u) * 720898027u * slot;
						hitp = hitp + r.d * r.distance;
						hitp = hitp + pn * 2.f * kEpsilon;
						view = cone_sample(fc, sseed);
						sunLight = dot(pn, view);
						cast = sunLight > 0.f;
						terminated = !(bounces < fc.max_bounces);
						if (terminated) {
							accum[local_pixel].w += 1.f; // kernel.cu:301
						} else {
							// kernel.cu:281-299: cosine-weighted bounce, drawn right after the cone sample as in shade();
							// the direction is kept in `bdir` until the shadow ray (if any) has been traced.
							const float r1 = 2.f * kPi * random_float(sseed);
							const float r2 = random_float(sseed);
							const float r2s = sqrtf(r2);
							// computeOrthonormalBasisNaive (kernel.cu:76-84)
							f3 u = fabs(static_cast<double>(pn.x)) > .9 ? mk(0.0f, 1.0f, 0.0f) : mk(1.0f, 0.0f, 0.0f);
							u = normalize(cross(u, pn));
							const f3 v = cross(pn, u);
							float sn, cs;
							det_sincos(r1, sn, cs);
							bdir = normalize(((u * cs) * r2s + (v * sn) * r2s) + pn * sqrtf(1 - r2));
						}
						if (!cast) {
							if (terminated) { s++; pstate = P_GEN; }
							else { bounces++; ro = hitp; rd = bdir; r.n = pn; shadow = false; need_setup = true; }
						}
					}
					if (!is_hit || cast) {
						const SkyView sv = sky_view(fc, view);
						if (cast) {
							scolor = (sun_from_view(fc, sv) * sunLight) * 1E-5f; // kernel.cu:278
							ro = hitp; rd = view;
							shadow = true;
							need_setup = true;
						} else {
							// ---- shade, miss branch (kernel.cu:316-323)
							f3 c;
							if (bounces == 0) c = fc.sun_angular_cos == 1.0f ? mk(1.0f, 0.0f, 0.0f) : sunsky_from_view(fc, sv);
							else c = sky_from_view(fc, sv);
							miss_color = c;
						}
					}
					if (!is_hit || primary_only) { // the path ends here: one read-modify-write of the pixel's accumulator
						float4 a = accum[local_pixel];
						a.x += miss_color.x; a.y += miss_color.y; a.z += miss_color.z; // (0,0,0) for a primary-only hit
						a.w += 1.f;
						accum[local_pixel] = a;
						s++;
						pstate = P_GEN;
					}
				}
				if (pstate == P_GEN) {
					if (s >= fc.spp) {
						// pixel finished (its accumulator lives in memory and is already up to date): wait for the next one
						if (DBG && dbg) {
							uint32_t* d = dbg + static_cast<size_t>(local_pixel) * 8;
							d[0] = d0; d[1] = d1; d[2] = d2; d[3] = d3; d[4] = hseg; d[5] = hsh; d[6] = next | (nsh << 16);
							d[7] = tally.index_loads - loads0;
						}
						state = ST_IDLE;
					} else {
						// ---- primary_rays (kernel.cu:157-200) for queue slot `slot`, start_position 0
						const uint32_t slot = p + static_cast<uint32_t>(fc.sample_base + s) * W * H;
						uint32_t seed = (fc.base_frame * 147565741u) * 720898027u * slot;
						const f3 cam_right = ld3(fc.right), cam_up = ld3(fc.up), cam_dir = ld3(fc.dir), cam_o = ld3(fc.origin);
						// Random2DStratifiedSample (kernel.cu:40-61)
						const int stratum = static_cast<int>(random_float(seed) * (16 + 0.99999f));
						const int stratumX = stratum % 4, stratumY = (stratum / 4) % 4;
						const float jx = 0.25f * stratumX + (random_float(seed) * 0.25f);
						const float jy = 0.25f * stratumY + (random_float(seed) * 0.25f);
						const float ppx = static_cast<float>(xy & 0xFFFFu) - jx;
						const float ppy = static_cast<float>(xy >> 16) - jy;
						const float ni = (ppx / static_cast<float>(W)) - 0.5f;
						const float nj = ((static_cast<float>(H) - ppy) / static_cast<float>(H)) - 0.5f;
						const f3 to_focal = normalize(cam_dir + cam_right * ni + cam_up * nj);
						const f3 convergence = cam_o + to_focal * fc.focal3;
						hitp = cam_o;
						if (fc.lens_radius != 0.f) {
							// thin-lens sample (kernel.cu:194-196).  With lens radius 0 the reference multiplies the disk
							// sample by 0, and nothing downstream reads this seed again, so the draws can be skipped.
							const float l0 = random_float(seed); // canonical order: left to right
							const float l1 = random_float(seed);
							float lx = 0.f, lyy = 0.f;
							const float ox = 2.f * l0 - 1.f, oy = 2.f * l1 - 1.f; // ConcentricSampleDisk (kernel.cu:85-103)
							if (!(ox == 0 && oy == 0)) {
								float theta, rr;
								if (fabsf(ox) > fabsf(oy)) { rr = ox; theta = kPi / 4 * (oy / ox); }
								else { rr = oy; theta = kPi / 2 - kPi / 4 * (ox / oy); }
								float sn, cs;
								det_sincos(theta, sn, cs);
								lx = rr * cs;
								lyy = rr * sn;
							}
							const float plx = fc.lens_radius * lx, ply = fc.lens_radius * lyy;
							hitp = cam_o + cam_right * plx + cam_up * ply;
						}
						rd = normalize(convergence - hitp);
						pn = mk(0.f, 0.f, 0.f);
						bounces = 0;
						terminated = false;
						if (DBG) tally.paths++;
						ro = hitp;
						r.n = pn;
						shadow = false;
						need_setup = true;
					}
				}
				if (need_setup) {
					if (shadow) r.n = mk(0.f, 0.f, 0.f); // connect passes a zeroed normal (kernel.cu:338)
					pstate = shadow ? P_SHD_DONE : P_EXT_DONE;
					const int st = ray_setup<DBG>(sc, ro, rd, r, tally);
					state = (st == ST_NEED && shadow) ? ST_CONN : st;
				}
			}
		} else if (phase == 3) {
			if (BM_TIMED) { runsD++; lanesD += nD; }
			// ================= phase D: connect (kernel.cu:328-346) -- runs after shade within the same reference frame --
			// then the stored bounce ray is set up
			if (state == ST_CONN) {
				const bool occluded = r.hit;
				if (DBG) {
					tally.shadow_rays++;
					nsh++;
					hsh = hmix(hsh, static_cast<uint32_t>(occluded));
					if (occluded) {
						hsh = hmix(hsh, static_cast<uint32_t>(info.brick_id));
						hsh = hmix(hsh, static_cast<uint32_t>(info.sub_id) | (static_cast<uint32_t>(info.level) << 12));
					}
				}
				if (!occluded) {
					float4 a = accum[local_pixel];
					a.x += scolor.x; a.y += scolor.y; a.z += scolor.z;
					accum[local_pixel] = a;
				}
				if (terminated) {
					s++;
					pstate = P_GEN;
					state = ST_NEED; // the next primary ray (or the pixel hand-back) is phase C work
				} else {
					bounces++;
					r.n = pn;
					shadow = false;
					pstate = P_EXT_DONE;
					state = ray_setup<DBG>(sc, hitp, bdir, r, tally);
				}
			}
		} else if (phase == 1) {
			if (BM_TIMED) { runsB++; lanesB += nB; }
			// ================= phase B: resolve non-empty cells (index word, LoD / 8^3 bitmask DDA, streaming request)
			if (state == ST_CAND) {
				const int st = process_candidate<DBG>(sc, fc.campos, r, info, tally, lds_brick);
				state = (st == ST_NEED && shadow) ? ST_CONN : st;
			}
		} else {
			// ================= phase A: brick-grid DDA moves; lanes that reach a non-empty cell or leave the grid wait
#pragma unroll 1
			for (int k = 0; k < BM_STEPS_PER_ROUND; ++k) {
				if (BM_TIMED) { runsA++; lanesA += __popcll(__ballot(state == ST_OUTER)); }
				if (state == ST_OUTER) {
					const int st = outer_step<DBG>(sc, r, tally);
					state = (st == ST_NEED && shadow) ? ST_CONN : st;
				}
			}
		}
		if (BM_TIMED) {
			const unsigned long long dt = __builtin_amdgcn_s_memtime() - t_phase;
			if (phase == 0) cycA += dt; else if (phase == 1) cycB += dt; else if (phase == 2) cycC += dt; else cycD += dt;
		}
	}

	if (DBG && counters) { // wave-level sum, one atomic per wave and counter
		unsigned long long v[8] = {tally.index_loads, tally.brick_tests, tally.byte_tests, tally.voxel_steps,
								   tally.extend_rays, tally.shadow_rays, tally.requests, tally.paths};
		for (int k = 0; k < 8; ++k) {
			unsigned long long t = v[k];
			for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
			if (lane == 0 && t) atomicAdd(&counters->v[k], t);
		}
	}
	if (BM_TIMED && counters && lane == 0) {
		const unsigned long long st[8] = {runsA, lanesA, runsB, lanesB, runsC, lanesC, runsD, lanesD};
		for (int k = 0; k < 8; ++k) atomicAdd(&counters->sched[k], st[k]);
		const unsigned long long cy[8] = {cycA, cycB, cycC, cycD, __builtin_amdgcn_s_memtime() - t_begin, 0ull, 0ull, 1ull};
		for (int k = 0; k < 8; ++k) atomicAdd(&counters->cycles[k], cy[k]);
	}
}

// upload kernel (kernel.cu:141-151): scatter staged bricks into the arena and publish their index words
__global__ void upload_bricks(const DeviceScene sc, const uint32_t* __restrict__ bricks_queue, const uint32_t* __restrict__ indices_queue,
							  uint32_t* __restrict__ arena_rw, uint32_t count) {
	const uint32_t i = blockIdx.x * (blockDim.x / 16) + threadIdx.x / 16; // 16 lanes move one 64-byte brick
	const uint32_t w = threadIdx.x & 15;
	if (i >= count) return;
	const int px = sc.load_queue[3 * i + 0], py = sc.load_queue[3 * i + 1], pz = sc.load_queue[3 * i + 2];
	const int sci = (px / 16) + (py / 16) * sc.sg_xy + (pz / 16) * sc.sg_xy2;
	const uint32_t local = static_cast<uint32_t>((px % 16) + (py % 16) * 16 + (pz % 16) * 256);
	const uint32_t word = indices_queue[i];
	const uint32_t slot = sc.super_info[sci].brick_base + (word & kIndexBits);
	arena_rw[(static_cast<size_t>(slot) << 4) + w] = bricks_queue[(static_cast<size_t>(i) << 4) + w];
	if (w == 0) sc.index_grid[(static_cast<size_t>(sci) << 12) + local] = word; // plain store: clears unloaded + requested
}

// blit_onto_framebuffer (kernel.cu:348-364) into an offscreen float4 buffer
__global__ void resolve_kernel(const float4* __restrict__ accum, float4* __restrict__ out, long long n) {
	const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const float4 c = accum[i];
	float4 o;
	o.x = powf(c.x / c.w, 1.f / 2.2f);
	o.y = powf(c.y / c.w, 1.f / 2.2f);
	o.z = powf(c.z / c.w, 1.f / 2.2f);
	o.w = powf(1.f, 1.f / 2.2f);
	out[i] = o;
}

__global__ void debug_sincos_kernel(int n, const float* __restrict__ x, float* __restrict__ s, float* __restrict__ c) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) det_sincos(x[i], s[i], c[i]);
}

__global__ void debug_sky_kernel(const FrameConstants fc, int n, const float* __restrict__ v, float* __restrict__ sun,
								 float* __restrict__ sky, float* __restrict__ sunsky) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const f3 d = mk(v[3 * i], v[3 * i + 1], v[3 * i + 2]);
	const f3 a = sun_radiance(fc, d), b = sky_radiance(fc, d), c = sunsky_radiance(fc, d);
	sun[3 * i] = a.x; sun[3 * i + 1] = a.y; sun[3 * i + 2] = a.z;
	sky[3 * i] = b.x; sky[3 * i + 1] = b.y; sky[3 * i + 2] = b.z;
	sunsky[3 * i] = c.x; sunsky[3 * i + 1] = c.y; sunsky[3 * i + 2] = c.z;
}

// ---- host-callable launchers (kernels.h)
int trace_blocks_per_cu(bool instrumented) {
	int n = 0;
	const hipError_t e = instrumented ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trace_paths<true>, 256, 0)
									  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trace_paths<false>, 256, 0);
	return e == hipSuccess && n > 0 ? n : 1;
}

// Persistent launch: exactly as many 256-thread workgroups as the device keeps resident (compute_units x
// blocks per CU); the waves pull 4x4-pixel chunks from *work_counter, which must be zero at launch.
void launch_trace(const DeviceScene& sc, const FrameConstants& fc, const FrameConstants* fc_dev, float* accum, uint32_t* dbg, DeviceCounters* counters,
				  uint32_t* work_counter, bool instrumented, int resident_blocks, hipStream_t stream) {
	const long long chunks = static_cast<long long>(fc.tiles_x) * fc.tiles_y * 16;
	if (chunks <= 0) return;
	long long blocks = (chunks + 15) / 16; // never more workgroups than 64-pixel groups
	if (blocks > resident_blocks) blocks = resident_blocks;
	if (instrumented)
		hipLaunchKernelGGL(trace_paths<true>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, sc, fc_dev, reinterpret_cast<float4*>(accum), dbg,
						   counters, work_counter);
	else
#ifdef BM_PHASE_TIMING
		hipLaunchKernelGGL(trace_paths<false>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, sc, fc_dev, reinterpret_cast<float4*>(accum), nullptr,
						   counters, work_counter);
#else
		hipLaunchKernelGGL(trace_paths<false>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, sc, fc_dev, reinterpret_cast<float4*>(accum), nullptr,
						   nullptr, work_counter);
#endif
}

void launch_upload(const DeviceScene& sc, const uint32_t* bricks_queue, const uint32_t* indices_queue, uint32_t* arena, uint32_t count,
				   hipStream_t stream) {
	if (count == 0) return;
	const int per_block = 256 / 16;
	hipLaunchKernelGGL(upload_bricks, dim3((count + per_block - 1) / per_block), dim3(256), 0, stream, sc, bricks_queue, indices_queue, arena, count);
}

void launch_resolve(const float* accum, float* out, long long n, hipStream_t stream) {
	if (n <= 0) return;
	hipLaunchKernelGGL(resolve_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream, reinterpret_cast<const float4*>(accum),
					   reinterpret_cast<float4*>(out), n);
}

void launch_debug_sincos(int n, const float* x, float* s, float* c, hipStream_t stream) {
	hipLaunchKernelGGL(debug_sincos_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, x, s, c);
}

void launch_debug_sky(const FrameConstants& fc, int n, const float* v, float* sun, float* sky, float* sunsky, hipStream_t stream) {
	hipLaunchKernelGGL(debug_sky_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, fc, n, v, sun, sky, sunsky);
}

} // namespace bm
