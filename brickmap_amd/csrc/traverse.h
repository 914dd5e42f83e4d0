// traverse.h -- device-side building blocks shared by the path-trace kernels (trace.hip: fused per-pixel paths;
// wavefront.hip: the reference's queue-based extend / shade / connect schedule): GLM-order vector helpers, the
// xorshift RNG, the brick-grid / 8^3 / 2^3 DDA of src/voxel.cuh, the sky model of src/sunsky.cu.
// Per-ray arithmetic follows the reference operation for operation (IEEE fp32, no contraction; DESIGN.md "Numeric
// contract"), so every kernel built from these pieces produces hits bit-identical to the CPU oracle.
#pragma once
#include <hip/hip_runtime.h>

#include "detmath.h"
#include "device_types.h"
#include "jump.h"

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

// LDS-direct staging (the default of the fused and the queue kernels): the brick never passes through registers.  gfx950's
// global_load_lds_dword moves 4 bytes per lane from the lane's own global address to LDS at M0 + offset + lane * 4 -- lane-
// contiguous words, which is exactly a bank-conflict-free layout: word k (0...15) of thread t's brick at word k * 256 + t of
// the staging area.  Sixteen such loads (M0 steps by 1020 bytes between them: the instruction offset, k * 4, counts for the
// global AND the LDS address) replace four 16-byte loads + a wait + eight ds_write_b64; a z-slice is the two words 2z, 2z + 1,
// one ds_read2st64_b32.  The staging sits on the candidate pass's critical path -- staging the brick a second time cost 2.9 % of
// the 1080p frame (profiles/r05_brick_staging.txt) -- and inactive lanes write nothing, as with any masked store.
#ifndef BM_LDS_DMA
#define BM_LDS_DMA 1
#endif
typedef __attribute__((address_space(1))) const void* bm_global_cptr;
typedef __attribute__((address_space(3))) void* bm_lds_ptr;
__device__ __forceinline__ void brick_dma_to_lds(unsigned long long* lds_brick, const uint32_t* brick_words) {
	const uint32_t wave_off = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>((threadIdx.x & ~63u) * 4u))); // this wave's 64 columns
	char* const base = reinterpret_cast<char*>(lds_brick) + wave_off;
#define BM_DMA_WORD(k) __builtin_amdgcn_global_load_lds((bm_global_cptr)brick_words, (bm_lds_ptr)(base + (k) * 1020), 4, (k) * 4, 0)
	BM_DMA_WORD(0); BM_DMA_WORD(1); BM_DMA_WORD(2); BM_DMA_WORD(3); BM_DMA_WORD(4); BM_DMA_WORD(5); BM_DMA_WORD(6); BM_DMA_WORD(7);
	BM_DMA_WORD(8); BM_DMA_WORD(9); BM_DMA_WORD(10); BM_DMA_WORD(11); BM_DMA_WORD(12); BM_DMA_WORD(13); BM_DMA_WORD(14); BM_DMA_WORD(15);
#undef BM_DMA_WORD
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the loads write LDS: their data is there once they have returned
}
// measured alternative (-DBM_LDS_DMA=2): four 16-byte DMAs; chunk c (slices 2c, 2c + 1) of thread t at byte c * 4096 + t * 16 -- 12 fewer
// memory instructions per candidate pass, two more vector instructions per voxel step (the slice address has two fields): within
// noise of the default on every workload (profiles/r05_dma_variants.txt).  -DBM_LDS_DMA=0: bricks through registers (round 4).
__device__ __forceinline__ void brick_dma4_to_lds(unsigned long long* lds_brick, const uint32_t* brick_words) {
	const uint32_t wave_off = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>((threadIdx.x & ~63u) * 16u)));
	char* const base = reinterpret_cast<char*>(lds_brick) + wave_off;
#define BM_DMA_CHUNK(c) __builtin_amdgcn_global_load_lds((bm_global_cptr)brick_words, (bm_lds_ptr)(base + (c) * 4080), 16, (c) * 16, 0)
	BM_DMA_CHUNK(0); BM_DMA_CHUNK(1); BM_DMA_CHUNK(2); BM_DMA_CHUNK(3);
#undef BM_DMA_CHUNK
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Word j (0...15) of the 64-byte staging slot that belongs to lane k of the calling wave, as a float reference -- whatever the staging
// layout, a wave may scribble over the slots of ITS OWN lanes when none of them is inside a brick walk (trace.hip uses them to hand
// shadow rays from lane to lane in a shade pass); the slots of other waves may hold bricks that are being walked.
__device__ __forceinline__ float& staging_word(unsigned long long* lds_brick, uint32_t k, uint32_t j) {
	float* const f = reinterpret_cast<float*>(lds_brick);
	const uint32_t t = (threadIdx.x & ~63u) + k;
	if (BM_LDS_DMA == 2) return f[(j >> 2) * 1024u + t * 4u + (j & 3u)]; // chunk j / 4 of thread t
	if (BM_LDS_DMA) return f[j * 256u + t];                               // word j of thread t
	return f[((j >> 1) * 256u + t) * 2u + (j & 1u)];                      // half j & 1 of slice j / 2 of thread t
}

// z-slice (one 64-bit word: bit x + 8y) of the calling thread's staged brick
__device__ __forceinline__ unsigned long long brick_slice(const unsigned long long* lds_brick, uint32_t z) {
	if (BM_LDS_DMA == 2) return lds_brick[(z >> 1) * 512u + threadIdx.x * 2u + (z & 1u)];
	if (BM_LDS_DMA) {
		const uint32_t* w = reinterpret_cast<const uint32_t*>(lds_brick) + z * 512u + threadIdx.x;
		return static_cast<unsigned long long>(w[0]) | (static_cast<unsigned long long>(w[256]) << 32);
	}
	return lds_brick[z * 256u + threadIdx.x];
}
template <int N, bool DBG>
__device__ __forceinline__ bool intersect_grid(f3 origin, f3 dir, int sx, int sy, int sz, float dx, float dy, float dz, f3& normal, float& distance,
											   const BrickRegs& brick, uint32_t byte, int& sub_id, Tally& tally, unsigned long long* lds_brick = nullptr,
											   uint32_t* trips = nullptr, const uint32_t* brick_words = nullptr) {
	int px = static_cast<int>(origin.x), py = static_cast<int>(origin.y), pz = static_cast<int>(origin.z);
	const float cbx = dir.x > 0.f ? static_cast<float>(px + 1) : static_cast<float>(px);
	const float cby = dir.y > 0.f ? static_cast<float>(py + 1) : static_cast<float>(py);
	const float cbz = dir.z > 0.f ? static_cast<float>(pz + 1) : static_cast<float>(pz);
	// rdinv = sign * |1/d| exactly (0 for d == 0)
	const float rx = static_cast<float>(sx) * dx, ry = static_cast<float>(sy) * dy, rz = static_cast<float>(sz) * dz;
	float tx = dir.x != 0.f ? (cbx - origin.x) * rx : 1000000.f;
	float ty = dir.y != 0.f ? (cby - origin.y) * ry : 1000000.f;
	float tz = dir.z != 0.f ? (cbz - origin.z) * rz : 1000000.f;
	distance = 0.f;
	// The walk cell is ONE register: three 5-bit fields, each holding (coordinate % N) + 8, so a coordinate that
	// leaves [0, N) shows up as a change of its field's upper bits -- "still inside" is one AND + one compare for all
	// three axes, a move is one add of a per-axis constant, and the loop below has a single exit condition (a divergent
	// loop with several breaks spends more scalar instructions on exit-mask bookkeeping than vector ones on the walk).
	// (`& (N - 1)` only defines what the reference leaves undefined, a negative start cell; no effect otherwise.)
	constexpr uint32_t kOnes = 1u | (1u << 5) | (1u << 10);
	constexpr uint32_t kGuard = (~static_cast<uint32_t>(N - 1) & 0x1Fu) * kOnes, kInside = 8u * kOnes;
	uint32_t cell = ((static_cast<uint32_t>(px % N) & (N - 1)) | ((static_cast<uint32_t>(py % N) & (N - 1)) << 5) | ((static_cast<uint32_t>(pz % N) & (N - 1)) << 10)) + kInside;
	const int step_x = sx, step_y = sy * 32, step_z = sz * 1024;
	auto test = [&](uint32_t c, unsigned long long slice) -> bool {
		if (N == 8) return static_cast<uint32_t>(slice >> ((c & 7u) | ((c >> 2) & 0x38u))) & 1u;          // bit x + 8y of the z-slice
		return (byte >> ((c & 1u) | ((c >> 4) & 2u) | ((c >> 8) & 4u))) & 1u;                                  // bit x + 2y + 4z of the LoD mask
	};
	if (N == 8) {
		if (BM_LDS_DMA == 2) {
			brick_dma4_to_lds(lds_brick, brick_words);
		} else if (BM_LDS_DMA) {
			brick_dma_to_lds(lds_brick, brick_words);
		} else {
			brick_to_lds(lds_brick, brick);
		}
	}
	unsigned long long slice = N == 8 ? brick_slice(lds_brick, (cell >> 10) & 7u) : 0ull;
	if (DBG) tally.voxel_steps++;
	bool solid = test(cell, slice);
	bool inside = true;
	int last = 0; // packed increment of the last move: which axis it was
	// at most 3N-2 cells lie on a line through an N^3 block; the bound only guards against NaN input
	int guard = 3 * N + 1;
	for (; !solid && inside && guard > 0; --guard) {
		// select-style move (voxel.cuh:122-130); `t += mask ? delta : 0` is `tmax += mask * tdelta` for finite deltas
		const bool mx = tx < ty && tx < tz;
		const bool my = ty <= tx && ty < tz; // mx implies !my
		const bool mz = !(mx || my);
		last = mx ? step_x : (my ? step_y : step_z);
		cell += static_cast<uint32_t>(last);
		inside = (cell & kGuard) == kInside; // false: left the block (the values below are then unused)
		tx += mx ? dx : 0.f;
		ty += my ? dy : 0.f;
		tz += mz ? dz : 0.f;
		if (N == 8) slice = brick_slice(lds_brick, (cell >> 10) & 7u);
		if (DBG && inside) tally.voxel_steps++;
		solid = static_cast<bool>(static_cast<int>(inside) & static_cast<int>(test(cell, slice))); // no branch: (cell's fields are masked, any value is safe to test)
	}
	if (trips) *trips = static_cast<uint32_t>(3 * N + 2 - guard); // profiling builds: cells tested by this lane
	if (!solid) return false;
	// voxel.cuh:114-118, by select; a hit in the very first cell (no move) keeps distance 0 and the caller's normal
	const int a = last < 0 ? -last : last; // 0 = no move, 1 = x, 32 = y, 1024 = z
	normal = mk(a == 0 ? normal.x : (a == 1 ? -static_cast<float>(sx) : 0.f), a == 0 ? normal.y : (a == 32 ? -static_cast<float>(sy) : 0.f),
				a == 0 ? normal.z : (a == 1024 ? -static_cast<float>(sz) : 0.f));
	distance = a == 0 ? 0.f : (a == 1 ? tx - dx : (a == 32 ? ty - dy : tz - dz));
	sub_id = static_cast<int>((cell & (N - 1)) + ((cell >> 5) & (N - 1)) * N + ((cell >> 10) & (N - 1)) * N * N);
	return true;
}

// ---- brick-grid DDA (voxel.cuh:135-261), split into the three pieces the wave scheduler interleaves
//
// The reference reads one 32-bit index word per visited cell (two dependent loads through its pointer table); ~96 % of
// those words are zero (air).  The walk here reads one byte of the octant cube field per cell it stops in ("cube-field
// walk" below) and touches the index grid only at cells known to hold a brick; a border byte of that field is the
// reference's per-step exit test (voxel.cuh:256).  The tmax values it lands on are the reference's, bit for bit.
// Everything in a move is straight-line, select-style code: in a 64-lane wave every branch of a hot loop is taken by
// some lane on almost every iteration, so a rarely-needed path costs its full instruction count anyway.
struct RayState {
	f3 o, d;            // origin (brick units once set up) and direction
	float tx, ty, tz;   // tmax
	float dx, dy, dz;   // tdelta = |1/d|
	uint32_t p;         // current brick cell AS THE BYTE OFFSET OF ITS CUBE-FIELD ENTRY (octant plane included, see cell_offset): the
	                    // walk's lookup is a load at this offset, no address arithmetic; coordinates are recovered at candidates only
	int sx, stepy, stepz; // increment of that offset for a move along x / y / z: sign(d) * (1, row pitch, slice pitch)
	float tminn;
	f3 n;               // normal carried in/out of the traversal (voxel.cuh:135 `normal`)
	int last_step;      // offset increment of the last move (0 before the first): which axis it was, see move_axis
	uint32_t field_off;      // byte offset of the ray's octant plane in DeviceScene::cube_field
	uint32_t cube;           // edge of the empty cube ahead of the current cell (its cube_field byte)
	float distance;     // result
	bool hit;
};

enum : int { ST_NEED = 0, ST_OUTER = 1, ST_CAND = 2, ST_JUMP = 3 };

// The current cell is kept as the byte offset of its entry in the ray's octant plane of the cube field (DeviceScene::cube_field):
//     offset = octant * cf_plane + ((z + 1) * (cells + 2) + (y + 1)) * 2^cf_shift + (x + 1)
// (bordered coordinates: one border cell on every side, rows padded to a power of two).  A move is ONE add of a per-axis constant,
// a jump three multiply-adds, and the byte the walk needs next is at cube_field[offset]: round 4 kept the three coordinates in
// bit fields and spent seven vector instructions per lookup turning them into this offset (and / bfe / shift / two 24-bit
// multiplies / adds), 1.8 M + 1.0 M times per 1080p frame.  The coordinates come back out where they are needed -- at candidates,
// 0.4 M per frame: index word address, LoD distance, request -- with one exact multiply-high division by the slice pitch.
__device__ __forceinline__ uint32_t cell_offset(const DeviceScene& sc, uint32_t field_off, int x, int y, int z) {
	return field_off + __umul24(static_cast<uint32_t>(z + 1), sc.cf_pxy) + (static_cast<uint32_t>(y + 1) << sc.cf_shift) + static_cast<uint32_t>(x + 1);
}
__device__ __forceinline__ void cell_coords(const DeviceScene& sc, const RayState& r, int& x, int& y, int& z) {
	const uint32_t rel = r.p - r.field_off;                               // < cf_plane < 2^30
	const uint32_t fz = __umulhi(rel, sc.cf_magic) >> sc.cf_magic_shift;  // floor(rel / cf_pxy), exact (scene.cpp division_magic)
	const uint32_t low = rel - __umul24(fz, sc.cf_pxy);
	x = static_cast<int>(low & ((1u << sc.cf_shift) - 1u)) - 1;
	y = static_cast<int>(low >> sc.cf_shift) - 1;
	z = static_cast<int>(fz) - 1;
}
__device__ __forceinline__ int step_sign(int step) { return (step > 0) - (step < 0); }

// axis of the last move from its offset increment: +-1 = x, +- row pitch = y, +- slice pitch = z, 0 = no move yet (-1).  (A zero
// increment can only be selected for a direction with a zero component whose tmax of 1e6 is the smallest of the three:
// impossible for a unit direction inside the grid.)
__device__ __forceinline__ int move_axis(const DeviceScene& sc, int last_step) {
	const uint32_t a = static_cast<uint32_t>(last_step < 0 ? -last_step : last_step);
	return a == 0u ? -1 : (a == 1u ? 0 : (a == (1u << sc.cf_shift) ? 1 : 2));
}

// ---- cube-field walk.  DeviceScene::cube_field holds, per direction octant and brick cell, the edge n of the largest
// cube of EMPTY cells that has the cell as its near corner and extends along the octant's direction (0 = the cell itself
// holds a brick, 255 = border cell outside the grid).  A ray in that cell cannot meet a brick before one of its axes has
// moved n cells, so the walk may take up to that many steps without looking at the grid: field_jump does it in one go,
// landing on the bit-exact tmax values of the reference's cell-by-cell walk (jump.h); cubes too small to pay for a jump
// are crossed by single steps.  One byte per visited cell replaces the 16-byte block record + bit test of the mask walk.
#ifndef BM_JUMP_MIN
#define BM_JUMP_MIN 4 // smallest cube edge worth a jump (a jump costs about four single steps)
#endif
constexpr uint32_t kCubeNoJump = 0x100u; // RayState::cube flag: tmax is outside the range of jump.h, take single moves
// The lookup in three pieces -- where the cell's byte lives, whether a jump may start from the current tmax, what the byte means;
// field_lookup is their sum.
__device__ __forceinline__ uint32_t field_index(const DeviceScene& sc, const RayState& r) {
	return r.p; // the cell IS its entry's offset (cell_offset)
}
__device__ __forceinline__ bool field_jump_possible(const RayState& r) { // jump_possible(): tmax in the range jump.h handles
	const float m = fminf(fminf(r.tx, r.ty), r.tz);
	return __float_as_uint(m) - kJumpMinBits < kJumpMaxBits - kJumpMinBits;
}
__device__ __forceinline__ int field_state(uint32_t v, bool possible, uint32_t& cube) {
	// select-style, no short-circuit: a branchy version costs its full instruction count in a divergent wave anyway
	cube = possible ? v : (v | kCubeNoJump); // remembered for the walk pass, which may be several scheduler rounds away
	const int jump = static_cast<int>(v >= static_cast<uint32_t>(BM_JUMP_MIN)) & static_cast<int>(possible);
	int st = jump ? ST_JUMP : ST_OUTER;
	st = v == 0u ? ST_CAND : st;
	st = v == 255u ? ST_NEED : st; // left the grid (voxel.cuh:256): a miss
	return st;
}
__device__ __forceinline__ int field_lookup(const DeviceScene& sc, RayState& r) {
	const uint32_t v = sc.cube_field[field_index(sc, r)];
	return field_state(v, field_jump_possible(r), r.cube);
}

// voxel.cuh:249-258: one Amanatides-Woo move to the next cell.  Select-style (no per-axis branches); `t += mask ? delta : 0` is
// the reference's `tmax += mask * tdelta` for finite deltas.
__device__ __forceinline__ void step_advance(RayState& r) {
	const float tx = r.tx, ty = r.ty, tz = r.tz;
	const bool mx = tx < ty && tx < tz;
	const bool my = ty <= tx && ty < tz; // mx implies !my
	const bool mz = !(mx || my);
	const int step_x = r.sx, step_y = r.stepy, step_z = r.stepz; // scalar copies: selects between struct members pin the struct in scratch
	const int step = mx ? step_x : (my ? step_y : step_z);
	r.p += static_cast<uint32_t>(step);
	r.last_step = step;
	r.tx = tx + (mx ? r.dx : 0.f);
	r.ty = ty + (my ? r.dy : 0.f);
	r.tz = tz + (mz ? r.dz : 0.f);
}
// ... then the new cell's byte
template <bool DBG>
__device__ __forceinline__ int field_step(const DeviceScene& sc, RayState& r, Tally& tally) {
	step_advance(r);
	const int st = field_lookup(sc, r);
	if (DBG && st != ST_NEED) tally.index_loads++;
	return st;
}

// Cross the empty cube ahead of the current cell (or as much of it as the current binade of tmax allows) in one go; returns the
// number of cells moved.
// DIR: RayState::d holds the ray direction (the fused kernel); otherwise |direction| is recovered from tdelta with the
// hardware reciprocal (the queue kernels keep the direction in the queue record, not in the pooled walk state) -- it only
// feeds a quotient estimate that is corrected exactly (jump.h).
template <bool DIR = true>
__device__ __forceinline__ uint32_t jump_advance(RayState& r) {
	uint32_t cx, cy, cz;
	int axis;
	float tx = r.tx, ty = r.ty, tz = r.tz;
	const float dx = r.dx, dy = r.dy, dz = r.dz;
	const int step_x = r.sx, step_y = r.stepy, step_z = r.stepz; // scalar copies, see step_advance
	const uint32_t n = r.cube & 0xFFu; // (a cell whose brick the ray just passed through has 0: one plain move, valid anywhere)
	const float ix = DIR ? fabsf(r.d.x) : (dx > 0.f ? __builtin_amdgcn_rcpf(dx) : 0.f);
	const float iy = DIR ? fabsf(r.d.y) : (dy > 0.f ? __builtin_amdgcn_rcpf(dy) : 0.f);
	const float iz = DIR ? fabsf(r.d.z) : (dz > 0.f ? __builtin_amdgcn_rcpf(dz) : 0.f);
	dda_jump(tx, ty, tz, dx, dy, dz, ix, iy, iz, n, cx, cy, cz, axis);
	r.tx = tx; r.ty = ty; r.tz = tz;
	// all three products fit 24-bit signed multiplies: counts <= 255, increments +-1 / +- row pitch / +- slice pitch (< 2^23, Scene::init checks)
	r.p += static_cast<uint32_t>(__mul24(static_cast<int>(cx), step_x) + __mul24(static_cast<int>(cy), step_y) + __mul24(static_cast<int>(cz), step_z));
	r.last_step = axis == 0 ? step_x : (axis == 1 ? step_y : step_z); // (only read after a cube exit, where it is the exit axis)
	return cx + cy + cz;
}
template <bool DBG, bool DIR = true>
__device__ __forceinline__ int field_jump(const DeviceScene& sc, RayState& r, Tally& tally) {
	const uint32_t cells = jump_advance<DIR>(r);
	const int st = field_lookup(sc, r);
	if (DBG) tally.index_loads += cells - (st == ST_NEED ? 1u : 0u); // the cells the reference would have loaded: all but a final one outside the grid
	return st;
}

// One round of the brick-grid walk for a wave (used by the queue kernels of wavefront.hip; trace.hip inlines the same
// policy into its scheduler).  Walking lanes are of two kinds: ST_JUMP lanes have an empty cube of BM_JUMP_MIN cells or
// more ahead, ST_OUTER lanes are close to the surface.  With enough jumpers every walking lane takes the jump pass (a
// cube of edge 1-3 is crossed just the same, and a jump with n = 1 is exactly one move, valid in any cell); otherwise the
// lanes near the surface make STEPS single moves and the jumpers wait for company.  runs / lanes: DBG statistics.
#ifndef BM_JUMP_RATIO
#define BM_JUMP_RATIO 4 // a move round is a jump pass when (lanes with a cube ahead) * ratio >= (lanes near the surface)
#endif
#ifndef BM_JUMP_PASSES
#define BM_JUMP_PASSES 6 // jump passes per round while at least BM_JUMP_KEEP_NUM / BM_JUMP_KEEP_DIV of the walkers keep walking
#endif
#ifndef BM_JUMP_KEEP_NUM
#define BM_JUMP_KEEP_NUM 1
#endif
#ifndef BM_JUMP_KEEP_DIV
#define BM_JUMP_KEEP_DIV 4
#endif
template <bool DBG, int STEPS>
__device__ __forceinline__ int walk_round(const DeviceScene& sc, RayState& r, int state, int n_jump, int n_outer, Tally& tally, uint32_t& runs, uint32_t& lanes) {
	if (n_jump * BM_JUMP_RATIO >= n_outer) {
		int walkers = n_jump + n_outer;
#pragma unroll 1
		for (int pass = 0; pass < BM_JUMP_PASSES; ++pass) {
			if (DBG) { runs++; lanes += static_cast<uint32_t>(walkers); }
			if (state == ST_JUMP || state == ST_OUTER) {
				if (!(r.cube & kCubeNoJump)) state = field_jump<DBG, false>(sc, r, tally);
				else state = field_step<DBG>(sc, r, tally); // tmax outside the range jump.h handles (first move of a ray that starts on a cell face)
			}
			// another pass right away while most of the walkers are still walking: keeps the rays of a wave together on their
			// way to the next candidate and saves a scheduler round
			const int still = __popcll(__ballot(state == ST_JUMP || state == ST_OUTER));
			if (still * BM_JUMP_KEEP_DIV < walkers * BM_JUMP_KEEP_NUM || still == 0) break;
			walkers = still;
		}
	} else {
#pragma unroll 1
		for (int k = 0; k < STEPS; ++k) {
			if (DBG) { runs++; lanes += static_cast<uint32_t>(__popcll(__ballot(state == ST_OUTER))); }
			if (state == ST_OUTER) state = field_step<DBG>(sc, r, tally);
		}
	}
	return state;
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
	const int px = static_cast<int>(origin.x), py = static_cast<int>(origin.y), pz = static_cast<int>(origin.z);
	const int cells = sc.cells, cells_h = sc.cells_height;
	if (px < 0 || px >= cells || py < 0 || py >= cells || pz < 0 || pz >= cells_h) return ST_NEED;
	// octant of the direction: a zero component never moves, either plane is valid for it
	const uint32_t oct = (dir.x < 0.f ? 1u : 0u) | (dir.y < 0.f ? 2u : 0u) | (dir.z < 0.f ? 4u : 0u);
	r.field_off = oct * sc.cf_plane;
	r.p = cell_offset(sc, r.field_off, px, py, pz);
	const float cbx = dir.x > 0.f ? static_cast<float>(px + 1) : static_cast<float>(px);
	const float cby = dir.y > 0.f ? static_cast<float>(py + 1) : static_cast<float>(py);
	const float cbz = dir.z > 0.f ? static_cast<float>(pz + 1) : static_cast<float>(pz);
	const int sx = isign(dir.x), sy = isign(dir.y), sz = isign(dir.z);
	r.sx = sx; r.stepy = sy << sc.cf_shift; r.stepz = __mul24(sz, static_cast<int>(sc.cf_pxy));
	const float rx = dir.x == 0.0f ? 0.0f : 1.f / dir.x;
	const float ry = dir.y == 0.0f ? 0.0f : 1.f / dir.y;
	const float rz = dir.z == 0.0f ? 0.0f : 1.f / dir.z;
	r.tx = dir.x != 0.f ? (cbx - origin.x) * rx : 1000000.f;
	r.ty = dir.y != 0.f ? (cby - origin.y) * ry : 1000000.f;
	r.tz = dir.z != 0.f ? (cbz - origin.z) * rz : 1000000.f;
	r.dx = static_cast<float>(sx) * rx; r.dy = static_cast<float>(sy) * ry; r.dz = static_cast<float>(sz) * rz;
	r.last_step = 0;
	if (DBG) tally.index_loads++; // one per visited cell = the reference's index loads (algorithmic count)
	return field_lookup(sc, r); // inside the grid: never a border cell
}

// voxel.cuh:200-247: the current cell holds a non-empty brick -- read its index word and resolve it.
template <bool DBG>
__device__ __forceinline__ int process_candidate(const DeviceScene& sc, const int* campos, RayState& r, HitInfo& info, Tally& tally,
													 unsigned long long* lds_brick, uint32_t* walk_trips = nullptr) {
	int px, py, pz;
	cell_coords(sc, r, px, py, pz);
	const int sx = r.sx, sy = step_sign(r.stepy), sz = step_sign(r.stepz); // step signs back from the offset increments
	// inside the grid 0 <= pos < cells, so >>4 and &15 equal the reference's signed /16 and %16
	// (24-bit multiplies: all operands are far below 2^24; a 32-bit v_mul_lo_u32 issues at a quarter of the rate)
	const uint32_t sci = static_cast<uint32_t>((px >> 4) + __mul24(py >> 4, sc.sg_xy) + __mul24(pz >> 4, sc.sg_xy2));
	const uint32_t flat = (sci << 12) + static_cast<uint32_t>((px & 15) + ((py & 15) << 4) + ((pz & 15) << 8));
	// the reference's addressing (voxel.cuh:222): pool of the supercell + the 12-bit slot carried by the index word.  The
	// pool base is read together with the index word; the brick is fetched once the word says it is resident and close
	// enough to be walked at voxel level.
	const uint32_t pool = sc.pool_base[sci];
	const uint32_t index = sc.index_grid[flat];
	BrickRegs brick;
	brick.q0 = brick.q1 = brick.q2 = brick.q3 = make_uint4(0u, 0u, 0u, 0u);
	// voxel.cuh:202-206, by select: entry normal and entry distance from the axis of the last move; a ray that starts
	// inside this cell (no move yet) keeps its normal and enters at distance 0
	const int axis = move_axis(sc, r.last_step);
	const float new_distance = axis == 0 ? r.tx - r.dx : (axis == 1 ? r.ty - r.dy : (axis == 2 ? r.tz - r.dz : 0.f));
	r.n = mk(axis == -1 ? r.n.x : (axis == 0 ? -static_cast<float>(sx) : 0.f), axis == -1 ? r.n.y : (axis == 1 ? -static_cast<float>(sy) : 0.f),
			 axis == -1 ? r.n.z : (axis == 2 ? -static_cast<float>(sz) : 0.f));
	const int ddx = campos[0] - px, ddy = campos[1] - py, ddz = campos[2] - pz;
	const int lod2 = __mul24(ddx, ddx) + __mul24(ddy, ddy) + __mul24(ddz, ddz); // |dd| < 2^11: exact
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
		if (intersect_grid<2, DBG>(o2, r.d, sx, sy, sz, r.dx, r.dy, r.dz, r.n, sub_distance, brick, (index & kLodBits) >> 12, sub, tally)) {
			r.distance = new_distance * 8.f + sub_distance * 4.f + r.tminn;
			if (DBG) { info.level = 1; info.sub_id = sub; }
			r.hit = true;
			return ST_NEED;
		}
	} else if (index & kLoadedBit) {
		if (DBG) tally.brick_tests++;
		int sub = 0;
		const f3 o8 = (r.o + r.d * new_distance) * 8.f - r.n * kEpsilon;
		const uint4* bq = reinterpret_cast<const uint4*>(sc.brick_arena + (static_cast<size_t>(pool + (index & kIndexBits)) << 4));
		if (!BM_LDS_DMA) { brick.q0 = bq[0]; brick.q1 = bq[1]; brick.q2 = bq[2]; brick.q3 = bq[3]; }
		if (intersect_grid<8, DBG>(o8, r.d, sx, sy, sz, r.dx, r.dy, r.dz, r.n, sub_distance, brick, 0u, sub, tally, lds_brick, walk_trips, reinterpret_cast<const uint32_t*>(bq))) {
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
// Radiance is compared with the reference at 1e-4 relative (not bit-exact like the geometry), so the sky uses the
// hardware's ~1 ulp reciprocal / square root / exp2 instead of correctly rounded division and sqrt (a dozen
// instructions each) and libm's expf: 11 divisions / roots and 3 exponentials per evaluation.
__device__ __forceinline__ float fast_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ float fast_sqrt(float a) { return __builtin_amdgcn_sqrtf(a); }
__device__ __forceinline__ float fast_exp(float a) { return __expf(a); }
struct SkyView {
	f3 Fex;           // combined extinction factor
	float cosViewSun;
};
__device__ __forceinline__ SkyView sky_view(const FrameConstants& fc, f3 viewDir) {
	SkyView o;
	o.cosViewSun = dot(viewDir, ld3(fc.sun_direction));
	const float cosUpView = dot(mk(0.f, 0.f, 1.f), viewDir);
	const float zenith = gmax(0.0f, cosUpView);
	const float inv_zenith = __builtin_amdgcn_rcpf(zenith); // +inf below the horizon: Fex = 0, as in the reference
	const float rayleighLen = 8.4E3f * inv_zenith;
	const float mieLen = 1.25E3f * inv_zenith;
	const f3 a = ld3(fc.rayleigh) * rayleighLen + ld3(fc.mie) * mieLen;
	o.Fex = mk(fast_exp(-a.x), fast_exp(-a.y), fast_exp(-a.z));
	return o;
}
// in-scattered sky light: `sky` of sunsky.cu:109-111 (before the 0.01 / SkyFactor scaling)
__device__ __forceinline__ f3 sky_scatter(const FrameConstants& fc, const SkyView& v) {
	const float c = v.cosViewSun;
	const float rayleighPhase = (3.0f / (16.0f * kPi)) * (1.0f + c * c);
	const float g = 0.80f, g2 = 0.80f * 0.80f;
	const float base = 1.0f - 2.0f * g * c + g2;
	const float hg = (1.0f / (4.0f * kPi)) * fast_div(1.0f - g2, base * fast_sqrt(base));
	const f3 light = ld3(fc.rayleigh) * rayleighPhase + ld3(fc.mie) * hg;
	const f3 somethingElse = mk(light.x * fc.inv_total[0], light.y * fc.inv_total[1], light.z * fc.inv_total[2]) * fc.sunE;
	const f3 sky = somethingElse * mk(1.0f - v.Fex.x, 1.0f - v.Fex.y, 1.0f - v.Fex.z);
	const f3 q = somethingElse * v.Fex;
	const f3 p = mk(fast_sqrt(q.x), fast_sqrt(q.y), fast_sqrt(q.z)); // pow(x, 0.5)
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
	const float s = gmin(gmax(fast_div(v.cosViewSun - e0, e1 - e0), 0.0f), 1.0f); // smoothstep(c, c + 2e-5, x)
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

// primary_rays (kernel.cu:157-200): camera ray through pixel (x, y) for the RNG stream `seed`.
// Random2DStratifiedSample (kernel.cu:40-61), ConcentricSampleDisk (kernel.cu:85-103); lens sample drawn left to right.
__device__ __forceinline__ void primary_ray(const FrameConstants& fc, uint32_t seed, uint32_t x, uint32_t y, f3& origin, f3& direction) {
	const f3 cam_right = ld3(fc.right), cam_up = ld3(fc.up), cam_dir = ld3(fc.dir), cam_o = ld3(fc.origin);
	const float W = static_cast<float>(static_cast<uint32_t>(fc.width)), H = static_cast<float>(static_cast<uint32_t>(fc.height));
	const int stratum = static_cast<int>(random_float(seed) * (16 + 0.99999f));
	const int stratumX = stratum % 4, stratumY = (stratum / 4) % 4;
	const float jx = 0.25f * stratumX + (random_float(seed) * 0.25f);
	const float jy = 0.25f * stratumY + (random_float(seed) * 0.25f);
	const float ppx = static_cast<float>(x) - jx;
	const float ppy = static_cast<float>(y) - jy;
	const float ni = (ppx / W) - 0.5f;
	const float nj = ((H - ppy) / H) - 0.5f;
	const f3 to_focal = normalize(cam_dir + cam_right * ni + cam_up * nj);
	const f3 convergence = cam_o + to_focal * fc.focal3;
	origin = cam_o;
	if (fc.lens_radius != 0.f) {
		// with lens radius 0 the reference multiplies the disk sample by 0 and nothing reads this seed again: draws skipped
		const float l0 = random_float(seed);
		const float l1 = random_float(seed);
		float lx = 0.f, ly = 0.f;
		const float ox = 2.f * l0 - 1.f, oy = 2.f * l1 - 1.f;
		if (!(ox == 0 && oy == 0)) {
			float theta, rr;
			if (fabsf(ox) > fabsf(oy)) { rr = ox; theta = kPi / 4 * (oy / ox); }
			else { rr = oy; theta = kPi / 2 - kPi / 4 * (ox / oy); }
			float sn, cs;
			det_sincos(theta, sn, cs);
			lx = rr * cs;
			ly = rr * sn;
		}
		const float plx = fc.lens_radius * lx, ply = fc.lens_radius * ly;
		origin = cam_o + cam_right * plx + cam_up * ply;
	}
	direction = normalize(convergence - origin);
}

// cosine-weighted bounce direction of shade() (kernel.cu:281-297), RNG stream continued from the cone sample
__device__ __forceinline__ f3 bounce_direction(f3 n, uint32_t& seed) {
	const float r1 = 2.f * kPi * random_float(seed);
	const float r2 = random_float(seed);
	const float r2s = sqrtf(r2);
	// computeOrthonormalBasisNaive (kernel.cu:76-84)
	f3 u = fabs(static_cast<double>(n.x)) > .9 ? mk(0.0f, 1.0f, 0.0f) : mk(1.0f, 0.0f, 0.0f);
	// cross(u, n) of a grid normal is already a unit axis vector: dot = 1, 1 / sqrt(1) = 1, v * 1 = v -- the square root and
	// the division are only executed for the odd normal that is not axis-aligned (the identity is exact for any unit dot)
	u = cross(u, n);
	const float uu = dot(u, u);
	if (uu != 1.0f) u = u * (1.0f / sqrtf(uu));
	const f3 v = cross(n, u);
	float sn, cs;
	det_sincos(r1, sn, cs);
	return normalize(((u * cs) * r2s + (v * sn) * r2s) + n * sqrtf(1 - r2));
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

} // namespace bm
