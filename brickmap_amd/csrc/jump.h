// jump.h -- exact multi-cell advance of the Amanatides-Woo state of src/voxel.cuh:249-258.
//
// The reference walks the brick grid one cell at a time:
//     axis = argmin(tmax) (ties: z before y before x);  pos[axis] += step[axis];  tmax[axis] = fl(tmax[axis] + tdelta[axis])
// and a hit's distance / normal are functions of the tmax values at the hit cell.  tmax is a chain of ROUNDED fp32
// additions, so skipping empty cells "geometrically" (t = (cell - origin) / dir) would change the low bits of every
// later hit.  This header advances the state over many cells at once and lands on bit-identical tmax values.
//
// Why that is possible.  While tmax[a] stays inside one binade [2^e, 2^(e+1)) with ulp u = 2^(e-23), the rounded
// addition of the constant tdelta[a] = q*u + r (|r| <= u/2) is an exact integer recurrence on the mantissa M:
//     M' = M + Q,   Q = q + (r > 0 after rounding), constant over the binade
// (round-to-nearest-even makes Q depend on the parity of M only when r is EXACTLY u/2; after one such addition M is even
// and stays even, so the recurrence is constant again -- an odd M with a tie is flagged "irregular" and gets one step).
// The three tmax values always lie within max(tdelta) of each other, i.e. usually in the SAME binade, where the bit
// patterns of positive floats are ordered like the values and differ by (difference / u): the whole walk becomes
// integer arithmetic on the bit patterns.  The walk is a 3-way merge of the arithmetic sequences M_a + j*Q_a ordered by
// (value, z before y before x); "axis a takes its n-th step" happens at E_a = M_a + (n-1)*Q_a, the first such event ends
// the jump, and the number of steps every other axis has taken by then is a floor division.  The last addition on each
// axis is carried out as a real fp32 addition, so an addition that lands in the next binade is rounded by the hardware
// exactly as in the reference.  A jump never looks past the end of the current binade of the smallest tmax: it stops
// there (all steps with tmax < 2^(e+1) taken) and the caller simply jumps again.
//
// The caller guarantees that the n cells ahead of the current cell along every axis -- the cube
// [c, c + (n-1)*step]^3 -- are empty (Scene's octant cube field), so no visited cell is skipped that the reference would
// have stopped in.  Plain C++ (host + device): tests/jump_check.cpp replays it against one-cell-at-a-time stepping.
#pragma once
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define BM_JHD __host__ __device__ __forceinline__
#else
#define BM_JHD inline
#endif

namespace bm {

BM_JHD uint32_t jump_bits(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
	return __float_as_uint(f);
#else
	uint32_t u;
	std::memcpy(&u, &f, 4);
	return u;
#endif
}
BM_JHD float jump_float(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
	return __uint_as_float(u);
#else
	float f;
	std::memcpy(&f, &u, 4);
	return f;
#endif
}
// product of two values below 2^24 (full-rate 24-bit multiplier on the device); the low 32 bits are what is used
BM_JHD uint32_t jump_mul24(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
	return __umul24(a, b);
#else
	return (a & 0xFFFFFFu) * (b & 0xFFFFFFu);
#endif
}
BM_JHD float jump_fabs(float x) { return jump_float(jump_bits(x) & 0x7FFFFFFFu); }

// A jump is worth attempting when the smallest tmax is a normal number in [2^-10, 2^19): below that the binades are
// shorter than one step anyway, above it the 1e6 sentinel of a zero direction component (voxel.cuh:185-187) could share
// a binade with a live axis.  (-0.0f, a possible first tmax, fails the test as well.)
#ifndef BM_JUMP_TMIN_EXP
#define BM_JUMP_TMIN_EXP (-10)
#endif
constexpr uint32_t kJumpMinBits = static_cast<uint32_t>(127 + BM_JUMP_TMIN_EXP) << 23, kJumpMaxBits = (127u + 19u) << 23;
BM_JHD bool jump_possible(float tx, float ty, float tz) {
	float m = tx < ty ? tx : ty;
	m = m < tz ? m : tz;
	return jump_bits(m) - kJumpMinBits < kJumpMaxBits - kJumpMinBits;
}

// ---- cost model behind the shape of this code (tools/ubench/valu_rates.hip, gfx950): plain float / integer add, sub, mul,
// fma, and / or / xor, shift right issue in ~2.6 cycles per wave; compares (5.3), selects (4.9), min / max, conversions,
// 24-bit multiplies, three-operand forms (~4.8) and the reciprocal (8.6) are the expensive ones.  So: 0 / -1 masks from
// arithmetic shifts instead of compare + select, `x & mask` instead of `cond ? x : 0`, and no reciprocal at all.

BM_JHD float jump_min3(float a, float b, float c) {
#if defined(__HIP_DEVICE_COMPILE__)
	return __builtin_fminf(__builtin_fminf(a, b), c); // v_min3_f32 (no NaN reaches a jump: jump_possible)
#else
	float m = a < b ? a : b;
	return m < c ? m : c;
#endif
}
BM_JHD uint32_t jump_umin(uint32_t a, uint32_t b) { return a < b ? a : b; }

// Advance (tx, ty, tz) until one axis has taken n steps (exited the empty cube), or to the end of the current binade,
// whichever comes first.  cx / cy / cz = steps taken per axis (at least one in total); last_axis = axis of the final
// step (0 / 1 / 2) when the jump ended with a cube exit or an irregular step, and is only meaningful then -- a jump
// that stops at a binade end is still inside the cube (every count < n).
//
// ix / iy / iz = |direction| per axis, i.e. 1 / tdelta up to rounding (tdelta = |1 / d|, voxel.cuh:180-187): the number of
// steps an axis takes below a threshold is a quotient a / Q with Q = tdelta / ulp, estimated as a * (ulp * |d|) and
// settled by one exact integer correction.  The estimate is within (a / Q) * (1 / Q + 2^-22) of the true quotient; the
// quotient is below 255 (no axis takes n <= 255 steps before the jump ends) and Q >= 2^23 / (steps taken so far + 2)
// for every axis that can still step -- tmax[a] <= (steps + 1) * tdelta[a] and the binade is that of the SMALLEST tmax --
// so with at most 1026 cells per axis (bm_scene_create refuses larger worlds) the error is below 1/32: the estimate is
// biased upwards by 1/16 and is then the true floor or one more, never less.
//
// Requires jump_possible(tx, ty, tz), n <= 255 (0 counts as 1), tdelta >= 0 finite.
// Returns true when the jump ended with a final step (cube exit / irregular step), false when it stopped at the binade end.
// bud_x / bud_y / bud_z: the step budget per axis (dda_jump below gives every axis the cube edge n; a jump that CONTINUES after a
// binade stop hands each axis what is left of it, n - steps already taken, see field_jump).
BM_JHD bool dda_jump3(float& tx, float& ty, float& tz, float dx, float dy, float dz, float ix, float iy, float iz, uint32_t bud_x, uint32_t bud_y, uint32_t bud_z,
					  uint32_t& cx, uint32_t& cy, uint32_t& cz, int& last_axis) {
	const float m = jump_min3(tx, ty, tz);
	const uint32_t e_bits = jump_bits(m) & 0x7F800000u;
	const uint32_t B = e_bits + (1u << 23); // bit pattern of 2^(e+1): an axis whose tmax is at or above it cannot step inside this jump
	const float C = jump_float(e_bits);
	const float ulp = jump_float(e_bits - (23u << 23));
	const float half_ulp = jump_float(e_bits - (24u << 23));
	const uint32_t k0x = bud_x ? bud_x - 1u : 0u, k0y = bud_y ? bud_y - 1u : 0u, k0z = bud_z ? bud_z - 1u : 0u; // steps before the last allowed one
	const uint32_t Mx = jump_bits(tx), My = jump_bits(ty), Mz = jump_bits(tz);
	// per axis: Q = increment of the bit pattern per step inside this binade (C + d rounds d to a multiple of ulp(C) -- the
	// same rounding tmax + tdelta gets), capped at 2^23 (tdelta >= 2^e: only the current tmax is inside the binade; keeps
	// every product below 2^32 and every factor below 2^24); r = the bits of tdelta below ulp(C), exact
	const float Cx = C + dx, Cy = C + dy, Cz = C + dz;
	const uint32_t Qx = jump_umin(jump_bits(Cx) - e_bits, 1u << 23), Qy = jump_umin(jump_bits(Cy) - e_bits, 1u << 23),
				   Qz = jump_umin(jump_bits(Cz) - e_bits, 1u << 23);
	const float rx = dx - (Cx - C), ry = dy - (Cy - C), rz = dz - (Cz - C);
	uint32_t kx = k0x, ky = k0y, kz = k0z;
	// a tie (r exactly half an ulp) on an odd mantissa: the next increment differs from the later ones -> one step only.
	// Rare (tdelta needs a run of zero bits), so the wave branches around it.
	const bool tie_x = jump_fabs(rx) == half_ulp, tie_y = jump_fabs(ry) == half_ulp, tie_z = jump_fabs(rz) == half_ulp;
	if (tie_x | tie_y | tie_z) {
		kx = (tie_x && (Mx & 1u)) ? 0u : k0x;
		ky = (tie_y && (My & 1u)) ? 0u : k0y;
		kz = (tie_z && (Mz & 1u)) ? 0u : k0z;
	}
	// bit pattern of tmax at the moment each axis takes its last allowed step; the earliest of them ends the jump, in the
	// reference's order (voxel.cuh:249-252: at equal tmax z moves before y before x)
	const uint32_t Ex = Mx + jump_mul24(kx, Qx), Ey = My + jump_mul24(ky, Qy), Ez = Mz + jump_mul24(kz, Qz);
	const uint32_t E = jump_umin(jump_umin(Ex, Ey), Ez);
	const uint32_t nz = jump_umin(Ez - E, 1u), ny = jump_umin(Ey - E, 1u); // 1: the axis is NOT the one that ends the jump at E
	const uint32_t nyz = nz & ny;                                          // 1: x ends it
	last_axis = static_cast<int>(2u - nz - nyz);
	// steps of axis b that come before-or-with the final step: tmax < E, or == E for an axis that moves before the exit
	// axis at equal tmax (that includes the exit axis itself); nothing at or beyond the binade end
	const uint32_t thx = jump_umin(E + nyz, B), thy = jump_umin(E + nz, B), thz = jump_umin(E + 1u, B);
	// count = #{ j >= 0 : M + j*Q < th } = floor((th - M - 1) / Q) + 1 when th > M, else 0
	const float bias = 0.0625f;
	// all-ones where the axis steps at all (th > M; both are bit patterns below 2^31), zero where it does not
	const uint32_t vx = static_cast<uint32_t>(static_cast<int32_t>(Mx - thx) >> 31), vy = static_cast<uint32_t>(static_cast<int32_t>(My - thy) >> 31),
				   vz = static_cast<uint32_t>(static_cast<int32_t>(Mz - thz) >> 31);
	const uint32_t ax_ = (thx + ~Mx) & vx, ay_ = (thy + ~My) & vy, az_ = (thz + ~Mz) & vz; // th - M - 1, or 0 for an axis that does not step
	uint32_t qx = static_cast<uint32_t>(static_cast<float>(ax_) * (ulp * ix) + bias);
	uint32_t qy = static_cast<uint32_t>(static_cast<float>(ay_) * (ulp * iy) + bias);
	uint32_t qz = static_cast<uint32_t>(static_cast<float>(az_) * (ulp * iz) + bias);
	// exact correction: one less when the remainder is negative (the arithmetic shift yields the -1); q = steps before the last one
	qx += static_cast<uint32_t>(static_cast<int32_t>(ax_ - jump_mul24(qx, Qx)) >> 31);
	qy += static_cast<uint32_t>(static_cast<int32_t>(ay_ - jump_mul24(qy, Qy)) >> 31);
	qz += static_cast<uint32_t>(static_cast<int32_t>(az_ - jump_mul24(qz, Qz)) >> 31);
	cx = (qx + 1u) & vx; cy = (qy + 1u) & vy; cz = (qz + 1u) & vz;
	// the last addition on every axis is a real one (it may leave the binade); the ones before it follow the recurrence.
	// An axis that does not step adds +0 to its (positive) tmax.
	tx = jump_float(Mx + jump_mul24(qx, Qx)) + jump_float(jump_bits(dx) & vx);
	ty = jump_float(My + jump_mul24(qy, Qy)) + jump_float(jump_bits(dy) & vy);
	tz = jump_float(Mz + jump_mul24(qz, Qz)) + jump_float(jump_bits(dz) & vz);
	return E < B;
}

BM_JHD bool dda_jump(float& tx, float& ty, float& tz, float dx, float dy, float dz, float ix, float iy, float iz, uint32_t n, uint32_t& cx,
					 uint32_t& cy, uint32_t& cz, int& last_axis) {
	return dda_jump3(tx, ty, tz, dx, dy, dz, ix, iy, iz, n, n, n, cx, cy, cz, last_axis);
}

} // namespace bm
