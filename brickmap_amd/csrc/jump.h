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
BM_JHD float jump_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
	return __builtin_amdgcn_rcpf(x); // ~1 ulp; the quotient below is corrected exactly
#else
	return x != 0.0f ? 1.0f / x : 0.0f; // (a zero increment belongs to an axis that cannot step; its quotient is masked)
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

// #{ j >= 0 : M + j*Q < thr }  for bit patterns of one binade (thr - M <= 2^23), Q >= 1; 0 when thr <= M.
// The quotient is at most 255 for every axis that can step inside the jump (see dda_jump), so the fp32 estimate is
// within 2^-13 of the true quotient and one exact integer correction settles it.
BM_JHD uint32_t jump_count_below(uint32_t thr, uint32_t M, uint32_t Q, float rcpQ) {
	const uint32_t a = thr - M - 1u; // garbage when thr <= M: masked at the end
	// the estimate is biased upwards by 2^-12 (the quotient is below 2^8, its error below 2^-13): it is the true floor or
	// one more, never less, so a single correction settles it
	uint32_t q = static_cast<uint32_t>(static_cast<float>(a) * rcpQ + 0.000244140625f);
	const int32_t rem = static_cast<int32_t>(a - jump_mul24(q, Q)); // q <= 2^9 here, Q <= 2^23
	q -= rem < 0 ? 1u : 0u;
	return thr > M ? q + 1u : 0u;
}

struct JumpAxis {
	uint32_t M, Q; // bit pattern of tmax, per-step increment of the bit pattern inside the current binade
	uint32_t E;    // bit pattern of tmax at the moment this axis takes its last allowed step
	float rcpQ;
};

// Per-axis set-up.  e_bits = exponent field of the smallest tmax, C = 2^e, n = steps this axis may take (>= 1).
BM_JHD JumpAxis jump_axis(float t, float d, uint32_t n, uint32_t e_bits, float C, float half_ulp) {
	JumpAxis ax;
	ax.M = jump_bits(t);
	const float Ca = C + d;                    // rounds d to a multiple of ulp(C) -- the same rounding tmax + tdelta gets in this binade
	uint32_t Q = jump_bits(Ca) - e_bits;
	Q = Q < (1u << 23) ? Q : (1u << 23);       // tdelta >= 2^e: only the current tmax is inside the binade; keeps E below 2^32
	const float r = d - (Ca - C);              // exact: the bits of tdelta below ulp(C)
	const bool irregular = jump_fabs(r) == half_ulp && (ax.M & 1u); // tie on an odd mantissa: the next increment differs from the later ones
	ax.Q = Q;
	ax.rcpQ = jump_rcp(static_cast<float>(Q));
	ax.E = ax.M + jump_mul24(irregular ? 0u : n - 1u, Q);
	return ax;
}

// Advance (tx, ty, tz) until one axis has taken n steps (exited the empty cube), or to the end of the current binade,
// whichever comes first.  cx / cy / cz = steps taken per axis (at least one in total); last_axis = axis of the final
// step (0 / 1 / 2) when the jump ended with a cube exit or an irregular step, and is only meaningful then -- a jump
// that stops at a binade end is still inside the cube (every count < n).
// Requires jump_possible(tx, ty, tz), 1 <= n <= 255, tdelta >= 0 finite.
// Returns true when the jump ended with a final step (cube exit / irregular step), false when it stopped at the binade end.
BM_JHD bool dda_jump(float& tx, float& ty, float& tz, float dx, float dy, float dz, uint32_t n, uint32_t& cx, uint32_t& cy, uint32_t& cz, int& last_axis) {
	float m = tx < ty ? tx : ty;
	m = m < tz ? m : tz;
	const uint32_t e_bits = jump_bits(m) & 0x7F800000u;
	const uint32_t B = e_bits + (1u << 23); // bit pattern of 2^(e+1): an axis whose tmax is at or above it cannot step inside this jump
	const float C = jump_float(e_bits);
	const float half_ulp = jump_float(e_bits - (24u << 23));
	const JumpAxis X = jump_axis(tx, dx, n, e_bits, C, half_ulp);
	const JumpAxis Y = jump_axis(ty, dy, n, e_bits, C, half_ulp);
	const JumpAxis Z = jump_axis(tz, dz, n, e_bits, C, half_ulp);
	// which axis reaches its last allowed step first, in the reference's order (voxel.cuh:249-252 applied to E)
	const bool mx = X.E < Y.E && X.E < Z.E;
	const bool my = Y.E <= X.E && Y.E < Z.E;
	const uint32_t E = mx ? X.E : (my ? Y.E : Z.E);
	last_axis = mx ? 0 : (my ? 1 : 2);
	// steps of axis b that come before-or-with the final step: tmax < E, or == E for an axis at least as late in the tie
	// order as the exit axis (that includes the exit axis itself); nothing at or beyond the binade end
	const uint32_t Ex = E + (mx ? 1u : 0u), Ey = E + ((mx || my) ? 1u : 0u), Ez = E + 1u;
	const uint32_t thx = Ex < B ? Ex : B, thy = Ey < B ? Ey : B, thz = Ez < B ? Ez : B;
	cx = jump_count_below(thx, X.M, X.Q, X.rcpQ);
	cy = jump_count_below(thy, Y.M, Y.Q, Y.rcpQ);
	cz = jump_count_below(thz, Z.M, Z.Q, Z.rcpQ);
	// the last addition on every axis is a real one (it may leave the binade); the ones before it follow the recurrence
	const float nx = jump_float(X.M + jump_mul24(cx - 1u, X.Q)) + dx; // (garbage for a count of 0: discarded)
	tx = cx ? nx : tx;
	const float ny = jump_float(Y.M + jump_mul24(cy - 1u, Y.Q)) + dy; // (garbage for a count of 0: discarded)
	ty = cy ? ny : ty;
	const float nz = jump_float(Z.M + jump_mul24(cz - 1u, Z.Q)) + dz; // (garbage for a count of 0: discarded)
	tz = cz ? nz : tz;
	return E < B;
}

} // namespace bm
