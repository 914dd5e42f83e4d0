// kslot.h -- slot records shared by the schedules that keep path state outside the register file (trace_k.hip: K slots per
// lane, every pass from LDS; trace_h.hip: two slots per lane, the walking ray cached in registers): the one-hot slot states, the
// packed meta word of a ray record, the 2-bit-per-component form of a carried normal.  See trace_k.hip for the record layout.
#pragma once
#include "traverse.h"

namespace bm {
namespace {

enum : int { KP_GEN = 0, KP_EXT_DONE = 1, KP_SHD_DONE = 2, KP_BOUNCE = 3 };
// one-hot slot states, one byte per slot in the lane's `states` word: bit ST_* of traverse.h, plus idle
constexpr uint32_t kBitNeed = 1u << ST_NEED, kBitOuter = 1u << ST_OUTER, kBitCand = 1u << ST_CAND, kBitJump = 1u << ST_JUMP, kBitIdle = 16u;
// meta word (chunk c1.w)
constexpr uint32_t kMetaCube = 0x1FFu;       // the cell's cube-field byte | kCubeNoJump
constexpr int kMetaAxisShift = 9;            // 2 bits: axis of the last move + 1 (0 = no move yet)
constexpr int kMetaSxShift = 11, kMetaSyShift = 13, kMetaSzShift = 15; // step signs, 2-bit two's complement each
constexpr int kMetaOctShift = 17;            // 3 bits: direction octant (plane of the cube field)
constexpr int kMetaNShift = 20;              // 6 bits: the carried normal, 2 bits per component
constexpr uint32_t kMetaHit = 1u << 26, kMetaShadow = 1u << 27;

// A carried normal only ever holds +-0 and +-1 (a grid-face normal of voxel.cuh:114-118,202-206, the box-entry normal of
// :145-152, or the zero normal of a fresh ray): two bits per component, bit 1 = sign, bit 0 = magnitude one.
__device__ __forceinline__ uint32_t pack_n(f3 n) {
	const uint32_t bx = __float_as_uint(n.x), by = __float_as_uint(n.y), bz = __float_as_uint(n.z);
	const uint32_t cx = ((bx >> 31) << 1) | ((bx << 1) == 0x7F000000u ? 1u : 0u);
	const uint32_t cy = ((by >> 31) << 1) | ((by << 1) == 0x7F000000u ? 1u : 0u);
	const uint32_t cz = ((bz >> 31) << 1) | ((bz << 1) == 0x7F000000u ? 1u : 0u);
	return cx | (cy << 2) | (cz << 4);
}
__device__ __forceinline__ float unpack_n1(uint32_t q) { return __uint_as_float(((q & 2u) << 30) | ((q & 1u) ? 0x3F800000u : 0u)); }
__device__ __forceinline__ f3 unpack_n(uint32_t c) { return mk(unpack_n1(c), unpack_n1(c >> 2), unpack_n1(c >> 4)); }
__device__ __forceinline__ bool n_representable(f3 n) {
	const f3 m = unpack_n(pack_n(n));
	return __float_as_uint(m.x) == __float_as_uint(n.x) && __float_as_uint(m.y) == __float_as_uint(n.y) && __float_as_uint(m.z) == __float_as_uint(n.z);
}

// the walk's part of RayState from chunks c0 / c1 (what field_jump / field_step / field_lookup touch)
__device__ __forceinline__ void unpack_walk(const DeviceScene& sc, const uint4& c0, const uint4& c1, RayState& r) {
	r.tx = __uint_as_float(c0.x); r.ty = __uint_as_float(c0.y); r.tz = __uint_as_float(c0.z); r.p = c0.w;
	r.dx = __uint_as_float(c1.x); r.dy = __uint_as_float(c1.y); r.dz = __uint_as_float(c1.z);
	const uint32_t meta = c1.w;
	r.cube = meta & kMetaCube;
	r.sx = __builtin_amdgcn_sbfe(static_cast<int>(meta), kMetaSxShift, 2);
	r.stepy = __builtin_amdgcn_sbfe(static_cast<int>(meta), kMetaSyShift, 2) << sc.cf_shift;
	r.stepz = __mul24(__builtin_amdgcn_sbfe(static_cast<int>(meta), kMetaSzShift, 2), static_cast<int>(sc.cf_pxy));
	r.field_off = ((meta >> kMetaOctShift) & 7u) * sc.cf_plane; // (a plane of the widest world has more than 2^24 bytes: no 24-bit multiply here)
	r.last_axis = static_cast<int>((meta >> kMetaAxisShift) & 3u) - 1;
}
__device__ __forceinline__ uint32_t meta_after_walk(uint32_t meta, const RayState& r) {
	return (meta & ~(kMetaCube | (3u << kMetaAxisShift))) | r.cube | (static_cast<uint32_t>(r.last_axis + 1) << kMetaAxisShift);
}

} // namespace

} // namespace bm
