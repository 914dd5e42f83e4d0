// trace_k.hip -- "K-slot lanes": the fused per-pixel path tracer of trace.hip with K paths per lane and the path state
// OUTSIDE the register file (gfx950, wave64, hand-written HIP).
//
// Why.  trace.hip keeps one path per lane in registers.  A wave's 64 paths want different things -- a walk pass, a
// candidate pass, a shade pass -- and a pass costs the same with 15 or 64 lanes active, so the vector pipes run saturated at
// ~20 of 64 lanes per instruction (docs/HISTORY.md 5.4: the bound of a one-path-per-lane state machine is ~1/k for k pass types of
// equal weight).  Here every lane owns K paths ("slots"); a pass serves a lane if ANY of its slots wants it, so a pass that
// one path in three wants is wanted by 1 - (2/3)^K of the lanes.  Two paths per lane in REGISTERS cost the occupancy that hides
// the walk's load latencies (round 2: 170-250 VGPRs, 2.6 ms); with the state in LDS a pass holds only its own working set:
//
//   LDS, per slot 64 bytes = four 16-byte chunks, chunk c of slot k of thread t at lds[(k * CH + c) * 256 + t] (every access is
//   lane-contiguous: ds_read_b128 / ds_write_b128 without bank conflicts whatever slot each lane picks):
//     c0  tmax.xyz, packed cell            c1  tdelta.xyz, meta (cube byte, last axis, step signs, octant, normal, hit, kind)
//     c2  origin (brick units), tminn      c3  direction, hit distance
//   a walk pass reads c0 c1 c3 and writes c0 + meta; a candidate pass reads all four, stages the 64-byte brick IN the slot's own
//   four chunks (they are in registers by then), walks it, writes the four chunks back; a shade pass reads c1 c3 and writes all;
//   global scratch (L2-resident), per slot four 16-byte chunks, chunk-major and thread-contiguous (coalesced):
//     hit point + path flags | accumulator | shadow colour + pixel | bounce direction + local pixel   -- shade passes only;
//   registers across passes: ONE word of one-hot slot states per lane (a byte per slot) and the wave-uniform scheduler state.
//
// K * 64 bytes of LDS per lane bound the occupancy: K = 2 -> 4 waves per SIMD, K = 3 -> 3, K = 4 -> 2 (160 KiB per CU).
// tools/sim/sched_sim.cpp replays the real paths of bench config 2 through this organisation: see docs/HISTORY.md 5.5.
//
// Per-ray arithmetic is that of trace.hip -- the same device functions of traverse.h / jump.h, the same operands in the same
// order -- so hit records are bit-identical to the CPU oracle; scheduling changes WHEN a path's operations happen, never what
// they are (reference: src/kernel.cu:154-346, src/voxel.cuh:135-261).
#include "kslot.h"

namespace bm {

#ifndef BM_K
#define BM_K 3
#endif
#ifndef BM_K_WAVES
#define BM_K_WAVES (BM_K <= 2 ? 4 : (BM_K == 3 ? 3 : 2))
#endif
#ifndef BM_K_REFILL_MIN
#define BM_K_REFILL_MIN 16 // lanes with an idle slot before the wave takes tickets
#endif
// pass choice: the type with the most lanes wanting it, candidates and shade passes weighted (num / 4)
#ifndef BM_K_WB
#define BM_K_WB 4
#endif
#ifndef BM_K_WC
#define BM_K_WC 4
#endif
#ifndef BM_K_PRIO
#define BM_K_PRIO 1
#endif
#ifndef BM_K_ITEM_LANES
#define BM_K_ITEM_LANES 4
#endif

namespace {
constexpr uint32_t kRepAll = BM_K >= 4 ? 0x01010101u : (BM_K == 3 ? 0x00010101u : (BM_K == 2 ? 0x00000101u : 0x00000001u));
} // namespace

template <bool DBG>
__global__ __launch_bounds__(256, DBG ? 2 : BM_K_WAVES) void trace_paths_k(const DeviceScene sc, const FrameConstants* __restrict__ fcp, float4* __restrict__ accum,
																		 uint32_t* __restrict__ dbg, DeviceCounters* __restrict__ counters,
																		 uint32_t* __restrict__ work_counter, uint4* __restrict__ cold) {
	constexpr int K = BM_K;
	constexpr uint32_t CH = DBG ? 5u : 4u; // LDS chunks per slot (instrumented: + cells visited, hit level / brick / voxel)
	constexpr uint32_t CC = DBG ? 6u : 4u; // scratch chunks per slot (instrumented: + first-hit record, path hashes)
	__shared__ uint4 lds[K * CH * 256];
	const FrameConstants& fc = *fcp;
	const uint32_t tid = threadIdx.x;
	const int lane = static_cast<int>(tid & 63u);
	const uint32_t nthreads = gridDim.x * 256u, gtid = blockIdx.x * 256u + tid;
	const uint32_t W = static_cast<uint32_t>(fc.width), H = static_cast<uint32_t>(fc.height);
	const uint32_t total_chunks = static_cast<uint32_t>(fc.tiles_x) * static_cast<uint32_t>(fc.tiles_y) * 16u;
	const bool sample_items = (fc.flags & 4u) != 0u; // BM_FLAG_SAMPLE_ITEMS (trace.hip "work items")
	constexpr uint32_t kParts = 16u / BM_K_ITEM_LANES;
	const uint32_t items_per_chunk = (sample_items ? static_cast<uint32_t>(fc.spp > 0 ? fc.spp : 1) : 1u) * kParts;

	uint32_t states = kBitIdle * kRepAll; // one-hot state byte per slot
	Tally tally;

	bool work_left = true;
	constexpr uint32_t kCounters = 8, kCounterStride = 32;
	int my_counter = static_cast<int>((blockIdx.x * 4u + static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6)))) % kCounters);
	int counters_done = 0;
	const long long round_budget = (static_cast<long long>(total_chunks) + 64) * (static_cast<long long>(fc.spp) + 1) * (fc.max_bounces + 2) *
								   (2ll * sc.cells + sc.cells_height + 64);
	long long rounds_left = static_cast<long long>((static_cast<unsigned long long>(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(round_budget >> 32)))) << 32) |
												   static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(round_budget))));
	uint32_t runsA = 0, lanesA = 0, runsB = 0, lanesB = 0, runsC = 0, lanesC = 0, runsJ = 0, lanesJ = 0;
	unsigned long long cycA = 0, cycB = 0, cycC = 0;
	const unsigned long long t_begin = DBG ? __builtin_amdgcn_s_memtime() : 0ull;
	unsigned long long t_dry = 0ull;

	for (;;) {
		// `any`: the OR of the lane's slot states -- what the lane can take part in
		uint32_t any = states | (states >> 8);
		if (K > 2) any |= states >> 16;
		if (K > 3) any |= states >> 24;
		// ---- refill: every lane with an idle slot takes one pixel into its first idle slot (BM_K_ITEM_LANES lanes share a ticket)
		const unsigned long long idle = __ballot((any & kBitIdle) != 0u);
		const int nI = __popcll(idle);
		if (work_left && nI >= BM_K_REFILL_MIN) {
			const int want = nI / BM_K_ITEM_LANES;
			uint32_t base = 0;
			if (lane == 0) base = atomicAdd(work_counter + my_counter * kCounterStride, static_cast<uint32_t>(want));
			base = __builtin_amdgcn_readfirstlane(base);
			const uint32_t total_groups = (total_chunks + 3u) >> 2;
			const uint32_t my_groups = total_groups > static_cast<uint32_t>(my_counter) ? (total_groups - static_cast<uint32_t>(my_counter) + kCounters - 1u) / kCounters : 0u;
			const uint32_t my_tickets = my_groups * 4u * items_per_chunk;
			const uint32_t counter_now = static_cast<uint32_t>(my_counter);
			if (base + want >= my_tickets) {
				my_counter = (my_counter + 1) % static_cast<int>(kCounters);
				if (++counters_done >= static_cast<int>(kCounters)) { work_left = false; if (DBG) t_dry = __builtin_amdgcn_s_memtime(); }
			}
			const int rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(idle >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(idle), 0u));
			if ((any & kBitIdle) != 0u && rank < want * BM_K_ITEM_LANES) {
				const uint32_t item = base + static_cast<uint32_t>(rank / BM_K_ITEM_LANES);
				const uint32_t ticket = item / items_per_chunk, item_sub = item - ticket * items_per_chunk;
				const uint32_t item_sample = item_sub / kParts, part = item_sub % kParts;
				const uint32_t chunk = ((ticket >> 2) * kCounters + counter_now) * 4u + (ticket & 3u);
				if (item < my_tickets && chunk < total_chunks) {
					const uint32_t tile = chunk >> 4, kk = chunk & 15u;
					const int tile_x = static_cast<int>(tile % static_cast<uint32_t>(fc.tiles_x));
					const int tile_y = static_cast<int>(tile / static_cast<uint32_t>(fc.tiles_x));
					const int cx = static_cast<int>((kk & 1u) | ((kk >> 1) & 2u)), cy = static_cast<int>(((kk >> 1) & 1u) | ((kk >> 2) & 2u));
					const uint32_t q = part * BM_K_ITEM_LANES + (static_cast<uint32_t>(rank) % BM_K_ITEM_LANES);
					const int x = tile_x * 16 + cx * 4 + static_cast<int>(q & 3u);
					const int ly = tile_y * 16 + cy * 4 + static_cast<int>(q >> 2);
					const int y = ((ly / fc.band_rows) * fc.shard_count + fc.shard_rank) * fc.band_rows + ly % fc.band_rows;
					if (x < fc.width && ly < fc.local_rows && y < fc.height) {
						const uint32_t k = static_cast<uint32_t>(__builtin_ctz(states & (kBitIdle * kRepAll))) >> 3;
						const uint32_t cb = k * CC * nthreads + gtid;
						const uint32_t local_pixel = static_cast<uint32_t>(ly) * W + static_cast<uint32_t>(x);
						const uint32_t s0 = sample_items ? item_sample : 0u;
						const float4 a0 = sample_items ? make_float4(0.f, 0.f, 0.f, 0.f) : accum[local_pixel];
						cold[cb] = make_uint4(0u, 0u, 0u, s0 | (static_cast<uint32_t>(KP_GEN) << 21)); // hit point (unset), flags: sample, bounces 0, P_GEN
						cold[cb + nthreads] = make_uint4(__float_as_uint(a0.x), __float_as_uint(a0.y), __float_as_uint(a0.z), __float_as_uint(a0.w));
						cold[cb + 2u * nthreads] = make_uint4(0u, 0u, 0u, static_cast<uint32_t>(x) | (static_cast<uint32_t>(y) << 16));
						cold[cb + 3u * nthreads] = make_uint4(0u, 0u, 0u, local_pixel);
						if (DBG) {
							cold[cb + 4u * nthreads] = make_uint4(0u, 0u, 0xFFFFFFFFu, 0u);
							cold[cb + 5u * nthreads] = make_uint4(2166136261u, 2166136261u, 0u, 0u);
							lds[(k * CH + 4u) * 256u + tid] = make_uint4(0u, 0u, 0xFFFFFFFFu, 0u); // cells visited by this item's rays; hit level / brick / voxel
						}
						// (no ray yet: the slot's meta word only has to say "not a shadow ray"; the shade pass generates the primary ray)
						lds[(k * CH + 1u) * 256u + tid] = make_uint4(0u, 0u, 0u, 0u);
						states = (states & ~(0xFFu << (8u * k))) | (kBitNeed << (8u * k));
					}
				}
			}
			any = states | (states >> 8);
			if (K > 2) any |= states >> 16;
			if (K > 3) any |= states >> 24;
		}
		const int nJ = __popcll(__ballot((any & kBitJump) != 0u));
		const int nO = __popcll(__ballot((any & (kBitJump | kBitOuter)) == kBitOuter)); // lanes whose only walkers are near the surface
		const int nA = nJ + nO;
		const int nB = __popcll(__ballot((any & kBitCand) != 0u));
		const int nC = __popcll(__ballot((any & kBitNeed) != 0u));
		const int live = __popcll(__ballot((any & (kBitJump | kBitOuter | kBitCand | kBitNeed)) != 0u));
		--rounds_left;
		if (rounds_left < 0 || (live == 0 && !work_left)) break;
		// Policy: the pass type wanted by the most lanes runs, candidate and shade passes weighted (they are the expensive ones,
		// but starving them starves the walk of new rays)
		int phase; // 0 = walk, 1 = candidates, 2 = shade
		{
			const int vA = nA * 4, vB = nB * BM_K_WB, vC = nC * BM_K_WC;
			phase = (vC >= vA && vC >= vB) ? 2 : (vB >= vA ? 1 : 0);
		}
		const unsigned long long t_phase = DBG ? __builtin_amdgcn_s_memtime() : 0ull;
		if (phase == 2) {
			if (BM_K_PRIO) __builtin_amdgcn_s_setprio(0);
			if (DBG) { runsC++; lanesC += nC; }
			// ================= shade / connect / next primary ray, then ONE ray set-up (trace.hip phase C)
			const uint32_t m = states & (kBitNeed * kRepAll);
			if (m != 0u) {
				const uint32_t k = static_cast<uint32_t>(__builtin_ctz(m)) >> 3;
				const uint32_t sb = k * CH * 256u + tid;
				const uint32_t cb = k * CC * nthreads + gtid;
				const uint4 q0 = cold[cb], q1 = cold[cb + nthreads], q2 = cold[cb + 2u * nthreads], q3 = cold[cb + 3u * nthreads];
				const uint4 c1 = lds[sb + 256u], c3 = lds[sb + 768u];
				uint4 g4 = make_uint4(0u, 0u, 0u, 0u), g5 = g4, l4 = g4;
				if (DBG) { g4 = cold[cb + 4u * nthreads]; g5 = cold[cb + 5u * nthreads]; l4 = lds[sb + 1024u]; }
				f3 hitp = mk(__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z));
				const uint32_t flags = q0.w;
				float4 acc = make_float4(__uint_as_float(q1.x), __uint_as_float(q1.y), __uint_as_float(q1.z), __uint_as_float(q1.w));
				f3 scolor = mk(__uint_as_float(q2.x), __uint_as_float(q2.y), __uint_as_float(q2.z));
				const uint32_t xy = q2.w;
				f3 bdir = mk(__uint_as_float(q3.x), __uint_as_float(q3.y), __uint_as_float(q3.z));
				const uint32_t local_pixel = q3.w;
				int s = static_cast<int>(flags & 0xFFFFu);
				const int s_end = sample_items ? s + 1 : fc.spp;
				int bounces = static_cast<int>((flags >> 16) & 15u);
				bool terminated = ((flags >> 20) & 1u) != 0u;
				int pstate = static_cast<int>((flags >> 21) & 3u);
				f3 pn = unpack_n(flags >> 23);
				const uint32_t p = (xy >> 16) * W + (xy & 0xFFFFu);
				const uint32_t meta_in = c1.w;
				RayState r;
				r.hit = (meta_in & kMetaHit) != 0u;
				r.n = unpack_n(meta_in >> kMetaNShift);
				r.d = mk(__uint_as_float(c3.x), __uint_as_float(c3.y), __uint_as_float(c3.z));
				r.distance = __uint_as_float(c3.w);
				uint32_t d0 = g4.x, d1 = g4.y, d2 = g4.z, d3 = g4.w, hseg = g5.x, hsh = g5.y, next = g5.z & 0xFFFFu, nsh = g5.z >> 16;
				HitInfo info;
				if (DBG) { info.level = static_cast<int>(l4.y); info.brick_id = static_cast<int>(l4.z); info.sub_id = static_cast<int>(l4.w); }
				const uint32_t loads_before = tally.index_loads;

				bool shadow = false;
				bool need_setup = false, finished = false;
				f3 ro = mk(0.f, 0.f, 0.f), rd = mk(0.f, 0.f, 0.f);
				if (pstate == KP_SHD_DONE) {
					// ---- connect (kernel.cu:328-346)
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
					if (!occluded) { acc.x += scolor.x; acc.y += scolor.y; acc.z += scolor.z; }
					if (terminated) {
						s++;
						pstate = KP_GEN;
					} else {
						bounces++;
						pstate = KP_BOUNCE;
						ro = hitp; rd = bdir; r.n = pn; shadow = false; need_setup = true;
					}
				}
				if (pstate == KP_EXT_DONE) {
					// ---- extend finished (kernel.cu:226-238)
					const bool is_hit = r.hit;
					pn = r.n;
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
					const bool primary_only = fc.flags & 1u;
					f3 view = r.d;
					f3 miss_color = mk(0.f, 0.f, 0.f);
					float sunLight = 0.f;
					bool cast = false;
					if (is_hit && !primary_only) {
						// ---- shade, hit branch (kernel.cu:255-302)
						const uint32_t frame = fc.base_frame + static_cast<uint32_t>(bounces);
						const uint32_t slot = p + static_cast<uint32_t>(fc.sample_base + s) * W * H;
						uint32_t sseed = (frame * p * 147565741u) * 720898027u * slot;
						hitp = hitp + r.d * r.distance;
						hitp = hitp + pn * 2.f * kEpsilon;
						view = cone_sample(fc, sseed);
						sunLight = dot(pn, view);
						cast = sunLight > 0.f;
						terminated = !(bounces < fc.max_bounces);
						if (terminated) acc.w += 1.f; // kernel.cu:301
						else bdir = bounce_direction(pn, sseed);
						if (!cast) {
							if (terminated) { s++; pstate = KP_GEN; }
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
					if (!is_hit || primary_only) { // the path ends here
						acc.x += miss_color.x; acc.y += miss_color.y; acc.z += miss_color.z;
						acc.w += 1.f;
						s++;
						pstate = KP_GEN;
					}
				}
				if (pstate == KP_GEN) {
					if (s >= s_end) {
						// item finished: write the accumulator back (or add this sample's share); the slot is idle again
						if (sample_items) {
							float* a = reinterpret_cast<float*>(accum + local_pixel);
							unsafeAtomicAdd(a + 0, acc.x); unsafeAtomicAdd(a + 1, acc.y); unsafeAtomicAdd(a + 2, acc.z); unsafeAtomicAdd(a + 3, acc.w);
						} else {
							accum[local_pixel] = acc;
						}
						if (DBG && dbg) {
							uint32_t* d = dbg + static_cast<size_t>(local_pixel) * 8;
							const uint32_t cells = l4.x + (tally.index_loads - loads_before);
							if (sample_items) {
								if (s_end == 1) { d[0] = d0; d[1] = d1; d[2] = d2; d[3] = d3; }
								atomicAdd(d + 4, hseg); atomicAdd(d + 5, hsh); atomicAdd(d + 6, next | (nsh << 16));
								atomicAdd(d + 7, cells);
							} else {
								d[0] = d0; d[1] = d1; d[2] = d2; d[3] = d3; d[4] = hseg; d[5] = hsh; d[6] = next | (nsh << 16);
								d[7] = cells;
							}
						}
						finished = true;
					} else {
						// ---- primary_rays (kernel.cu:157-200) for queue slot `slot`, start_position 0
						const uint32_t slot = p + static_cast<uint32_t>(fc.sample_base + s) * W * H;
						uint32_t seed = (fc.base_frame * 147565741u) * 720898027u * slot;
						primary_ray(fc, seed, xy & 0xFFFFu, xy >> 16, hitp, rd);
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
				uint32_t new_bit = kBitIdle;
				if (need_setup) {
					if (shadow) r.n = mk(0.f, 0.f, 0.f); // connect passes a zeroed normal (kernel.cu:338)
					pstate = shadow ? KP_SHD_DONE : KP_EXT_DONE;
					r.tx = r.ty = r.tz = r.dx = r.dy = r.dz = 0.f; r.p = 0u; r.sx = r.stepy = r.stepz = 0; r.tminn = 0.f; r.cube = 0u; r.field_off = 0u;
					r.o = mk(0.f, 0.f, 0.f); r.distance = 0.f;
					const int st = ray_setup<DBG>(sc, ro, rd, r, tally);
					new_bit = 1u << st;
					if (DBG && !n_representable(r.n)) tally.paths += 1u << 20; // (cannot happen: the packed normal would lose it -- poisons the path counter the tests compare)
					const uint32_t oct = (rd.x < 0.f ? 1u : 0u) | (rd.y < 0.f ? 2u : 0u) | (rd.z < 0.f ? 4u : 0u);
					const uint32_t meta = (r.cube & kMetaCube) | ((static_cast<uint32_t>(r.sx) & 3u) << kMetaSxShift) | ((static_cast<uint32_t>(step_sign(r.stepy)) & 3u) << kMetaSyShift) |
										  ((static_cast<uint32_t>(step_sign(r.stepz)) & 3u) << kMetaSzShift) | (oct << kMetaOctShift) | (pack_n(r.n) << kMetaNShift) |
										  (shadow ? kMetaShadow : 0u);
					lds[sb] = make_uint4(__float_as_uint(r.tx), __float_as_uint(r.ty), __float_as_uint(r.tz), r.p);
					lds[sb + 256u] = make_uint4(__float_as_uint(r.dx), __float_as_uint(r.dy), __float_as_uint(r.dz), meta);
					lds[sb + 512u] = make_uint4(__float_as_uint(r.o.x), __float_as_uint(r.o.y), __float_as_uint(r.o.z), __float_as_uint(r.tminn));
					lds[sb + 768u] = make_uint4(__float_as_uint(rd.x), __float_as_uint(rd.y), __float_as_uint(rd.z), 0u);
				}
				if (!finished) {
					const uint32_t nf = static_cast<uint32_t>(s) | (static_cast<uint32_t>(bounces) << 16) | (terminated ? 1u << 20 : 0u) | (static_cast<uint32_t>(pstate) << 21) |
										(pack_n(pn) << 23);
					if (DBG && !n_representable(pn)) tally.paths += 1u << 20;
					cold[cb] = make_uint4(__float_as_uint(hitp.x), __float_as_uint(hitp.y), __float_as_uint(hitp.z), nf);
					cold[cb + nthreads] = make_uint4(__float_as_uint(acc.x), __float_as_uint(acc.y), __float_as_uint(acc.z), __float_as_uint(acc.w));
					cold[cb + 2u * nthreads] = make_uint4(__float_as_uint(scolor.x), __float_as_uint(scolor.y), __float_as_uint(scolor.z), xy);
					cold[cb + 3u * nthreads] = make_uint4(__float_as_uint(bdir.x), __float_as_uint(bdir.y), __float_as_uint(bdir.z), local_pixel);
					if (DBG) {
						cold[cb + 4u * nthreads] = make_uint4(d0, d1, d2, d3);
						cold[cb + 5u * nthreads] = make_uint4(hseg, hsh, next | (nsh << 16), 0u);
						lds[sb + 1024u] = make_uint4(l4.x + (tally.index_loads - loads_before), 0u, 0xFFFFFFFFu, 0u);
					}
				}
				states = (states & ~(0xFFu << (8u * k))) | (new_bit << (8u * k));
			}
		} else if (phase == 1) {
			if (BM_K_PRIO) __builtin_amdgcn_s_setprio(2);
			if (DBG) { runsB++; lanesB += nB; }
			// ================= candidates: index word, LoD / 8^3 bitmask DDA, streaming request (trace.hip phase B)
			const uint32_t m = states & (kBitCand * kRepAll);
			if (m != 0u) {
				const uint32_t k = static_cast<uint32_t>(__builtin_ctz(m)) >> 3;
				const uint32_t sb = k * CH * 256u + tid;
				const uint4 c0 = lds[sb], c1 = lds[sb + 256u], c2 = lds[sb + 512u], c3 = lds[sb + 768u];
				uint4 l4 = make_uint4(0u, 0u, 0u, 0u);
				if (DBG) l4 = lds[sb + 1024u];
				RayState r;
				unpack_walk(sc, c0, c1, r);
				const uint32_t meta_in = c1.w;
				r.o = mk(__uint_as_float(c2.x), __uint_as_float(c2.y), __uint_as_float(c2.z)); r.tminn = __uint_as_float(c2.w);
				r.d = mk(__uint_as_float(c3.x), __uint_as_float(c3.y), __uint_as_float(c3.z)); r.distance = __uint_as_float(c3.w);
				r.n = unpack_n(meta_in >> kMetaNShift);
				r.hit = false;
				HitInfo info;
				// the brick is staged in the slot's own four chunks: slice z is the 64-bit half (z & 1) of chunk z >> 1
				unsigned long long* stage = reinterpret_cast<unsigned long long*>(lds + sb);
				const int st = process_candidate<DBG, true, true>(sc, fc.campos, r, info, tally, stage);
				if (DBG && !n_representable(r.n)) tally.paths += 1u << 20;
				const uint32_t meta = (meta_in & ~((63u << kMetaNShift) | kMetaHit)) | (pack_n(r.n) << kMetaNShift) | (r.hit ? kMetaHit : 0u);
				lds[sb] = c0;
				lds[sb + 256u] = make_uint4(c1.x, c1.y, c1.z, meta);
				lds[sb + 512u] = c2;
				lds[sb + 768u] = make_uint4(c3.x, c3.y, c3.z, __float_as_uint(r.distance));
				if (DBG && r.hit) lds[sb + 1024u] = make_uint4(l4.x, static_cast<uint32_t>(info.level), static_cast<uint32_t>(info.brick_id), static_cast<uint32_t>(info.sub_id));
				states = (states & ~(0xFFu << (8u * k))) | ((1u << st) << (8u * k));
			}
		} else {
			// ================= the brick-grid walk (trace.hip phase A): jump passes in bursts, or single moves near the surface
			if (BM_K_PRIO) __builtin_amdgcn_s_setprio(1);
			if (nJ * BM_JUMP_RATIO >= nO) {
				int walkers = nA;
#pragma unroll 1
				for (int pass = 0; pass < BM_JUMP_PASSES; ++pass) {
					if (DBG) { runsJ++; lanesJ += walkers; }
					// a lane's jumper goes first, else one of its slots near the surface rides along (a jump with n = 1 is one move)
					const uint32_t mj = states & (kBitJump * kRepAll), mo = states & (kBitOuter * kRepAll);
					const uint32_t m = mj != 0u ? mj : mo;
					if (m != 0u) {
						const uint32_t k = static_cast<uint32_t>(__builtin_ctz(m)) >> 3;
						const uint32_t sb = k * CH * 256u + tid;
						const uint4 c0 = lds[sb], c1 = lds[sb + 256u], c3 = lds[sb + 768u];
						RayState r;
						unpack_walk(sc, c0, c1, r);
						r.d = mk(__uint_as_float(c3.x), __uint_as_float(c3.y), __uint_as_float(c3.z));
						const uint32_t loads_before = tally.index_loads;
						int st;
						if (!(r.cube & kCubeNoJump)) st = field_jump<DBG>(sc, r, tally);
						else st = field_step<DBG>(sc, r, tally);
						lds[sb] = make_uint4(__float_as_uint(r.tx), __float_as_uint(r.ty), __float_as_uint(r.tz), r.p);
						reinterpret_cast<uint32_t*>(lds + sb + 256u)[3] = meta_after_walk(c1.w, r);
						if (DBG) reinterpret_cast<uint32_t*>(lds + sb + 1024u)[0] += tally.index_loads - loads_before;
						states = (states & ~(0xFFu << (8u * k))) | ((1u << st) << (8u * k));
					}
					if (BM_JUMP_PASSES > 1) {
						uint32_t a2 = states | (states >> 8);
						if (K > 2) a2 |= states >> 16;
						if (K > 3) a2 |= states >> 24;
						const int still = __popcll(__ballot((a2 & (kBitJump | kBitOuter)) != 0u));
						if (still * BM_JUMP_KEEP_DIV < walkers * BM_JUMP_KEEP_NUM || still == 0) break;
						walkers = still;
					}
				}
			} else {
				if (DBG) { runsA++; lanesA += nO; }
				const uint32_t m = states & (kBitOuter * kRepAll);
				if (m != 0u) {
					const uint32_t k = static_cast<uint32_t>(__builtin_ctz(m)) >> 3;
					const uint32_t sb = k * CH * 256u + tid;
					const uint4 c0 = lds[sb], c1 = lds[sb + 256u];
					RayState r;
					unpack_walk(sc, c0, c1, r);
					const uint32_t loads_before = tally.index_loads;
					int st = ST_OUTER;
#pragma unroll 1
					for (int i = 0; i < 4; ++i) {
						if (st == ST_OUTER) st = field_step<DBG>(sc, r, tally);
						if (__ballot(st == ST_OUTER) == 0ull) break;
					}
					lds[sb] = make_uint4(__float_as_uint(r.tx), __float_as_uint(r.ty), __float_as_uint(r.tz), r.p);
					reinterpret_cast<uint32_t*>(lds + sb + 256u)[3] = meta_after_walk(c1.w, r);
					if (DBG) reinterpret_cast<uint32_t*>(lds + sb + 1024u)[0] += tally.index_loads - loads_before;
					states = (states & ~(0xFFu << (8u * k))) | ((1u << st) << (8u * k));
				}
			}
		}
		if (DBG) {
			const unsigned long long dt = __builtin_amdgcn_s_memtime() - t_phase;
			if (phase == 0) cycA += dt; else if (phase == 1) cycB += dt; else cycC += dt;
		}
	}

	if (DBG && counters) {
		unsigned long long v[8] = {tally.index_loads, tally.brick_tests, tally.byte_tests, tally.voxel_steps,
								   tally.extend_rays, tally.shadow_rays, tally.requests, tally.paths};
		for (int k = 0; k < 8; ++k) {
			unsigned long long t = v[k];
			for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
			if (lane == 0 && t) atomicAdd(&counters->v[k], t);
		}
		if (lane == 0) {
			const unsigned long long st[8] = {runsA, lanesA, runsB, lanesB, runsC, lanesC, 0ull, 0ull};
			for (int k = 0; k < 8; ++k) atomicAdd(&counters->sched[k], st[k]);
			const unsigned long long t_end = __builtin_amdgcn_s_memtime();
			const unsigned long long cy[8] = {cycA, cycB, cycC, t_dry ? t_end - t_dry : 0ull, t_end - t_begin, runsJ, lanesJ, 1ull};
			for (int k = 0; k < 8; ++k) atomicAdd(&counters->cycles[k], cy[k]);
		}
	}
}

// ---- host-callable launchers (kernels.h)
int trace_k_blocks_per_cu(bool instrumented) {
	int n = 0;
	const hipError_t e = instrumented ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trace_paths_k<true>, 256, 0)
									  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trace_paths_k<false>, 256, 0);
#ifdef BM_K_MAX_BLOCKS // occupancy experiments (docs/HISTORY.md 5.5): fewer resident workgroups per CU than the LDS allows
	if (n > BM_K_MAX_BLOCKS) n = BM_K_MAX_BLOCKS;
#endif
	return e == hipSuccess && n > 0 ? n : 1;
}
size_t trace_k_scratch_bytes(bool instrumented, int resident_blocks) {
	return static_cast<size_t>(resident_blocks) * 256u * static_cast<size_t>(BM_K) * (instrumented ? 6u : 4u) * sizeof(uint4);
}
int trace_k_slots() { return BM_K; }

void launch_trace_k(const DeviceScene& sc, const FrameConstants& fc, const FrameConstants* fc_dev, float* accum, uint32_t* dbg, DeviceCounters* counters,
					uint32_t* work_counter, bool instrumented, int resident_blocks, void* scratch, hipStream_t stream) {
	const long long chunks = static_cast<long long>(fc.tiles_x) * fc.tiles_y * 16;
	if (chunks <= 0) return;
	long long blocks = (chunks + 15) / 16;
	if (blocks > resident_blocks) blocks = resident_blocks;
	if (instrumented)
		hipLaunchKernelGGL(trace_paths_k<true>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, sc, fc_dev, reinterpret_cast<float4*>(accum), dbg, counters,
						   work_counter, reinterpret_cast<uint4*>(scratch));
	else
		hipLaunchKernelGGL(trace_paths_k<false>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, sc, fc_dev, reinterpret_cast<float4*>(accum), nullptr, nullptr,
						   work_counter, reinterpret_cast<uint4*>(scratch));
}

} // namespace bm
