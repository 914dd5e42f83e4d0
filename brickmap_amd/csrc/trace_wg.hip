// trace_wg.hip -- the fused per-pixel path tracer of trace.hip with WORKGROUP-level work balancing.
//
// trace.hip lets every wave schedule its own 64 paths: a phase runs with whatever fraction of the wave's lanes wants
// it (32 of 64 lanes in a brick-grid move round, ~20 in the candidate / shade rounds).  Here the four waves of a
// workgroup pool their 256 paths, as the wavefront kernels do (wavefront.hip, wf_trace_wg): every BM_TWG_PERIOD phases
// the HOT state of every path (what the brick-grid move needs: 16 dwords) goes through LDS, stably partitioned by the
// phase the path needs next (moves, candidates, shade, connect, idle), and is picked up by thread `position`, so that
// waves become homogeneous.  The COLD state of a path (pixel, sample, bounce, surface point, colours, the ray's origin
// and direction: what only the shade / connect / candidate phases read) lives in a 128-byte record in global memory
// (one per slot of the workgroup, L2-resident) that is addressed by the slot id travelling with the hot state.
// Per-path arithmetic, RNG streams and the per-pixel accumulation order are those of trace.hip: results are identical.
#include "traverse.h"

#include "kernels.h"

namespace bm {

namespace {
enum : int { P_GEN = 0, P_EXT_DONE = 1, P_SHD_DONE = 2 };
enum : int { ST_IDLE = 4, ST_CONN = 5 };
} // namespace

#ifndef BM_TWG_PERIOD
#define BM_TWG_PERIOD 4
#endif
#ifndef BM_TWG_STEPS
#define BM_TWG_STEPS 10
#endif
#ifndef BM_TWG_MIN
#define BM_TWG_MIN 8 // a wave runs an expensive phase (candidates, shade, connect) once this many of its lanes want it
#endif
#ifndef BM_WORK_COUNTERS
#define BM_WORK_COUNTERS 8
#endif

// cold per-path record: 32 dwords
struct PathRec {
	float o[3], tminn;    // ray origin in brick units / entry distance (ray_setup), read by candidate resolution
	float d[3];           // ray direction
	uint32_t flags;       // bounces | shadow << 8 | terminated << 9 | pstate << 10
	uint32_t p, local_pixel, xy;
	int s;
	float hitp[3], pn[3], scolor[3], bdir[3];
	uint32_t dbg[8];      // d0 d1 d2 d3 hseg hsh next|nsh<<16 index loads of the pixel (instrumented variant)
};
static_assert(sizeof(PathRec) == 128, "one cache line per path");

template <bool DBG>
__global__ __launch_bounds__(256, DBG ? 2 : 4) void trace_paths_wg(const DeviceScene sc, const FrameConstants* __restrict__ fcp, float4* __restrict__ accum,
																	 uint32_t* __restrict__ dbg, DeviceCounters* __restrict__ counters,
																	 uint32_t* __restrict__ work_counter, PathRec* __restrict__ scratch) {
	const FrameConstants& fc = *fcp;
	constexpr int kHot = DBG ? 19 : 16;
	__shared__ unsigned long long lds_brick[8 * 256];
	__shared__ uint32_t pool[kHot][256];
	__shared__ uint32_t wave_cnt[4][4]; // [class: moves, candidates, shade, connect][wave]
	__shared__ uint32_t s_base, s_want, s_counter, s_tickets, s_work_left;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const uint32_t W = static_cast<uint32_t>(fc.width), H = static_cast<uint32_t>(fc.height);
	const uint32_t total_chunks = static_cast<uint32_t>(fc.tiles_x) * static_cast<uint32_t>(fc.tiles_y) * 16u;
	PathRec* const recs = scratch + static_cast<size_t>(blockIdx.x) * 256;

	RayState r;
	r.hit = false;
	r.n = mk(0.f, 0.f, 0.f);
	r.tx = r.ty = r.tz = r.dx = r.dy = r.dz = 0.f;
	r.p = 0u; r.sx = r.stepy = r.stepz = 0; r.last_step = 0;
	r.distance = 0.f;
	Tally tally;
	HitInfo info;
	uint32_t ray_loads = 0; // index loads of the ray in flight (instrumented variant)
	int state = ST_IDLE;
	uint32_t slot = static_cast<uint32_t>(tid);
	bool shadow = false;

	// thread 0: chunk tickets of the workgroup (same dealing as trace.hip: 8 interleaved counters, 4x4-pixel chunks)
	bool work_left = true;
	constexpr uint32_t kCounters = BM_WORK_COUNTERS, kCounterStride = 32;
	int my_counter = static_cast<int>(blockIdx.x % kCounters);
	int counters_done = 0;
	long long rounds_left = (static_cast<long long>(total_chunks) + 64) * (static_cast<long long>(fc.spp) + 1) * (fc.max_bounces + 2) *
							(2ll * sc.cells + sc.cells_height + 64);
	uint32_t runsA = 0, lanesA = 0, runsB = 0, lanesB = 0, runsC = 0, lanesC = 0, runsD = 0, lanesD = 0;

	for (;;) {
		// ---- stable partition of the workgroup's 256 slots by the phase each path needs next
		const bool c0 = state == ST_OUTER, c1 = state == ST_CAND, c2 = state == ST_NEED, c3 = state == ST_CONN;
		const unsigned long long b0 = __ballot(c0), b1 = __ballot(c1), b2 = __ballot(c2), b3 = __ballot(c3);
		if (lane == 0) {
			wave_cnt[0][wave] = static_cast<uint32_t>(__popcll(b0)); wave_cnt[1][wave] = static_cast<uint32_t>(__popcll(b1));
			wave_cnt[2][wave] = static_cast<uint32_t>(__popcll(b2)); wave_cnt[3][wave] = static_cast<uint32_t>(__popcll(b3));
		}
		__syncthreads();
		uint32_t n[4] = {0, 0, 0, 0}, pre[4] = {0, 0, 0, 0};
		uint32_t idle_before = 0; // idle lanes in the waves before this one
		for (int w = 0; w < 4; ++w) {
			uint32_t busy = 0;
			for (int c = 0; c < 4; ++c) {
				const uint32_t v = wave_cnt[c][w];
				if (w < wave) pre[c] += v;
				n[c] += v;
				busy += v;
			}
			if (w < wave) idle_before += 64u - busy;
		}
		const uint32_t live = n[0] + n[1] + n[2] + n[3];
		const unsigned long long below = (1ull << lane) - 1ull;
		uint32_t dest;
		if (c0) dest = pre[0] + static_cast<uint32_t>(__popcll(b0 & below));
		else if (c1) dest = n[0] + pre[1] + static_cast<uint32_t>(__popcll(b1 & below));
		else if (c2) dest = n[0] + n[1] + pre[2] + static_cast<uint32_t>(__popcll(b2 & below));
		else if (c3) dest = n[0] + n[1] + n[2] + pre[3] + static_cast<uint32_t>(__popcll(b3 & below));
		else dest = live + idle_before + static_cast<uint32_t>(__popcll(~(b0 | b1 | b2 | b3) & below));
		pool[15][dest] = slot | (shadow ? 0x100u : 0u) | (r.hit ? 0x200u : 0u);
		if (c0 || c1 || c2 || c3) {
			pool[0][dest] = __float_as_uint(r.tx); pool[1][dest] = __float_as_uint(r.ty); pool[2][dest] = __float_as_uint(r.tz);
			pool[3][dest] = __float_as_uint(r.dx); pool[4][dest] = __float_as_uint(r.dy); pool[5][dest] = __float_as_uint(r.dz);
			pool[6][dest] = r.p; pool[7][dest] = static_cast<uint32_t>(r.sx); pool[8][dest] = static_cast<uint32_t>(r.stepy);
			pool[9][dest] = static_cast<uint32_t>(r.stepz);
			pool[10][dest] = __float_as_uint(r.n.x); pool[11][dest] = __float_as_uint(r.n.y); pool[12][dest] = __float_as_uint(r.n.z);
			pool[13][dest] = static_cast<uint32_t>(r.last_step); pool[14][dest] = __float_as_uint(r.distance);
			if (DBG) {
				pool[16][dest] = static_cast<uint32_t>(info.level) | (static_cast<uint32_t>(info.sub_id) << 8);
				pool[17][dest] = static_cast<uint32_t>(info.brick_id);
				pool[18][dest] = ray_loads;
			}
		}
		if (tid == 0) { // chunk tickets for the idle lanes (positions live .. 255), 16 pixels (one 4x4 chunk) at a time
			const uint32_t idle = 256u - live;
			uint32_t want = 0;
			if (work_left && idle >= 16u) {
				want = idle >> 4;
				const uint32_t base = atomicAdd(work_counter + my_counter * kCounterStride, want);
				const uint32_t total_groups = (total_chunks + 3u) >> 2;
				const uint32_t my_groups = total_groups > static_cast<uint32_t>(my_counter)
											   ? (total_groups - static_cast<uint32_t>(my_counter) + kCounters - 1u) / kCounters : 0u;
				const uint32_t my_tickets = my_groups * 4u;
				s_base = base; s_counter = static_cast<uint32_t>(my_counter); s_tickets = my_tickets;
				if (base + want >= my_tickets) {
					my_counter = (my_counter + 1) % static_cast<int>(kCounters);
					if (++counters_done >= static_cast<int>(kCounters)) work_left = false;
				}
			}
			s_want = want;
			s_work_left = work_left ? 1u : 0u;
		}
		__syncthreads();
		const uint32_t want = s_want;
		if (live == 0 && want == 0 && s_work_left == 0u) break;
		if (--rounds_left < 0) break;
		{
			const uint32_t w = pool[15][tid];
			slot = w & 0xFFu;
			shadow = (w & 0x100u) != 0u;
			r.hit = (w & 0x200u) != 0u;
		}
		const uint32_t pos = static_cast<uint32_t>(tid);
		if (pos < live) {
			r.tx = __uint_as_float(pool[0][tid]); r.ty = __uint_as_float(pool[1][tid]); r.tz = __uint_as_float(pool[2][tid]);
			r.dx = __uint_as_float(pool[3][tid]); r.dy = __uint_as_float(pool[4][tid]); r.dz = __uint_as_float(pool[5][tid]);
			r.p = pool[6][tid]; r.sx = static_cast<int>(pool[7][tid]); r.stepy = static_cast<int>(pool[8][tid]);
			r.stepz = static_cast<int>(pool[9][tid]);
			r.n = mk(__uint_as_float(pool[10][tid]), __uint_as_float(pool[11][tid]), __uint_as_float(pool[12][tid]));
			r.last_step = static_cast<int>(pool[13][tid]); r.distance = __uint_as_float(pool[14][tid]);
			if (DBG) {
				const uint32_t li = pool[16][tid];
				info.level = static_cast<int>(li & 0xFFu); info.sub_id = static_cast<int>(li >> 8);
				info.brick_id = static_cast<int>(pool[17][tid]);
				ray_loads = pool[18][tid];
			}
			state = pos < n[0] ? ST_OUTER : (pos < n[0] + n[1] ? ST_CAND : (pos < n[0] + n[1] + n[2] ? ST_NEED : ST_CONN));
		} else {
			state = ST_IDLE;
			// ---- refill: hand a pixel to an idle slot
			const uint32_t k = pos - live;
			if (k < want * 16u) {
				const uint32_t ticket = s_base + (k >> 4), my_tickets = s_tickets, counter_now = s_counter;
				const uint32_t chunk = ((ticket >> 2) * kCounters + counter_now) * 4u + (ticket & 3u);
				if (ticket < my_tickets && chunk < total_chunks) {
					const uint32_t tile = chunk >> 4, kk = chunk & 15u;
					const int tile_x = static_cast<int>(tile % static_cast<uint32_t>(fc.tiles_x));
					const int tile_y = static_cast<int>(tile / static_cast<uint32_t>(fc.tiles_x));
					const int cx = static_cast<int>((kk & 1u) | ((kk >> 1) & 2u)), cy = static_cast<int>(((kk >> 1) & 1u) | ((kk >> 2) & 2u));
					const int x = tile_x * 16 + cx * 4 + static_cast<int>(k & 3u);
					const int ly = tile_y * 16 + cy * 4 + static_cast<int>((k >> 2) & 3u);
					const int y = ((ly / fc.band_rows) * fc.shard_count + fc.shard_rank) * fc.band_rows + ly % fc.band_rows;
					if (x < fc.width && ly < fc.local_rows && y < fc.height) {
						PathRec& c = recs[slot];
						c.p = static_cast<uint32_t>(y) * W + static_cast<uint32_t>(x);
						c.local_pixel = static_cast<uint32_t>(ly) * W + static_cast<uint32_t>(x);
						c.xy = static_cast<uint32_t>(x) | (static_cast<uint32_t>(y) << 16);
						c.s = 0;
						c.flags = static_cast<uint32_t>(P_GEN) << 10;
						if (DBG) {
							c.dbg[0] = 0; c.dbg[1] = 0; c.dbg[2] = 0xFFFFFFFFu; c.dbg[3] = 0; c.dbg[4] = 2166136261u; c.dbg[5] = 2166136261u;
							c.dbg[6] = 0; c.dbg[7] = 0;
						}
						state = ST_NEED;
						shadow = false;
						r.hit = false;
					}
				}
			}
		}

#pragma unroll 1
		for (int ph = 0; ph < BM_TWG_PERIOD; ++ph) {
			const int nA = __popcll(__ballot(state == ST_OUTER)), nB = __popcll(__ballot(state == ST_CAND));
			const int nC = __popcll(__ballot(state == ST_NEED)), nD = __popcll(__ballot(state == ST_CONN));
			if (nA + nB + nC + nD == 0) break;
			int phase; // 0 moves, 1 candidates, 2 shade, 3 connect: an expensive phase once BM_TWG_MIN lanes want it, else the largest group
			if (nC >= BM_TWG_MIN && nC >= nB && nC >= nD) phase = 2;
			else if (nB >= BM_TWG_MIN && nB >= nD) phase = 1;
			else if (nD >= BM_TWG_MIN) phase = 3;
			else if (nA > 0) phase = 0;
			else phase = (nC >= nB && nC >= nD) ? 2 : (nB >= nD ? 1 : 3);

			if (phase == 2) {
				if (DBG && lane == 0) { runsC++; lanesC += nC; }
				// ================= shade the finished extend ray / generate the next primary ray, then set the new ray up
				if (state == ST_NEED) {
					PathRec& c = recs[slot];
					const uint4 q0 = reinterpret_cast<const uint4*>(&c)[1], q1 = reinterpret_cast<const uint4*>(&c)[2];
					const uint4 q2 = reinterpret_cast<const uint4*>(&c)[3], q3 = reinterpret_cast<const uint4*>(&c)[4], q4 = reinterpret_cast<const uint4*>(&c)[5];
					const f3 ray_d = mk(__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z));
					const uint32_t flags = q0.w;
					int bounces = static_cast<int>(flags & 0xFFu);
					bool terminated = (flags & 0x200u) != 0u;
					int pstate = static_cast<int>((flags >> 10) & 3u);
					const uint32_t p = q1.x, local_pixel = q1.y, xy = q1.z;
					int s = static_cast<int>(q1.w);
					f3 hitp = mk(__uint_as_float(q2.x), __uint_as_float(q2.y), __uint_as_float(q2.z));
					f3 pn = mk(__uint_as_float(q2.w), __uint_as_float(q3.x), __uint_as_float(q3.y));
					f3 scolor = mk(__uint_as_float(q3.z), __uint_as_float(q3.w), __uint_as_float(q4.x));
					f3 bdir = mk(__uint_as_float(q4.y), __uint_as_float(q4.z), __uint_as_float(q4.w));
					uint32_t d0 = 0, d1 = 0, d2 = 0, d3 = 0, hseg = 0, hsh = 0, nxt = 0, nsh = 0, loads = 0;
					if (DBG) {
						d0 = c.dbg[0]; d1 = c.dbg[1]; d2 = c.dbg[2]; d3 = c.dbg[3]; hseg = c.dbg[4]; hsh = c.dbg[5];
						nxt = c.dbg[6] & 0xFFFFu; nsh = c.dbg[6] >> 16; loads = c.dbg[7];
					}
					bool need_setup = false;
					f3 ro = mk(0.f, 0.f, 0.f), rd = mk(0.f, 0.f, 0.f);
					if (pstate == P_EXT_DONE) {
						const bool is_hit = r.hit;
						pn = r.n;
						if (DBG) {
							tally.extend_rays++;
							nxt++;
							loads += ray_loads;
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
						f3 view = ray_d;
						f3 miss_color = mk(0.f, 0.f, 0.f);
						float sunLight = 0.f;
						bool cast = false;
						if (is_hit && !primary_only) {
							const uint32_t frame = fc.base_frame + static_cast<uint32_t>(bounces);
							const uint32_t slot_index = p + static_cast<uint32_t>(fc.sample_base + s) * W * H;
							uint32_t sseed = (frame * p * 147565741u) * 720898027u * slot_index;
							hitp = hitp + ray_d * r.distance;
							hitp = hitp + pn * 2.f * kEpsilon;
							view = cone_sample(fc, sseed);
							sunLight = dot(pn, view);
							cast = sunLight > 0.f;
							terminated = !(bounces < fc.max_bounces);
							if (terminated) accum[local_pixel].w += 1.f;
							else bdir = bounce_direction(pn, sseed);
							if (!cast) {
								if (terminated) { s++; pstate = P_GEN; }
								else { bounces++; ro = hitp; rd = bdir; r.n = pn; shadow = false; need_setup = true; }
							}
						}
						if (!is_hit || cast) {
							const SkyView sv = sky_view(fc, view);
							if (cast) {
								scolor = (sun_from_view(fc, sv) * sunLight) * 1E-5f;
								ro = hitp; rd = view;
								shadow = true;
								need_setup = true;
							} else {
								f3 col;
								if (bounces == 0) col = fc.sun_angular_cos == 1.0f ? mk(1.0f, 0.0f, 0.0f) : sunsky_from_view(fc, sv);
								else col = sky_from_view(fc, sv);
								miss_color = col;
							}
						}
						if (!is_hit || primary_only) {
							float4 a = accum[local_pixel];
							a.x += miss_color.x; a.y += miss_color.y; a.z += miss_color.z;
							a.w += 1.f;
							accum[local_pixel] = a;
							s++;
							pstate = P_GEN;
						}
					}
					if (pstate == P_GEN) {
						if (s >= fc.spp) {
							if (DBG && dbg) {
								uint32_t* dd = dbg + static_cast<size_t>(local_pixel) * 8;
								dd[0] = d0; dd[1] = d1; dd[2] = d2; dd[3] = d3; dd[4] = hseg; dd[5] = hsh; dd[6] = nxt | (nsh << 16); dd[7] = loads;
							}
							state = ST_IDLE;
						} else {
							const uint32_t slot_index = p + static_cast<uint32_t>(fc.sample_base + s) * W * H;
							const uint32_t seed = (fc.base_frame * 147565741u) * 720898027u * slot_index;
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
					if (need_setup) {
						if (shadow) r.n = mk(0.f, 0.f, 0.f);
						pstate = shadow ? P_SHD_DONE : P_EXT_DONE;
						if (DBG) ray_loads = 0;
						const uint32_t before = tally.index_loads;
						const int st = ray_setup<DBG>(sc, ro, rd, r, tally);
						if (DBG) ray_loads = tally.index_loads - before;
						state = (st == ST_NEED && shadow) ? ST_CONN : st;
						reinterpret_cast<uint4*>(&c)[0] = make_uint4(__float_as_uint(r.o.x), __float_as_uint(r.o.y), __float_as_uint(r.o.z), __float_as_uint(r.tminn));
					}
					if (state != ST_IDLE) {
						const uint32_t nf = static_cast<uint32_t>(bounces) | (shadow ? 0x100u : 0u) | (terminated ? 0x200u : 0u) | (static_cast<uint32_t>(pstate) << 10);
						reinterpret_cast<uint4*>(&c)[1] = make_uint4(__float_as_uint(rd.x), __float_as_uint(rd.y), __float_as_uint(rd.z), nf);
						c.s = s;
						reinterpret_cast<uint4*>(&c)[3] = make_uint4(__float_as_uint(hitp.x), __float_as_uint(hitp.y), __float_as_uint(hitp.z), __float_as_uint(pn.x));
						reinterpret_cast<uint4*>(&c)[4] = make_uint4(__float_as_uint(pn.y), __float_as_uint(pn.z), __float_as_uint(scolor.x), __float_as_uint(scolor.y));
						reinterpret_cast<uint4*>(&c)[5] = make_uint4(__float_as_uint(scolor.z), __float_as_uint(bdir.x), __float_as_uint(bdir.y), __float_as_uint(bdir.z));
						if (DBG) { c.dbg[0] = d0; c.dbg[1] = d1; c.dbg[2] = d2; c.dbg[3] = d3; c.dbg[4] = hseg; c.dbg[5] = hsh; c.dbg[6] = nxt | (nsh << 16); c.dbg[7] = loads; }
					}
				}
			} else if (phase == 3) {
				if (DBG && lane == 0) { runsD++; lanesD += nD; }
				// ================= connect (kernel.cu:328-346), then the stored bounce ray is set up
				if (state == ST_CONN) {
					PathRec& c = recs[slot];
					const uint4 q0 = reinterpret_cast<const uint4*>(&c)[1], q1 = reinterpret_cast<const uint4*>(&c)[2];
					const uint4 q2 = reinterpret_cast<const uint4*>(&c)[3], q3 = reinterpret_cast<const uint4*>(&c)[4], q4 = reinterpret_cast<const uint4*>(&c)[5];
					const uint32_t flags = q0.w;
					int bounces = static_cast<int>(flags & 0xFFu);
					const bool terminated = (flags & 0x200u) != 0u;
					const uint32_t local_pixel = q1.y;
					int s = static_cast<int>(q1.w);
					const f3 hitp = mk(__uint_as_float(q2.x), __uint_as_float(q2.y), __uint_as_float(q2.z));
					const f3 pn = mk(__uint_as_float(q2.w), __uint_as_float(q3.x), __uint_as_float(q3.y));
					const f3 scolor = mk(__uint_as_float(q3.z), __uint_as_float(q3.w), __uint_as_float(q4.x));
					const f3 bdir = mk(__uint_as_float(q4.y), __uint_as_float(q4.z), __uint_as_float(q4.w));
					const bool occluded = r.hit;
					if (DBG) {
						tally.shadow_rays++;
						uint32_t hsh = c.dbg[5], nsh = c.dbg[6] >> 16;
						nsh++;
						hsh = hmix(hsh, static_cast<uint32_t>(occluded));
						if (occluded) {
							hsh = hmix(hsh, static_cast<uint32_t>(info.brick_id));
							hsh = hmix(hsh, static_cast<uint32_t>(info.sub_id) | (static_cast<uint32_t>(info.level) << 12));
						}
						c.dbg[5] = hsh; c.dbg[6] = (c.dbg[6] & 0xFFFFu) | (nsh << 16); c.dbg[7] += ray_loads;
					}
					if (!occluded) {
						float4 a = accum[local_pixel];
						a.x += scolor.x; a.y += scolor.y; a.z += scolor.z;
						accum[local_pixel] = a;
					}
					int pstate;
					f3 rd = bdir;
					if (terminated) {
						s++;
						pstate = P_GEN;
						state = ST_NEED;
						shadow = false;
						r.hit = false;
						c.s = s;
					} else {
						bounces++;
						r.n = pn;
						shadow = false;
						pstate = P_EXT_DONE;
						const uint32_t before = tally.index_loads;
						state = ray_setup<DBG>(sc, hitp, bdir, r, tally);
						if (DBG) ray_loads = tally.index_loads - before;
						reinterpret_cast<uint4*>(&c)[0] = make_uint4(__float_as_uint(r.o.x), __float_as_uint(r.o.y), __float_as_uint(r.o.z), __float_as_uint(r.tminn));
					}
					const uint32_t nf = static_cast<uint32_t>(bounces) | (terminated ? 0x200u : 0u) | (static_cast<uint32_t>(pstate) << 10);
					reinterpret_cast<uint4*>(&c)[1] = make_uint4(__float_as_uint(rd.x), __float_as_uint(rd.y), __float_as_uint(rd.z), nf);
				}
			} else if (phase == 1) {
				if (DBG && lane == 0) { runsB++; lanesB += nB; }
				// ================= resolve non-empty cells
				if (state == ST_CAND) {
					const PathRec& c = recs[slot];
					const uint4 q = reinterpret_cast<const uint4*>(&c)[0], q0 = reinterpret_cast<const uint4*>(&c)[1];
					r.o = mk(__uint_as_float(q.x), __uint_as_float(q.y), __uint_as_float(q.z));
					r.tminn = __uint_as_float(q.w);
					r.d = mk(__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z));
					load_block(sc, r);
					const int st = process_candidate<DBG>(sc, fc.campos, r, info, tally, lds_brick);
					state = (st == ST_NEED && shadow) ? ST_CONN : st;
				}
			} else {
				// ================= brick-grid moves
#pragma unroll 1
				for (int k = 0; k < BM_TWG_STEPS; ++k) {
					if (DBG) { const int nn = __popcll(__ballot(state == ST_OUTER)); if (lane == 0) { runsA++; lanesA += nn; } }
					if (state == ST_OUTER) {
						const uint32_t before = tally.index_loads;
						const int st = outer_step<DBG>(sc, r, tally);
						if (DBG) ray_loads += tally.index_loads - before;
						state = (st == ST_NEED && shadow) ? ST_CONN : st;
					}
				}
			}
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
			const unsigned long long st8[8] = {runsA, lanesA, runsB, lanesB, runsC, lanesC, runsD, lanesD};
			for (int k = 0; k < 8; ++k) atomicAdd(&counters->sched[k], st8[k]);
		}
	}
}

size_t trace_wg_scratch_bytes(int resident_blocks) { return static_cast<size_t>(resident_blocks) * 256 * sizeof(PathRec); }

int trace_wg_blocks_per_cu(bool instrumented) {
	int n = 0;
	const hipError_t e = instrumented ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trace_paths_wg<true>, 256, 0)
									  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trace_paths_wg<false>, 256, 0);
	return e == hipSuccess && n > 0 ? n : 1;
}

void launch_trace_wg(const DeviceScene& sc, const FrameConstants& fc, const FrameConstants* fc_dev, float* accum, uint32_t* dbg, DeviceCounters* counters,
					 uint32_t* work_counter, bool instrumented, int resident_blocks, void* scratch, hipStream_t stream) {
	const long long chunks = static_cast<long long>(fc.tiles_x) * fc.tiles_y * 16;
	if (chunks <= 0) return;
	long long blocks = (chunks + 15) / 16;
	if (blocks > resident_blocks) blocks = resident_blocks;
	PathRec* recs = reinterpret_cast<PathRec*>(scratch);
	if (instrumented)
		hipLaunchKernelGGL(trace_paths_wg<true>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, sc, fc_dev, reinterpret_cast<float4*>(accum), dbg,
						   counters, work_counter, recs);
	else
		hipLaunchKernelGGL(trace_paths_wg<false>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, sc, fc_dev, reinterpret_cast<float4*>(accum), nullptr,
						   nullptr, work_counter, recs);
}

} // namespace bm
