// wavefront.hip -- the reference's own schedule on gfx950: queue-based primary_rays / extend / shade / connect
// (src/kernel.cu:154-346), one launch_kernels call (kernel.cu:366-439) = one segment of every path in flight.
//
// What differs from the reference's kernels is only WHERE work runs, never what is computed per ray:
//   * extend and connect are persistent wave64 kernels.  A wave owns a private range of queue slots (one atomic per
//     BM_WF_GRAB rays) and refills finished lanes from it, so lanes keep walking while their neighbours' rays end;
//     inside the wave the brick-grid moves (phase A) and the index word / bitmask walks (phase B) of different rays
//     are batched exactly as in trace.hip's scheduler.
//   * shade compacts with a prefix sum instead of atomicAdd tickets (kernel.cu:276,299): survivors and shadow rays
//     land in the order of their source slot.  The reference's order is whatever its atomics produce; slot order
//     is the one a sequential run of the reference gives, which is what the CPU oracle (mode A) computes, so the
//     queues are bit-identical to the oracle's.  Because the RNG seed of shade() contains the slot index, this
//     also makes the image reproducible run to run (the reference's is not).
//   * the frame buffer is accumulated with hardware float atomics like the reference (kernel.cu:319-322,341-343):
//     the order of two contributions to one pixel inside one kernel is not fixed, radiance agrees to ~1e-6.
#include "traverse.h"

#include "kernels.h"

namespace bm {

#ifndef BM_WF_GRAB
#define BM_WF_GRAB 64 // queue slots a wave reserves per atomic (32: the ticket atomic becomes the bottleneck; 128: too few
					  // ranges per wave -- 2 Mi slots over ~6000 resident waves -- and the kernel's tail grows)
#endif
#ifndef BM_WF_REFILL
#define BM_WF_REFILL 32 // idle lanes that trigger a refill (ray loads + ray_setup run for that many lanes at once; 16 and 48 are ~5 % slower)
#endif
#ifndef BM_WF_QUORUM_DIV
#define BM_WF_QUORUM_DIV 4
#endif
#ifndef BM_WF_WG
#define BM_WF_WG 1 // 1: workgroup-balanced kernel (wf_trace_wg), 0: wave-private rays (wf_trace)
#endif
#if BM_WF_WG
#define BM_WF_TRACE wf_trace_wg
#else
#define BM_WF_TRACE wf_trace
#endif
#ifndef BM_WF_STEPS
#define BM_WF_STEPS 4
#endif


namespace {
constexpr float kVeryFar = 1e20f; // kernel.cu:12
enum : int { WF_DEAD = 4 };

__device__ __forceinline__ void atomic_add_rgb(float4* px, f3 c) {
	float* p = reinterpret_cast<float*>(px);
	unsafeAtomicAdd(p + 0, c.x);
	unsafeAtomicAdd(p + 1, c.y);
	unsafeAtomicAdd(p + 2, c.z);
}
} // namespace

// primary_rays (kernel.cu:154-223): fill the work queue behind the survivors of the previous frame.
__global__ __launch_bounds__(256) void wf_primary(const WfState* __restrict__ st, WfRay* __restrict__ work, const FrameConstants* __restrict__ fcp,
												 uint32_t queue_size) {
	const FrameConstants& fc = *fcp;
	const uint32_t index = blockIdx.x * 256u + threadIdx.x;
	const uint32_t ray_index_buffer = index + st->primary_ray_cnt;
	if (index >= queue_size || ray_index_buffer > queue_size - 1) return;
	const uint32_t W = static_cast<uint32_t>(fc.width), H = static_cast<uint32_t>(fc.height);
	const uint32_t seed = (fc.base_frame * 147565741u) * 720898027u * index;
	const uint32_t x = (st->start_position + index) % W;
	const uint32_t y = ((st->start_position + index) / W) % H;
	f3 o, d;
	primary_ray(fc, seed, x, y, o, d);
	float4* rec = reinterpret_cast<float4*>(work + ray_index_buffer);
	rec[0] = make_float4(o.x, o.y, o.z, d.x);
	rec[1] = make_float4(d.y, d.z, 1.f, 1.f);
	rec[2] = make_float4(1.f, 0.f, 0.f, 0.f);
	rec[3] = make_float4(0.f, __int_as_float(0), __int_as_float(0), __uint_as_float(y * W + x));
}

// set_wavefront_globals (kernel.cu:122-139)
__global__ void wf_globals(WfState* st, uint32_t queue_size, uint32_t pixels) {
	const uint32_t progress_last_frame = queue_size - st->primary_ray_cnt;
	st->generated = progress_last_frame;
	st->start_position = (st->start_position + progress_last_frame) % pixels;
	st->shadow_ray_cnt = 0;
	st->primary_ray_cnt = 0;
	st->extend_ticket[0] = 0;
	st->connect_ticket[0] = 0;
}

// extend (kernel.cu:226-238) when !CONNECT: intersect every ray of the work queue, write distance + normal back.
// connect (kernel.cu:328-346) when CONNECT: intersect every shadow ray, add its colour to the pixel if unoccluded.
template <bool CONNECT, bool DBG>
__global__ __launch_bounds__(256, DBG ? 4 : 6) void wf_trace(const DeviceScene sc, const FrameConstants* __restrict__ fcp, WfState* __restrict__ st,
											   WfRay* __restrict__ work, const WfShadow* __restrict__ shadow, float4* __restrict__ accum,
											   DeviceCounters* __restrict__ counters, uint32_t queue_size, float4* __restrict__ /*cold: wf_trace_wg only*/) {
	const FrameConstants& fc = *fcp;
	__shared__ unsigned long long lds_brick[8 * 256];
	const int lane = threadIdx.x & 63;
	const uint32_t total = CONNECT ? st->shadow_ray_cnt : queue_size;
	uint32_t* ticket = CONNECT ? st->connect_ticket : st->extend_ticket;

	RayState r;
	r.hit = false;
	r.n = mk(0.f, 0.f, 0.f);
	Tally tally;
	HitInfo info;
	int state = ST_NEED;
	bool have = false; // the lane holds a ray whose result has not been written yet
	uint32_t idx = 0;
	uint32_t cur = 0, end = 0; // the wave's private slot range
	bool work_left = true;
	long long rounds_left = (static_cast<long long>(total) + 64) * (2ll * sc.cells + sc.cells_height + 64); // hang guard only
	uint32_t runsA = 0, lanesA = 0, runsB = 0, lanesB = 0, runsR = 0, lanesR = 0; // wave-uniform scheduler statistics (DBG)

	for (;;) {
		// ---- retire: write the result of every lane whose ray has ended
		if (state == ST_NEED && have) {
			have = false;
			if (CONNECT) {
				if (DBG) tally.shadow_rays++;
				if (!r.hit) {
					const float* s = reinterpret_cast<const float*>(shadow + idx);
					const uint32_t pixel = __float_as_uint(s[9]);
					atomic_add_rgb(accum + pixel, mk(s[6], s[7], s[8]));
				}
			} else {
				if (DBG) tally.extend_rays++;
				float* rec = reinterpret_cast<float*>(work + idx);
				rec[9] = r.n.x; rec[10] = r.n.y; rec[11] = r.n.z; // written on a miss too: the walk clobbers RayQueue::normal in place
				rec[12] = r.hit ? r.distance : kVeryFar;
			}
		}
		const unsigned long long need = __ballot(state == ST_NEED);
		const int nN = __popcll(need);
		const int nJ = __popcll(__ballot(state == ST_JUMP));
		const int nA = __popcll(__ballot(state == ST_OUTER)) + nJ; // walking lanes: cell by cell (ST_OUTER) or cube by cube (ST_JUMP)
		const int nB = __popcll(__ballot(state == ST_CAND));
		const bool more = work_left || cur < end;
		if (--rounds_left < 0) break;
		// ---- refill idle lanes from the wave's private slot range
		if (more && nN > 0 && (nN >= BM_WF_REFILL || nA + nB == 0)) {
			if (cur == end) {
				uint32_t base = 0;
				if (lane == 0) base = atomicAdd(ticket, static_cast<uint32_t>(BM_WF_GRAB));
				base = __builtin_amdgcn_readfirstlane(base);
				if (base >= total) {
					work_left = false;
				} else {
					cur = base;
					end = total - base < static_cast<uint32_t>(BM_WF_GRAB) ? total : base + static_cast<uint32_t>(BM_WF_GRAB);
				}
			}
			const uint32_t avail = end - cur;
			const uint32_t take = static_cast<uint32_t>(nN) < avail ? static_cast<uint32_t>(nN) : avail;
			if (DBG) { runsR++; lanesR += take; }
			if (take > 0) {
				const uint32_t rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(need >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(need), 0u));
				if (state == ST_NEED && rank < take) {
					idx = cur + rank;
					f3 o, d;
					if (CONNECT) {
						const float2* s = reinterpret_cast<const float2*>(shadow + idx);
						const float2 a = s[0], b = s[1], c = s[2];
						o = mk(a.x, a.y, b.x);
						d = mk(b.y, c.x, c.y);
						r.n = mk(0.f, 0.f, 0.f); // connect passes a zeroed normal (kernel.cu:338)
					} else {
						const float4* q = reinterpret_cast<const float4*>(work + idx);
						const float4 a = q[0], b = q[1], c = q[2];
						o = mk(a.x, a.y, a.z);
						d = mk(a.w, b.x, b.y);
						r.n = mk(c.y, c.z, c.w);
					}
					have = true;
					state = ray_setup<DBG>(sc, o, d, r, tally);
				}
				cur += take;
			}
			continue;
		}
		if (nA + nB == 0) {
			if (!more) break;
			continue;
		}
		const int live = nA + nB;
		if (nB >= (live + BM_WF_QUORUM_DIV - 1) / BM_WF_QUORUM_DIV || nA == 0) {
			// ---- phase B: resolve non-empty cells (index word, LoD / 8^3 bitmask walk, streaming request)
			if (DBG) { runsB++; lanesB += nB; }
			if (state == ST_CAND) state = process_candidate<DBG>(sc, fc.campos, r, info, tally, lds_brick);
		} else {
			// ---- phase A: brick-grid walk (trace.hip phase A): a jump pass for every walking lane when enough of them have
			// an empty cube ahead, single moves otherwise
			state = walk_round<DBG, BM_WF_STEPS>(sc, r, state, nJ, nA - nJ, tally, runsA, lanesA);
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
		if (lane == 0) { // scheduler statistics: A runs / lanes, B runs / lanes, refills / rays handed out
			const unsigned long long st8[8] = {runsA, lanesA, runsB, lanesB, runsR, lanesR, 0ull, 0ull};
			for (int k = 0; k < 8; ++k) atomicAdd(&counters->sched[k], st8[k]);
		}
	}
}

// ---- workgroup-balanced variant of wf_trace
// In wf_trace a wave owns its 64 rays, so a phase runs with whatever fraction of them wants it (~53 % of the lanes in
// a move round of extend, ~36 % in a candidate round).  Here the four waves of a workgroup pool their 256 rays: once
// per round every live ray goes through LDS, stably partitioned by what it needs next (brick-grid moves first, then
// candidates, empty slots last), and is picked up by thread `slot`.  Waves thereby become homogeneous -- a wave of
// moving rays runs moves with all its lanes, the candidates sit together in another wave, the empty lanes in the last
// wave refill from the queue -- at the price of 2 x 15 LDS transfers per ray and redistribution and two workgroup
// barriers.  Only what the brick-grid move needs travels through LDS; what only candidate resolution reads (origin in
// brick units, entry distance; the direction is in the queue record) is parked in a per-slot global scratch record.
// What a ray computes is unchanged (same device functions, same operands, results written per queue slot).
#ifndef BM_WG_STEPS
#define BM_WG_STEPS 4 // single brick-grid moves per phase (when the wave is not jumping)
#endif
#ifndef BM_WG_PERIOD
#define BM_WG_PERIOD 4 // phases a wave runs between two redistributions (1: 1.08 ms, 2: 1.02, 3-4: 1.01, 6: 1.03 per config-2 frame)
#endif
#ifndef BM_WG_CAND_MIN
#define BM_WG_CAND_MIN 8 // a wave resolves candidates when this many lanes hold one (or when they outnumber its moving lanes)
#endif
#ifndef BM_WG_GRAB
#define BM_WG_GRAB 256 // queue slots a workgroup reserves per ticket atomic
#endif
constexpr int kPoolFields = 17;

template <bool CONNECT, bool DBG>
__global__ __launch_bounds__(256, DBG ? 2 : 4) void wf_trace_wg(const DeviceScene sc, const FrameConstants* __restrict__ fcp, WfState* __restrict__ st,
															   WfRay* __restrict__ work, const WfShadow* __restrict__ shadow, float4* __restrict__ accum,
															   DeviceCounters* __restrict__ counters, uint32_t queue_size, float4* __restrict__ cold) {
	const FrameConstants& fc = *fcp;
	__shared__ unsigned long long lds_brick[8 * 256];
	__shared__ uint32_t pool[kPoolFields][256];
	__shared__ uint32_t wave_cnt[3][4]; // rays that want a cube jump / single moves / candidate resolution, per wave
	__shared__ uint32_t s_base, s_take, s_more;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const uint32_t total = CONNECT ? st->shadow_ray_cnt : queue_size;
	uint32_t* ticket = CONNECT ? st->connect_ticket : st->extend_ticket;

	RayState r;
	r.hit = false;
	r.n = mk(0.f, 0.f, 0.f);
	r.field_off = 0u;
	r.cube = 0u;
	Tally tally;
	HitInfo info;
	int state = ST_NEED;
	bool ended = false; // the lane holds a ray that has ended and whose result is not written yet
	uint32_t idx = 0;
	uint32_t cur = 0, end = 0; // thread 0: the workgroup's private slot range
	bool more = true;          // thread 0: the queue may hold more slots
	uint32_t runsA = 0, lanesA = 0, runsB = 0, lanesB = 0, runsR = 0, lanesR = 0;
	long long rounds_left = (static_cast<long long>(total) + 256) * (2ll * sc.cells + sc.cells_height + 64); // hang guard only

	for (;;) {
		// ---- retire
		if (ended) {
			ended = false;
			if (CONNECT) {
				if (DBG) tally.shadow_rays++;
				if (!r.hit) {
					const float* s = reinterpret_cast<const float*>(shadow + idx);
					atomic_add_rgb(accum + __float_as_uint(s[9]), mk(s[6], s[7], s[8]));
				}
			} else {
				if (DBG) tally.extend_rays++;
				float* rec = reinterpret_cast<float*>(work + idx);
				rec[9] = r.n.x; rec[10] = r.n.y; rec[11] = r.n.z;
				rec[12] = r.hit ? r.distance : kVeryFar;
			}
		}
		// ---- stable partition of the workgroup's live rays: moving, then candidates; slot = position in that order
		// three classes, so that waves become homogeneous in the kind of walk as well: cube jumps, single moves, candidates
		const bool isJ = state == ST_JUMP, isO = state == ST_OUTER, isC = state == ST_CAND;
		const unsigned long long bJ = __ballot(isJ), bO = __ballot(isO), bC = __ballot(isC);
		if (lane == 0) {
			wave_cnt[0][wave] = static_cast<uint32_t>(__popcll(bJ)); wave_cnt[1][wave] = static_cast<uint32_t>(__popcll(bO));
			wave_cnt[2][wave] = static_cast<uint32_t>(__popcll(bC));
		}
		__syncthreads();
		uint32_t nJ = 0, nO = 0, nC = 0, preJ = 0, preO = 0, preC = 0;
		for (int w = 0; w < 4; ++w) {
			const uint32_t j = wave_cnt[0][w], o = wave_cnt[1][w], c = wave_cnt[2][w];
			if (w < wave) { preJ += j; preO += o; preC += c; }
			nJ += j; nO += o; nC += c;
		}
		const uint32_t live = nJ + nO + nC;
		const unsigned long long below = (1ull << lane) - 1ull;
		if (isJ || isO || isC) {
			const uint32_t dest = isJ ? preJ + static_cast<uint32_t>(__popcll(bJ & below))
									  : (isO ? nJ + preO + static_cast<uint32_t>(__popcll(bO & below)) : nJ + nO + preC + static_cast<uint32_t>(__popcll(bC & below)));
			pool[0][dest] = __float_as_uint(r.tx); pool[1][dest] = __float_as_uint(r.ty); pool[2][dest] = __float_as_uint(r.tz);
			pool[3][dest] = __float_as_uint(r.dx); pool[4][dest] = __float_as_uint(r.dy); pool[5][dest] = __float_as_uint(r.dz);
			pool[6][dest] = r.p; pool[7][dest] = static_cast<uint32_t>(r.sx); pool[8][dest] = static_cast<uint32_t>(r.stepy);
			pool[9][dest] = static_cast<uint32_t>(r.stepz);
			pool[10][dest] = __float_as_uint(r.n.x); pool[11][dest] = __float_as_uint(r.n.y); pool[12][dest] = __float_as_uint(r.n.z);
			pool[13][dest] = static_cast<uint32_t>(r.last_step); pool[14][dest] = idx;
			pool[15][dest] = r.field_off; pool[16][dest] = r.cube;
		}
		if (tid == 0) { // hand out queue slots to the empty lanes (threads live .. 255) from the workgroup's private range
			if (cur == end && more) {
				const uint32_t b = atomicAdd(ticket, static_cast<uint32_t>(BM_WG_GRAB));
				if (b >= total) more = false;
				else { cur = b; end = total - b < static_cast<uint32_t>(BM_WG_GRAB) ? total : b + static_cast<uint32_t>(BM_WG_GRAB); }
			}
			const uint32_t avail = end - cur, empty = 256u - live;
			const uint32_t take = empty < avail ? empty : avail;
			s_base = cur; s_take = take;
			cur += take;
			s_more = (more || cur < end) ? 1u : 0u;
		}
		__syncthreads();
		const uint32_t take = s_take, base = s_base;
		if (live == 0 && take == 0 && s_more == 0u) break; // (uniform over the workgroup)
		if (--rounds_left < 0) break;
		if (static_cast<uint32_t>(tid) < live) {
			r.tx = __uint_as_float(pool[0][tid]); r.ty = __uint_as_float(pool[1][tid]); r.tz = __uint_as_float(pool[2][tid]);
			r.dx = __uint_as_float(pool[3][tid]); r.dy = __uint_as_float(pool[4][tid]); r.dz = __uint_as_float(pool[5][tid]);
			r.p = pool[6][tid]; r.sx = static_cast<int>(pool[7][tid]); r.stepy = static_cast<int>(pool[8][tid]);
			r.stepz = static_cast<int>(pool[9][tid]);
			r.n = mk(__uint_as_float(pool[10][tid]), __uint_as_float(pool[11][tid]), __uint_as_float(pool[12][tid]));
			r.last_step = static_cast<int>(pool[13][tid]); idx = pool[14][tid];
			r.field_off = pool[15][tid];
			r.cube = pool[16][tid];
			r.hit = false;
			state = static_cast<uint32_t>(tid) < nJ ? ST_JUMP : (static_cast<uint32_t>(tid) < nJ + nO ? ST_OUTER : ST_CAND);
		} else {
			state = ST_NEED;
			// ---- refill: the empty lanes are the last threads of the workgroup
			const uint32_t k = static_cast<uint32_t>(tid) - live;
			if (k < take) {
				idx = base + k;
				f3 o, d;
				if (CONNECT) {
					const float2* s = reinterpret_cast<const float2*>(shadow + idx);
					const float2 u = s[0], v = s[1], w2 = s[2];
					o = mk(u.x, u.y, v.x);
					d = mk(v.y, w2.x, w2.y);
					r.n = mk(0.f, 0.f, 0.f); // connect passes a zeroed normal (kernel.cu:338)
				} else {
					const float4* q = reinterpret_cast<const float4*>(work + idx);
					const float4 u = q[0], v = q[1], w2 = q[2];
					o = mk(u.x, u.y, u.z);
					d = mk(u.w, v.x, v.y);
					r.n = mk(w2.y, w2.z, w2.w);
				}
				state = ray_setup<DBG>(sc, o, d, r, tally);
				ended = state == ST_NEED; // missed the world box
				if (!ended) cold[idx] = make_float4(r.o.x, r.o.y, r.o.z, r.tminn); // read back by candidate resolution
			}
		}
		if (DBG && take > 0 && tid == 0) { runsR++; lanesR += take; }
		// ---- BM_WG_PERIOD phases per wave and redistribution: each time whichever the majority of its live lanes wants
#pragma unroll 1
		for (int ph = 0; ph < BM_WG_PERIOD; ++ph) {
			const int nJw = __popcll(__ballot(state == ST_JUMP));
			const int nAw = __popcll(__ballot(state == ST_OUTER)) + nJw, nBw = __popcll(__ballot(state == ST_CAND));
			if (nBw > 0 && (nBw >= BM_WG_CAND_MIN || nBw >= nAw)) {
				if (DBG && lane == 0) { runsB++; lanesB += nBw; }
				if (state == ST_CAND) {
					// what only this phase reads is not carried through the pool: the ray's origin in brick units and entry
					// distance (scratch record), its direction (queue record)
					const float4 c = cold[idx];
					const float* q = CONNECT ? reinterpret_cast<const float*>(shadow + idx) : reinterpret_cast<const float*>(work + idx);
					r.o = mk(c.x, c.y, c.z);
					r.tminn = c.w;
					r.d = mk(q[3], q[4], q[5]);
					state = process_candidate<DBG>(sc, fc.campos, r, info, tally, lds_brick);
					if (state == ST_NEED) ended = true;
				}
			} else if (nAw > 0) {
				const bool walking = state == ST_OUTER || state == ST_JUMP;
				uint32_t ra = 0, la = 0;
				state = walk_round<DBG, BM_WG_STEPS>(sc, r, state, nJw, nAw - nJw, tally, ra, la);
				if (DBG && lane == 0) { runsA += ra; lanesA += la; }
				if (walking && state == ST_NEED) ended = true; // left the grid
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
			const unsigned long long st8[8] = {runsA, lanesA, runsB, lanesB, runsR, lanesR, 0ull, 0ull};
			for (int k = 0; k < 8; ++k) atomicAdd(&counters->sched[k], st8[k]);
		}
	}
}

// shade (kernel.cu:242-325), two passes over the work queue with a prefix sum in between.
//   EMIT = false: count the survivors and shadow rays of each 256-slot block
//   EMIT = true : redo the (cheap, coherent) shading and write every output at offset[block] + rank in block
template <bool EMIT>
__global__ __launch_bounds__(256) void wf_shade(const WfRay* __restrict__ work, WfRay* __restrict__ next, WfShadow* __restrict__ shadow_out,
											   float4* __restrict__ accum, uint2* __restrict__ block_counts, const FrameConstants* __restrict__ fcp,
											   uint32_t queue_size) {
	const FrameConstants& fc = *fcp;
	__shared__ uint32_t wave_counts[2][4];
	const uint32_t index = blockIdx.x * 256u + threadIdx.x;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	bool survive = false, cast = false;
	f3 origin = mk(0.f, 0.f, 0.f), direction = mk(0.f, 0.f, 0.f), throughput = mk(1.f, 1.f, 1.f), normal = mk(0.f, 0.f, 0.f);
	f3 sun_dir = mk(0.f, 0.f, 0.f);
	float distance = 0.f, sunLight = 0.f;
	int identifier = 0, bounces = 0;
	uint32_t pixel_index = 0;
	uint32_t seed = 0;
	bool is_hit = false;
	if (index < queue_size) {
		const float4* q = reinterpret_cast<const float4*>(work + index);
		const float4 a = q[0], b = q[1], c = q[2], d = q[3];
		origin = mk(a.x, a.y, a.z);
		direction = mk(a.w, b.x, b.y);
		throughput = mk(b.z, b.w, c.x);
		normal = mk(c.y, c.z, c.w);
		distance = d.x;
		identifier = __float_as_int(d.y);
		bounces = __float_as_int(d.z);
		pixel_index = __float_as_uint(d.w);
		seed = (fc.base_frame * pixel_index * 147565741u) * 720898027u * index;
		is_hit = distance < kVeryFar;
		if (is_hit) {
			origin = origin + direction * distance;
			origin = origin + normal * 2.f * kEpsilon;
			throughput = throughput * mk(1.f, 1.f, 1.f); // kernel.cu:271
			sun_dir = cone_sample(fc, seed);
			sunLight = dot(normal, sun_dir);
			cast = sunLight > 0.f;
			survive = bounces < fc.max_bounces;
		}
	}
	const unsigned long long bs = __ballot(survive), bc = __ballot(cast);
	if (lane == 0) { wave_counts[0][wave] = static_cast<uint32_t>(__popcll(bs)); wave_counts[1][wave] = static_cast<uint32_t>(__popcll(bc)); }
	__syncthreads();
	if (!EMIT) {
		if (threadIdx.x == 0)
			block_counts[blockIdx.x] = make_uint2(wave_counts[0][0] + wave_counts[0][1] + wave_counts[0][2] + wave_counts[0][3],
												  wave_counts[1][0] + wave_counts[1][1] + wave_counts[1][2] + wave_counts[1][3]);
		return;
	}
	if (index >= queue_size) return;
	const uint2 base = block_counts[blockIdx.x]; // exclusive prefix sums after wf_scan
	uint32_t pos_s = base.x, pos_c = base.y;
	for (int w = 0; w < wave; ++w) { pos_s += wave_counts[0][w]; pos_c += wave_counts[1][w]; }
	const unsigned long long below = (1ull << lane) - 1ull;
	pos_s += static_cast<uint32_t>(__popcll(bs & below));
	pos_c += static_cast<uint32_t>(__popcll(bc & below));
	float* px = reinterpret_cast<float*>(accum + pixel_index);
	if (is_hit) {
		if (cast) { // kernel.cu:275-279
			const f3 color = ((throughput * sun_radiance(fc, sun_dir)) * sunLight) * 1E-5f;
			float2* s = reinterpret_cast<float2*>(shadow_out + pos_c);
			s[0] = make_float2(origin.x, origin.y);
			s[1] = make_float2(origin.z, sun_dir.x);
			s[2] = make_float2(sun_dir.y, sun_dir.z);
			s[3] = make_float2(color.x, color.y);
			s[4] = make_float2(color.z, __uint_as_float(pixel_index));
		}
		if (survive) { // kernel.cu:281-299
			const f3 nd = bounce_direction(normal, seed);
			float4* rec = reinterpret_cast<float4*>(next + pos_s);
			rec[0] = make_float4(origin.x, origin.y, origin.z, nd.x);
			rec[1] = make_float4(nd.y, nd.z, throughput.x, throughput.y);
			rec[2] = make_float4(throughput.z, normal.x, normal.y, normal.z);
			rec[3] = make_float4(distance, __int_as_float(identifier), __int_as_float(bounces + 1), __uint_as_float(pixel_index));
		} else {
			unsafeAtomicAdd(px + 3, 1.f); // kernel.cu:301
		}
	} else { // kernel.cu:316-323
		const f3 color = throughput * (bounces == 0 ? sunsky_radiance(fc, direction) : sky_radiance(fc, direction));
		unsafeAtomicAdd(px + 0, color.x);
		unsafeAtomicAdd(px + 1, color.y);
		unsafeAtomicAdd(px + 2, color.z);
		unsafeAtomicAdd(px + 3, 1.f);
	}
}

// exclusive prefix sums of the per-block counts (in place); the totals become the new queue lengths
__global__ __launch_bounds__(1024) void wf_scan(uint2* __restrict__ block_counts, uint32_t blocks, WfState* __restrict__ st) {
	__shared__ uint2 partial[1024];
	const uint32_t per = (blocks + 1023u) / 1024u;
	const uint32_t first = threadIdx.x * per;
	uint2 sum = make_uint2(0u, 0u);
	for (uint32_t i = first; i < first + per && i < blocks; ++i) { sum.x += block_counts[i].x; sum.y += block_counts[i].y; }
	partial[threadIdx.x] = sum;
	__syncthreads();
	for (uint32_t off = 1; off < 1024u; off <<= 1) { // Hillis-Steele inclusive scan
		uint2 v = partial[threadIdx.x];
		if (threadIdx.x >= off) { const uint2 o = partial[threadIdx.x - off]; v.x += o.x; v.y += o.y; }
		__syncthreads();
		partial[threadIdx.x] = v;
		__syncthreads();
	}
	uint2 run = threadIdx.x ? partial[threadIdx.x - 1] : make_uint2(0u, 0u);
	for (uint32_t i = first; i < first + per && i < blocks; ++i) {
		const uint2 c = block_counts[i];
		block_counts[i] = run;
		run.x += c.x; run.y += c.y;
	}
	if (threadIdx.x == 1023) {
		const uint2 t = partial[1023];
		st->primary_ray_cnt = t.x; st->last_survivors = t.x;
		st->shadow_ray_cnt = t.y; st->last_shadow = t.y;
	}
}

// ---- host-callable launchers (kernels.h)
int wavefront_blocks_per_cu(bool connect, bool instrumented) {
	int n = 0;
	hipError_t e;
	if (connect) e = instrumented ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, BM_WF_TRACE<true, true>, 256, 0)
								  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, BM_WF_TRACE<true, false>, 256, 0);
	else e = instrumented ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, BM_WF_TRACE<false, true>, 256, 0)
						  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, BM_WF_TRACE<false, false>, 256, 0);
	return e == hipSuccess && n > 0 ? n : 1;
}

void launch_wf_primary(WfState* st, WfRay* work, const FrameConstants* fc_dev, uint32_t queue_size, uint32_t pixels, hipStream_t stream) {
	hipLaunchKernelGGL(wf_primary, dim3((queue_size + 255u) / 256u), dim3(256), 0, stream, st, work, fc_dev, queue_size);
	hipLaunchKernelGGL(wf_globals, dim3(1), dim3(1), 0, stream, st, queue_size, pixels);
}

void launch_wf_trace(bool connect, const DeviceScene& sc, const FrameConstants* fc_dev, WfState* st, WfRay* work, const WfShadow* shadow, float* accum,
					 DeviceCounters* counters, uint32_t queue_size, int resident_blocks, void* cold_scratch, hipStream_t stream) {
	long long blocks = (static_cast<long long>(queue_size) + BM_WF_GRAB * 4 - 1) / (BM_WF_GRAB * 4); // never more waves than slot ranges
	if (blocks > resident_blocks) blocks = resident_blocks;
	if (blocks < 1) blocks = 1;
	const dim3 grid(static_cast<unsigned>(blocks)), block(256);
	float4* acc = reinterpret_cast<float4*>(accum);
	float4* cold = reinterpret_cast<float4*>(cold_scratch);
	if (connect) {
		if (counters) hipLaunchKernelGGL((BM_WF_TRACE<true, true>), grid, block, 0, stream, sc, fc_dev, st, work, shadow, acc, counters, queue_size, cold);
		else hipLaunchKernelGGL((BM_WF_TRACE<true, false>), grid, block, 0, stream, sc, fc_dev, st, work, shadow, acc, counters, queue_size, cold);
	} else {
		if (counters) hipLaunchKernelGGL((BM_WF_TRACE<false, true>), grid, block, 0, stream, sc, fc_dev, st, work, shadow, acc, counters, queue_size, cold);
		else hipLaunchKernelGGL((BM_WF_TRACE<false, false>), grid, block, 0, stream, sc, fc_dev, st, work, shadow, acc, counters, queue_size, cold);
	}
}

void launch_wf_shade(const WfRay* work, WfRay* next, WfShadow* shadow, float* accum, void* block_counts, WfState* st, const FrameConstants* fc_dev,
					 uint32_t queue_size, hipStream_t stream) {
	const uint32_t blocks = (queue_size + 255u) / 256u;
	uint2* counts = reinterpret_cast<uint2*>(block_counts);
	float4* acc = reinterpret_cast<float4*>(accum);
	hipLaunchKernelGGL(wf_shade<false>, dim3(blocks), dim3(256), 0, stream, work, next, shadow, acc, counts, fc_dev, queue_size);
	hipLaunchKernelGGL(wf_scan, dim3(1), dim3(1024), 0, stream, counts, blocks, st);
	hipLaunchKernelGGL(wf_shade<true>, dim3(blocks), dim3(256), 0, stream, work, next, shadow, acc, counts, fc_dev, queue_size);
}

} // namespace bm
