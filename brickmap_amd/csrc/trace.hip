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
// Launch geometry: persistent 256-thread workgroups (compute units x resident blocks); waves pull rows of 4x4-pixel
// chunks from interleaved ticket counters and schedule their lanes' work in phases (see trace_paths below).
// The device functions shared with the queue-based schedule (wavefront.hip) live in traverse.h.
#include <type_traits>

#include "kernels.h"
#include "traverse.h"

namespace bm {

// Persistent-wave path tracer.
//
// Work distribution: the shard's pixels are cut into 4x4-pixel chunks (ordered so that four consecutive
// chunks form an 8x8 block and sixteen a 16x16 tile).  Waves are persistent: whenever FrameConstants::refill_min or more of a
// wave's lanes have no work item (24 with helper lanes; in ordered frames 16 for one-sample items, fewer for long ones), the wave takes that
// many items, pixel by pixel through consecutive 4x4 chunks, from a global counter (one atomic per refill) and hands one to each idle
// lane.  ORDERED frames (BM_FLAG_ORDERED, hit records): an item is a pixel, the lane traces ALL its samples in order before it takes
// another one, so each pixel's accumulation order is fixed (sample by sample, event by event).  Production frames: an item is one
// sample of a pixel, shadow rays may run on helper lanes (HELP below), every event is added to the pixel with float atomics.
//
// Path state machine per lane:
//   GEN -> [extend ray] -> EXT_DONE (shade) -> [shadow ray] -> SHD_DONE (connect) -> BOUNCE -> [extend ray] ...
// A wave interleaves three kinds of work, each run only when enough lanes want it (or nothing else can
// run), so that the expensive, rarely-needed code never executes for a handful of lanes:
//   phase A  the brick-grid walk: exact jumps over empty cubes (ST_JUMP) / single moves near the surface (ST_OUTER),
//            in bursts of up to BM_JUMP_PASSES passes                                   (jump.h, traverse.h)
//   phase B  index word + 8^3 / 2^3 bitmask DDA   lanes in ST_CAND              (expensive, ~2.5 per ray)
//   phase C  shade / next primary ray, connect (lanes in ST_CONN: the result of a shadow ray), and ONE ray set-up for
//            whatever ray each of those lanes traces next                       (expensive, once per ray)
// Scheduling changes only WHEN a lane's operations happen, never their operands, so results are
// identical to the reference's per-ray functions run one ray at a time (the oracle).
// floor(n / d) for a per-frame constant d whose multiply-high constants the host prepared (FrameConstants::div_*; magic == 0: d == 1);
// n < 2^30.  Three instructions instead of the ~17 of a 32-bit division by a run-time value.
__device__ __forceinline__ uint32_t div_const(uint32_t n, uint32_t magic, int shift) { return magic ? (__umulhi(n, magic) >> shift) : n; }

// The buffers of a frame are named by its constants (the frame ring), i.e. by pointers READ FROM MEMORY: generic pointers to the
// compiler (flat_* instructions) unless an access says that it goes to global memory.
typedef float bm_v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) bm_v4f v4f_in_global_memory;
typedef __attribute__((address_space(1))) uint32_t u32_in_global_memory;
__device__ __forceinline__ float4 pixel_load(const float4* p) { const bm_v4f v = *(const v4f_in_global_memory*)p; return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void pixel_store(float4* p, float4 v) { const bm_v4f w = {v.x, v.y, v.z, v.w}; *(v4f_in_global_memory*)p = w; }
__device__ __forceinline__ void record_atomic_add(uint32_t* p, uint32_t v) { (void)__hip_atomic_fetch_add((u32_in_global_memory*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// The reference's atomicAdd on the accumulation buffer (kernel.cu:319-322,341-343), for a buffer whose address was READ FROM MEMORY
// (the frame ring's constants): to the compiler such a pointer is generic and the add becomes flat_atomic_add_f32; the pixel
// records live in global memory, and saying so gives global_atomic_add_f32 as with a kernel-argument pointer (same relaxed,
// device-scope, result-unused read-modify-write as hip's unsafeAtomicAdd).
__device__ __forceinline__ void pixel_atomic_add(float* p, float v) {
	typedef __attribute__((address_space(1))) float float_in_global_memory;
	[[clang::atomic(ignore_denormal_mode)]] { (void)__hip_atomic_fetch_add((float_in_global_memory*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
}

enum : int { P_GEN = 0, P_EXT_DONE = 1, P_SHD_DONE = 2, P_BOUNCE = 3, P_HELPER = 6 };
enum : int { ST_IDLE = 4, ST_CONN = 5 };

// Resident waves per SIMD of the production instantiations (the register budget follows: 512 / waves, in steps of 8).  Round 5:
// with the brick staged straight into LDS (traverse.h brick_dma_to_lds) no lane holds a brick's 16 registers any more -- 84 VGPRs
// instead of 95 -- and six waves fit without a spill: config 2 1.066 -> 1.020 ms, config 3 23.4 -> 21.8, config 5 113.7 -> 105.1.
// Seven waves (72 VGPRs) first cost five spilled registers and paid on the big workloads only; once helper-lane frames kept no
// accumulator (below) the two values still spilled are constants of a cold branch (rays that start outside the world), and seven
// waves pay everywhere: config 2 1.023 -> 1.009 ms, 1080p at 4 spp -2 %, configs 3 / 5 -2 % (profiles/r05_occupancy.txt,
// r05_deep.txt).  Eight waves (64 VGPRs, 13+ spilled) lose 8-16 %.
#ifndef BM_WAVES_PER_SIMD
#define BM_WAVES_PER_SIMD 7
#endif
#ifndef BM_ITEM_LANES
// pixels handed out per ticket: 1 = a refill fills EVERY idle lane (consecutive tickets still walk through a 4x4 chunk row by row).
// 4 (a 4x1 row per ticket: up to 3 idle lanes stay empty) is 1 % slower on every workload, 16 (whole chunks) 2.5 %
// (profiles/r04_refill_sweep.txt)
#define BM_ITEM_LANES 1
#endif
// Wave priorities (s_setprio) per pass: a wave in a walk or candidate pass -- short, and ending in a dependent load -- issues
// before a wave in the long arithmetic of a shade pass, which fills the gaps: 1.169 -> 1.118 ms on config 2 (any of
// walk / candidate / shade = 1/2/0, 2/3/0, 1/3/0, 2/3/1 within 0.5 %; shade highest: 1.157, all equal: no change).
#ifndef BM_PRIO
#define BM_PRIO 1
#endif
#ifndef BM_PRIO_A
#define BM_PRIO_A 1
#endif
#ifndef BM_PRIO_B
#define BM_PRIO_B 2
#endif
#ifndef BM_PRIO_C
#define BM_PRIO_C 0
#endif
// ticket counters of the interleaved hand-out (the XCD-aware hand-out has one per XCD: 8).  16 instead of 8: -0.3 ... -0.6 % on 1080p frames
// (profiles/r05_counters.txt); 4: +2.5 %
#ifndef BM_WORK_COUNTERS
#define BM_WORK_COUNTERS 16
#endif
#ifndef BM_QUORUM_NUM
#define BM_QUORUM_NUM 1
#endif
#ifndef BM_QUORUM_SHADE_NUM
#define BM_QUORUM_SHADE_NUM 1
#endif
#ifndef BM_QUORUM_DIV
#define BM_QUORUM_DIV 2
#endif
#ifndef BM_QUORUM_SHADE_DIV
#define BM_QUORUM_SHADE_DIV 4
#endif
#ifndef BM_STEPS_PER_ROUND
#define BM_STEPS_PER_ROUND 4
#endif
// -DBM_PHASE_TIMING: profiling build in which the plain kernel also reports the scheduler statistics
// -DBM_ISA_MARKERS: comment lines in the compiler's .s output at the boundaries of the scheduler's passes (tools/isa_passes.py
// turns them into the per-pass instruction table of profiles/); a listing aid only -- the product is built without them
#ifdef BM_ISA_MARKERS
#define BM_REGION(name) asm volatile("; BM_REGION " name)
#else
#define BM_REGION(name) do { } while (0)
#endif
#ifdef BM_PHASE_TIMING
#define BM_TIMED true
#else
#define BM_TIMED DBG
#endif
// XCD: the XCD-aware hand-out (FrameConstants::xcd_handout; big frames) -- an instantiation of its own, so that the hand-out code of
// the headline kernel stays what it was (as a run-time branch it cost config 2 0.5-1 %)
// HELP: shadow rays on helper lanes (FrameConstants::helpers).  A path's shadow ray (connect, kernel.cu:328-346) and its next extend
// ray are independent -- the reference itself traces them from two different queues -- but a lane that traces both does so one
// after the other.  With HELP, a shade pass that finds idle lanes in its wave hands the shadow rays of the lanes it just shaded to
// them (ten words through the LDS brick staging area, which no shade pass uses), the owner goes straight on with its bounce ray (or
// ends the path), and the helper adds the unoccluded sun light to the pixel with float atomics, as the reference's connect does
// (kernel.cu:341-343).  The rays traced are the same rays; what changes is who walks them: idle lanes do useful work without a
// refill, and a path's latency -- which is what the end of a frame waits for -- is its extend rays' alone.  Pixels are therefore
// written back with atomics as well (the accumulator starts at zero in the lane), so radiance is equal up to summation order.
// RING: 0 = a launch of one frame; 1 = several frames, a wave changes frame when all its lanes are idle; 2 = several frames of a
// UNIFORM launch, whose lanes carry their frame themselves (below) -- then every constant is read from the first frame's entry, at a
// fixed address, and the frame's index is only needed where a wave takes items.
// The launch may hold more than one frame (the frame ring, below).  An instantiation of its own because the scheduler loop runs
// at the limit of the scalar register file: the ring's one extra loop-carried scalar and the indexed constants cost a single-frame
// launch 2.6 % (config 2 1.005 -> 1.031 ms, config 3 19.8 -> 20.4; four more spilled scalars in the hot loop), which a launch of
// ONE frame has no reason to pay.
template <bool DBG, bool XCD = false, bool HELP = false, int RING = 0>
// (the instrumented variant carries hit records and counters: it gets the registers instead of the occupancy)
__global__ __launch_bounds__(256, DBG ? 2 : BM_WAVES_PER_SIMD) void trace_paths(const DeviceScene sc, const FrameConstants* __restrict__ fcp,
												  DeviceCounters* __restrict__ counters, uint32_t* __restrict__ work_counter) {
	// the per-frame constants live in device memory (not in the kernel-argument registers): they are read with scalar
	// loads where they are needed, which keeps the scalar register file free for the scheduler loop
	//
	// FRAME RING.  One launch traces one or more consecutive frames (bm_render_frames; the reference's loop is one launch_kernels call
	// per frame, main.cpp:117-147, kernel.cu:416-420): fcp[0], fcp[1], ... are their constants (FrameConstants::frames_after of the
	// last one is 0) -- camera, sun, sample_base, base_frame and the accumulation / hit-record buffers may differ from frame to
	// frame; everything that shapes the hand-out (size, samples, flags, shard, max_bounces) is the same for all of them
	// (Scene::render_frames checks) and is read through `fg` below -- and every frame has its own block of ticket counters behind
	// `work_counter`.  A wave that finds the counters of its frame used up lets its lanes finish their paths and then moves on to
	// the next frame BY ITSELF: the waves of a frame do not wait for one another, so the end of frame i -- the latency of the
	// paths that started last, a sixth of a 1080p / 1-spp frame -- is covered by the beginning of frame i+1 instead of an idle GPU,
	// whatever the runtime does with streams.  The constants stay wave-uniform: all lanes of a wave are always in the same frame.
	//
	// What the ring may cost a single frame is scalar registers -- the scheduler loop runs at the limit of the scalar file, and every
	// value carried round it for the ring's sake is a spill (v_readlane / v_writelane in the hot loop: the first version, with the
	// frame count, the first frame's buffers and the per-frame round budget as live scalars, was 4 % slower on every workload).  So
	// the loop carries ONE extra scalar, the frame's index: the buffers are read from the frame's constants when a wave enters it,
	// "is this the last frame" is a field of the constants read when a wave has run dry, and what a frame start needs (first ticket
	// counter, round budget) is recomputed there.  The frame is an INDEX into the `__restrict__` array, never a pointer carried
	// round the loop: only then does the compiler prove that the kernel's stores leave the constants alone (scalar loads).
	const FrameConstants& fg = *fcp;     // what all frames of the launch share
	int ring_pos = 0;                    // the frame of the launch this wave is in (RING; otherwise the constant 0)
	constexpr bool kRing = RING != 0, kUniform = RING == 2;
#define fc (fcp[RING == 1 ? ring_pos : 0])
	// are there frames after the one this wave is in?  (uniform: the first entry knows how many follow IT)
	auto more_frames = [&]() { return kUniform ? ring_pos < fg.frames_after : fc.frames_after > 0; };
	float4* accum = reinterpret_cast<float4*>(fc.accum);
	uint32_t* dbg = DBG ? fc.dbg : nullptr;
	__shared__ unsigned long long lds_brick[8 * 256]; // 16 KiB: one 64-byte brick per thread (traverse.h brick_dma_to_lds: word k of thread t at u32 word k * 256 + t)
	const int lane = threadIdx.x & 63;
	const uint32_t W = static_cast<uint32_t>(fg.width), H = static_cast<uint32_t>(fg.height);
	const uint32_t total_chunks = static_cast<uint32_t>(fg.tiles_x) * static_cast<uint32_t>(fg.tiles_y) * 16u;
	// Work items.  Default: a lane traces ALL samples of its pixel in order (fixed accumulation order per pixel, one plain
	// write-back).  BM_FLAG_SAMPLE_ITEMS: the item is ONE sample of a 4x4 chunk -- spp times more, spp times shorter items,
	// which keeps the persistent waves fed when a shard has few pixels and many samples (the 1/N row-band shards of a
	// multi-GPU frame); samples of one pixel then run on different lanes and are added with float atomics like the
	// reference does (kernel.cu:319-322,341-343), so radiance is equal up to summation order.
	const bool sample_items = (fg.flags & 4u) != 0u; // BM_FLAG_SAMPLE_ITEMS
	constexpr uint32_t kParts = 16u / BM_ITEM_LANES; // tickets per chunk and sample
	const uint32_t items_per_chunk = (sample_items ? static_cast<uint32_t>(fg.spp > 0 ? fg.spp : 1) : 1u) * kParts;
	const bool atomic_acc = HELP || sample_items; // other lanes may add to the pixel while this one holds it: add, never overwrite

	// per-pixel state
	uint32_t xy = 0;          // x | y << 16
	uint32_t p = 0;           // global pixel index y*W + x
	uint32_t local_pixel = 0; // index into this shard's packed buffers
	Tally tally;
	HitInfo info;
	uint32_t d0 = 0, d1 = 0, d2 = 0xFFFFFFFFu, d3 = 0, hseg = 2166136261u, hsh = 2166136261u, next = 0, nsh = 0, loads0 = 0;
	// BM_FLAG_RAY_DIGEST (instrumented frames only): the hit records of a frame whose rays are traced by whichever lane is free -- helper
	// lanes, (chunk, sample) items -- cannot be chains in path order; they are SUMS (mod 2^32) over the pixel's rays of keyed per-ray
	// hashes, added with atomics by the lane that traced the ray (oracle.c render_pixel has the same sums):
	//   word 4 += E(key, hit, distance bits, normal | level, brick, voxel) per extend ray, word 5 += S(key, occluded, brick, voxel | level)
	//   per shadow ray, word 6 += ray counts, word 7 += cells visited; key = sample << 8 | segment of the path the ray belongs to
	const bool ray_digest = DBG && (fg.flags & 32u) != 0u;
	uint32_t ray_key = 0, ray_loads0 = 0; // of the ray in flight

	RayState r;
	r.hit = false;
	r.n = mk(0.f, 0.f, 0.f);
	int state = ST_IDLE;
	int pstate = P_GEN;
	int s = 0;               // sample being traced
	int s_end = 0;           // first sample that is no longer this lane's
	int bounces = 0;
	bool shadow = false;     // kind of the ray in flight
	bool terminated = false; // path ends after its pending shadow ray
	f3 hitp = mk(0.f, 0.f, 0.f);   // surface point: shadow-ray origin and next extend origin
	f3 pn = mk(0.f, 0.f, 0.f);     // surface normal of the path (RayQueue::normal)
	f3 scolor = mk(0.f, 0.f, 0.f); // ShadowQueue::color
	f3 bdir = mk(0.f, 0.f, 0.f);   // next bounce direction, drawn in shade, used once the shadow ray is done
	// the pixel's accumulator (state.h:22): read when the lane takes the pixel, updated in path order in registers, written
	// back once when the pixel is finished -- a read-modify-write in memory per event stalls the wave for a load round trip
	// (HELP: no lane-private accumulator at all -- every event goes to the pixel with float atomics when it happens, like the reference's
	// shade / connect do (kernel.cu:301,319-322,341-343): four registers less per lane, i.e. fewer spills at seven waves per SIMD --
	// config 3 -1.4 %, config 5 -1.7 %, profiles/r05_event_atomics.txt)
	float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
	auto add_rgb = [&](f3 c) {
		if (HELP) { float* a = reinterpret_cast<float*>(accum + local_pixel); pixel_atomic_add(a + 0, c.x); pixel_atomic_add(a + 1, c.y); pixel_atomic_add(a + 2, c.z); }
		else { acc.x += c.x; acc.y += c.y; acc.z += c.z; }
	};
	auto add_terminated = [&]() {
		if (HELP) pixel_atomic_add(reinterpret_cast<float*>(accum + local_pixel) + 3, 1.f);
		else acc.w += 1.f;
	};

	typename std::conditional<RING == 1, int, bool>::type work_left = 1; // (a bool where two states do: as an int it cost the one-frame kernel three spilled vector registers)
	// 1: the frame has tickets left; 0: the launch has none; 2 (RING == 1): the frame is used up, the next one starts once every lane is idle
	constexpr uint32_t kCounters = XCD ? 8u : BM_WORK_COUNTERS, kCounterStride = 32; // one 128-byte line per counter
	// (the wave index is the same in all 64 lanes; saying so keeps everything derived from it -- the counter in use, `work_left`,
	// the loop's exit conditions -- in scalar registers and the scheduler loop's branches scalar)
	// XCD-aware hand-out (FrameConstants::xcd_handout, big frames): the image is cut into super-tiles of kXcdTiles x kXcdTiles tiles
	// (256 x 256 pixels), super-tile st belongs to counter st % 8, and a wave starts on the counter of ITS XCD -- workgroups are
	// dispatched round-robin over the 8 XCDs, so blockIdx % 8 names the L2 -- so that the rays of neighbouring pixels, which read
	// the same field rows, index words and bricks, are traced behind ONE L2 instead of all eight; a wave whose counter is used up
	// helps the next one.  Otherwise: groups of four chunks dealt to the counters, every wave of a workgroup on its own counter.
	// (blockIdx % 8 == XCD is the guide's "observed, for speed only" mapping: nothing here depends on it for correctness -- a wave that starts
	// on another XCD's counter traces the same pixels with worse locality; tests render the same frame under both hand-outs.)
	constexpr uint32_t kXcdTiles = 16u, kStChunks = kXcdTiles * kXcdTiles * 16u;
	constexpr bool xcd_handout = XCD;
	auto first_counter = [&]() {
		return xcd_handout ? static_cast<int>(blockIdx.x % kCounters)
						   : static_cast<int>((blockIdx.x * 4u + static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6)))) % kCounters);
	};
	int my_counter = first_counter();
	int counters_done = 0;
	// hang guard only (NaN directions): no wave needs more scheduler rounds than this
	// (64-bit products are computed by the vector unit: bring the count back into scalar registers, or every test of it
	// turns the scheduler loop's branches into exec-mask branches)
	auto round_budget = [&]() {
		const long long budget = (static_cast<long long>(total_chunks) + 64) * (static_cast<long long>(fg.spp) + 1) * (fg.max_bounces + 2) *
								 (2ll * sc.cells + sc.cells_height + 64) * (kRing ? fg.frames_after + 1 : 1); // (all frames of the launch: computed once, nothing of it lives across the loop)
		return static_cast<long long>((static_cast<unsigned long long>(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(budget >> 32)))) << 32) |
									  static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(budget))));
	};
	long long rounds_left = round_budget();
	uint32_t runsA = 0, lanesA = 0, runsB = 0, lanesB = 0, runsC = 0, lanesC = 0, runsD = 0, lanesD = 0, runsJ = 0, lanesJ = 0; // wave-uniform scheduler statistics

	unsigned long long cycA = 0, cycB = 0, cycC = 0, cycD = 0;
	unsigned long long det[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // BM_PHASE_TIMING: DeviceCounters::detail
#ifdef BM_PHASE_TIMING
#define BM_MARK(k, tprev) do { const unsigned long long bm_now_ = __builtin_amdgcn_s_memtime(); det[k] += bm_now_ - tprev; tprev = bm_now_; } while (0)
#else
#define BM_MARK(k, tprev) do { } while (0)
#endif
	const unsigned long long t_begin = BM_TIMED ? __builtin_amdgcn_s_memtime() : 0ull;
	unsigned long long t_dry = 0ull; // when this wave found the ticket counters empty (BM_TIMED): the rest of its life is the drain

	for (;;) {
		BM_REGION("refill");
		// ---- refill: hand pixels to idle lanes, BM_ITEM_LANES at a time (consecutive tickets walk through a 4x4 chunk)
		const unsigned long long idle = __ballot(state == ST_IDLE);
		const int nI = __popcll(idle);
		if (RING == 1 && work_left == 2 && nI == 64) {
			// this wave has nothing left to do in its frame: on to the next frame of the launch (its constants, its buffers, its own
			// ticket counters).  Every lane is idle here, helpers included, so nothing of the old frame is in flight in this wave;
			// other waves may still be tracing it.  (Whether there IS a next frame was settled when the counters ran dry, below: a
			// test of the constants here, or in the loop's exit condition, is a scalar load and a wait in EVERY scheduler round --
			// that was the ring's 2.6 % per frame.)
			++ring_pos;
			accum = reinterpret_cast<float4*>(fc.accum);
			if (DBG) dbg = fc.dbg;
			work_left = 1; // (same ticket counter, next frame)
		}
		if (work_left == 1 && nI >= fg.refill_min) {
			// One global word serves only ~90 returning atomics per microsecond chip-wide, and a refill stalls the whole
			// wave until its atomic returns; with thousands of waves on one counter that queue is tens of microseconds
			// long.  The chunk sequence is therefore dealt to kCounters interleaved counters (8x8-pixel groups of four
			// chunks, group g on counter g % kCounters, each counter on its own cache line): every counter still sweeps
			// the image top-down, so concurrently running waves keep working on neighbouring rows of the image.
			const int want = nI / BM_ITEM_LANES;
			uint32_t base = 0;
			if (lane == 0) base = atomicAdd(work_counter + (kRing ? static_cast<uint32_t>(ring_pos) * static_cast<uint32_t>(kWorkCounterBytes / sizeof(uint32_t)) : 0u) + my_counter * kCounterStride, static_cast<uint32_t>(want));
			base = __builtin_amdgcn_readfirstlane(base);
			// units dealt to the counters: groups of four chunks, or whole super-tiles
			const uint32_t st_x = (static_cast<uint32_t>(fg.tiles_x) + kXcdTiles - 1u) / kXcdTiles, st_y = (static_cast<uint32_t>(fg.tiles_y) + kXcdTiles - 1u) / kXcdTiles;
			const uint32_t total_units = xcd_handout ? st_x * st_y : (total_chunks + 3u) >> 2;
			const uint32_t my_units = total_units > static_cast<uint32_t>(my_counter)
										  ? (total_units - static_cast<uint32_t>(my_counter) + kCounters - 1u) / kCounters : 0u;
			const uint32_t my_tickets = my_units * (xcd_handout ? kStChunks : 4u) * items_per_chunk; // consecutive tickets = the samples of one chunk
			const uint32_t counter_now = static_cast<uint32_t>(my_counter);
			const int ring_at_refill = kRing ? ring_pos : 0;
			if (kRing && !XCD && base + want >= my_tickets && more_frames()) {
				// this counter of the wave's frame is used up and the launch has more frames: on to the SAME counter of the next frame --
				// at once in a uniform launch (see below), once every lane is idle otherwise.  (Going round the frame's other counters
				// first, as a launch's last frame does, is sixteen refills that hand out nothing, per wave and frame -- 1.5 % of a
				// 1080p frame; whatever the other counters still hold is traced by the waves that are on them.  Not with the XCD-aware
				// hand-out, whose counters own whole super-tiles and may differ by one: there the waves of a light counter would run
				// frames ahead of the others and could only help them on the launch's last frame.)
				if (kUniform) ++ring_pos;
				else work_left = 2;
			} else if (base + want >= my_tickets) { // this counter is used up: move to the next one (helping out), or finish
				my_counter = (my_counter + 1) % static_cast<int>(kCounters);
				if (++counters_done >= static_cast<int>(kCounters)) {
					// the frame's tickets are gone.  UNIFORM launches (FrameConstants::ring_uniform: the frames share view, sun and base_frame;
					// their sample_base and buffers step by constants -- a resting camera accumulating, the reference's own steady state,
					// main.cpp:117-147 with no input) go straight on: a lane's frame is folded into its sample index and its pixel offset
					// when it takes its item (below), every other constant is the same in all frames, so lanes of two frames share
					// the wave and the wave never runs empty between frames -- only at the end of the launch.
					if (kRing && more_frames()) { // (XCD-aware hand-out: the whole frame is handed out, the next one starts on the wave's own counter)
						counters_done = 0;
						my_counter = first_counter();
						if (kUniform) ++ring_pos;
						else work_left = 2;
					} else {
						work_left = 0; // (the launch's last frame: nothing left anywhere)
						if (BM_TIMED) t_dry = __builtin_amdgcn_s_memtime();
					}
				}
			}
			const int rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(idle >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(idle), 0u));
			if (state == ST_IDLE && rank < want * BM_ITEM_LANES) {
				const uint32_t item = base + static_cast<uint32_t>(rank / BM_ITEM_LANES);
				// items_per_chunk = samples x kParts: divide by the power of two first, then by the samples (a prepared constant)
				static_assert((kParts & (kParts - 1u)) == 0u, "kParts is a power of two");
				const uint32_t ticket = div_const(item / kParts, fg.div_samples_magic, fg.div_samples_shift), item_sub = item - ticket * items_per_chunk;
				const uint32_t item_sample = item_sub / kParts, part = item_sub % kParts; // (kParts == 1: part 0)
				uint32_t k;
				int tile_x, tile_y;
				bool in_frame;
				if (xcd_handout) {
					const uint32_t st = (ticket / kStChunks) * kCounters + counter_now, in_st = ticket % kStChunks;
					const uint32_t tw = in_st >> 4;
					k = in_st & 15u;
					const uint32_t st_row = div_const(st, fg.div_st_x_magic, fg.div_st_x_shift);
					tile_x = static_cast<int>((st - st_row * st_x) * kXcdTiles + tw % kXcdTiles);
					tile_y = static_cast<int>(st_row * kXcdTiles + tw / kXcdTiles);
					in_frame = tile_x < fg.tiles_x && tile_y < fg.tiles_y;
				} else {
					const uint32_t chunk = ((ticket >> 2) * kCounters + counter_now) * 4u + (ticket & 3u);
					const uint32_t tile = chunk >> 4;
					k = chunk & 15u;
					tile_y = static_cast<int>(div_const(tile, fg.div_tiles_x_magic, fg.div_tiles_x_shift));
					tile_x = static_cast<int>(tile - static_cast<uint32_t>(tile_y) * static_cast<uint32_t>(fg.tiles_x));
					in_frame = chunk < total_chunks;
				}
				if (item < my_tickets && in_frame) {
					const int cx = static_cast<int>((k & 1u) | ((k >> 1) & 2u)), cy = static_cast<int>(((k >> 1) & 1u) | ((k >> 2) & 2u));
					const uint32_t q = part * BM_ITEM_LANES + (static_cast<uint32_t>(rank) % BM_ITEM_LANES); // pixel of the 4x4 chunk
					const int x = tile_x * 16 + cx * 4 + static_cast<int>(q & 3u);
					const int ly = tile_y * 16 + cy * 4 + static_cast<int>(q >> 2); // row inside this shard's packed buffer
					const int band = static_cast<int>(div_const(static_cast<uint32_t>(ly), fg.div_band_magic, fg.div_band_shift)); // ly / band_rows
					const int y = (band * fg.shard_count + fg.shard_rank) * fg.band_rows + (ly - band * fg.band_rows);
					if (x < fg.width && ly < fg.local_rows && y < fg.height) {
						p = static_cast<uint32_t>(y) * W + static_cast<uint32_t>(x);
						local_pixel = static_cast<uint32_t>(ly) * W + static_cast<uint32_t>(x);
						xy = static_cast<uint32_t>(x) | (static_cast<uint32_t>(y) << 16);
						s = sample_items ? static_cast<int>(item_sample) : 0;
						s_end = sample_items ? s + 1 : fg.spp;
						if (kUniform) { // the frame as an offset of the sample index and of the pixel record
							const int sample_off = __mul24(ring_at_refill, fg.ring_sample_stride);
							s += sample_off; s_end += sample_off;
							local_pixel += static_cast<uint32_t>(ring_at_refill) * fg.ring_pixel_stride;
						}
						pstate = P_GEN;
						state = ST_NEED;
						if (!HELP) acc = atomic_acc ? make_float4(0.f, 0.f, 0.f, 0.f) : pixel_load(accum + local_pixel);
						if (DBG) {
							d0 = 0; d1 = 0; d2 = 0xFFFFFFFFu; d3 = 0; hseg = 2166136261u; hsh = 2166136261u; next = 0; nsh = 0;
							loads0 = tally.index_loads;
						}
					}
				}
			}
		}
		BM_REGION("policy");
		const int nJ = __popcll(__ballot(state == ST_JUMP));
		const int nA = __popcll(__ballot(state == ST_OUTER)) + nJ; // lanes walking the brick grid, cell by cell or cube by cube
		const int nB = __popcll(__ballot(state == ST_CAND));
		const int nC = __popcll(__ballot(state == ST_NEED) | __ballot(state == ST_CONN)); // shade / generate, and connect (same pass)
		const int live = nA + nB + nC;
		--rounds_left;
		if (rounds_left < 0 || (live == 0 && work_left == 0)) break;
		// (live == 0 with work_left == 2: the next round starts the next frame, see the top of the loop)
		// (live == 0 with chunks left: only pixels outside the image were handed out; the passes below find nothing to do)
		// Policy: an expensive phase runs once a quarter of the live lanes wait for it, the cheap connect phase
		// once an eighth does; otherwise the DDA keeps moving.  With no lane left in the DDA the largest group runs.
		const int quorum = (live * BM_QUORUM_NUM + BM_QUORUM_DIV - 1) / BM_QUORUM_DIV;
		const int quorum_shade = (live * BM_QUORUM_SHADE_NUM + BM_QUORUM_SHADE_DIV - 1) / BM_QUORUM_SHADE_DIV;
		int phase; // 0 = A (DDA moves), 1 = B (candidates), 2 = C (shade / generate), 3 = D (connect)
		if (nC >= quorum_shade) phase = 2;
		else if (nB >= quorum) phase = 1;
		else if (nA > 0) phase = 0;
		else phase = nC >= nB ? 2 : 1;

		const unsigned long long t_phase = BM_TIMED ? __builtin_amdgcn_s_memtime() : 0ull;
		if (phase == 2) {
			if (BM_PRIO) __builtin_amdgcn_s_setprio(BM_PRIO_C);
			if (BM_TIMED) {
				const int n_conn = __popcll(__ballot(state == ST_CONN));
				runsC++; lanesC += nC - n_conn;
				if (n_conn) { runsD++; lanesD += n_conn; } // "connect" statistics: passes that held shadow-ray results, and how many
			}
			// ================= phase C: shade the finished extend ray / generate the next primary ray, then set the new ray up
			unsigned long long t_sub = t_phase;
			(void)t_sub;
			BM_REGION("C.connect");
			bool need_setup = false;
			bool hand = false; // HELP: this lane has just drawn a shadow ray that an idle lane may take
			f3 ro = mk(0.f, 0.f, 0.f), rd = mk(0.f, 0.f, 0.f);
			if (state == ST_NEED || state == ST_CONN) {
				if (state == ST_CONN) {
					// ---- connect (kernel.cu:328-346) -- runs after shade within the same reference frame -- then the stored bounce
					// ray is set up by the code below, together with the rays of the lanes that were shaded in this pass
					const bool occluded = r.hit;
					if (DBG) {
						tally.shadow_rays++;
						nsh++;
						hsh = hmix(hsh, static_cast<uint32_t>(occluded));
						if (occluded) {
							hsh = hmix(hsh, static_cast<uint32_t>(info.brick_id));
							hsh = hmix(hsh, static_cast<uint32_t>(info.sub_id) | (static_cast<uint32_t>(info.level) << 12));
						}
						if (ray_digest && dbg) { // (a helper's local_pixel is the owner's: the record is the path's pixel's)
							uint32_t e = hmix(hmix(2166136261u, ray_key), static_cast<uint32_t>(occluded));
							if (occluded) {
								e = hmix(e, static_cast<uint32_t>(info.brick_id));
								e = hmix(e, static_cast<uint32_t>(info.sub_id) | (static_cast<uint32_t>(info.level) << 12));
							}
							uint32_t* d = dbg + static_cast<size_t>(local_pixel) * 8;
							record_atomic_add(d + 5, e); record_atomic_add(d + 6, 1u << 16); record_atomic_add(d + 7, tally.index_loads - ray_loads0);
						}
					}
					if (HELP && pstate == P_HELPER) {
						// a helper's shadow ray: the sun light goes to the OWNER's pixel (kernel.cu:341-343: atomicAdd), the lane is free again
						if (!occluded) {
							float* a = reinterpret_cast<float*>(accum + local_pixel);
							pixel_atomic_add(a + 0, scolor.x); pixel_atomic_add(a + 1, scolor.y); pixel_atomic_add(a + 2, scolor.z);
						}
						state = ST_IDLE; // (pstate stays P_HELPER: none of the blocks below applies)
					} else {
						if (!occluded) {
							add_rgb(scolor);
						}
						state = ST_NEED;
						if (terminated) {
							s++;
							pstate = P_GEN; // the next primary ray (or the pixel hand-back) follows below
						} else {
							bounces++;
							pstate = P_BOUNCE; // neither shaded nor generated below: only set up
							ro = hitp; rd = bdir; r.n = pn; shadow = false; need_setup = true;
						}
					}
				}
				BM_MARK(0, t_sub); // connect
				BM_REGION("C.shade");
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
						if (ray_digest && dbg) {
							uint32_t e = hmix(hmix(2166136261u, ray_key), static_cast<uint32_t>(is_hit));
							if (is_hit) {
								e = hmix(e, __float_as_uint(r.distance));
								e = hmix(e, pack_normal(pn) | (static_cast<uint32_t>(info.level) << 12));
								e = hmix(e, static_cast<uint32_t>(info.brick_id));
								e = hmix(e, static_cast<uint32_t>(info.sub_id));
							}
							uint32_t* d = dbg + static_cast<size_t>(local_pixel) * 8;
							record_atomic_add(d + 4, e); record_atomic_add(d + 6, 1u); record_atomic_add(d + 7, tally.index_loads - ray_loads0);
							if (ray_key == 0u) { // the first extend ray of the launch's first sample: the first-hit record, written by the one lane that traced it
								u32_in_global_memory* g = (u32_in_global_memory*)d;
								g[0] = d0; g[1] = d1; g[2] = d2; g[3] = d3;
							}
						}
					}
					const bool primary_only = fg.flags & 1u; // BM_FLAG_PRIMARY_ONLY
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
						terminated = !(bounces < fg.max_bounces);
						if (terminated) {
							add_terminated(); // kernel.cu:301
						} else {
							// kernel.cu:281-299: cosine-weighted bounce, drawn right after the cone sample as in shade();
							// the direction is kept in `bdir` until the shadow ray (if any) has been traced.
							bdir = bounce_direction(pn, sseed);
						}
						if (!cast) {
							if (terminated) { s++; pstate = P_GEN; }
							else { bounces++; ro = hitp; rd = bdir; r.n = pn; shadow = false; need_setup = true; }
						}
					}
					BM_MARK(1, t_sub); // shade, hit branch (cone sample, bounce direction)
					BM_REGION("C.sky");
					if (!is_hit || cast) {
						const SkyView sv = sky_view(fc, view);
						if (cast) {
							scolor = (sun_from_view(fc, sv) * sunLight) * 1E-5f; // kernel.cu:278
							ro = hitp; rd = view;
							shadow = true;
							need_setup = true;
							hand = HELP;
						} else {
							// ---- shade, miss branch (kernel.cu:316-323)
							f3 c;
							if (bounces == 0) c = fc.sun_angular_cos == 1.0f ? mk(1.0f, 0.0f, 0.0f) : sunsky_from_view(fc, sv);
							else c = sky_from_view(fc, sv);
							miss_color = c;
						}
					}
					if (!is_hit || primary_only) { // the path ends here
						if (!(is_hit && primary_only)) add_rgb(miss_color); // (nothing to add for a primary-only hit)
						add_terminated();
						s++;
						pstate = P_GEN;
					}
				}
				BM_MARK(2, t_sub); // sky model (+ the tail of the shade block)
			}
			BM_REGION("C.hand-over");
			if (HELP) {
				// ---- hand shadow rays to idle lanes: the k-th lane that has one writes (origin, direction, colour, pixel) to slot k of the
				// wave's part of the LDS staging area, the k-th idle lane reads slot k and becomes its helper
				const unsigned long long hand_m = __ballot(hand), free_m = __ballot(state == ST_IDLE);
				if (hand_m != 0ull && free_m != 0ull) {
					const int n_pairs = min(__popcll(hand_m), __popcll(free_m));
					const int hrank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(hand_m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(hand_m), 0u));
					const int frank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(free_m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(free_m), 0u));
					const bool gives = hand && hrank < n_pairs, takes = state == ST_IDLE && frank < n_pairs;
					if (gives) {
						const uint32_t k = static_cast<uint32_t>(hrank);
						staging_word(lds_brick, k, 0) = ro.x; staging_word(lds_brick, k, 1) = ro.y; staging_word(lds_brick, k, 2) = ro.z;
						staging_word(lds_brick, k, 3) = rd.x; staging_word(lds_brick, k, 4) = rd.y; staging_word(lds_brick, k, 5) = rd.z;
						staging_word(lds_brick, k, 6) = scolor.x; staging_word(lds_brick, k, 7) = scolor.y; staging_word(lds_brick, k, 8) = scolor.z;
						staging_word(lds_brick, k, 9) = __uint_as_float(local_pixel);
						if (DBG) staging_word(lds_brick, k, 10) = __uint_as_float((static_cast<uint32_t>(s) << 8) | static_cast<uint32_t>(bounces)); // the shadow ray's digest key
						// the owner is done with this shadow ray: what connect would have done next happens now.  (Invariants: the hand-over runs
						// BEFORE the generate block of the same pass, which either starts the owner's next primary ray or idles the lane; the owner
						// no longer has a shadow ray in flight either way -- say so; and a lane that TAKES a ray keeps pstate == P_HELPER through
						// the set-up below and through connect, so it never reaches the shade / generate / write-back code with the owner's pixel)
						shadow = false;
						if (terminated) { s++; pstate = P_GEN; need_setup = false; }
						else { bounces++; ro = hitp; rd = bdir; r.n = pn; }
					}
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
					__builtin_amdgcn_wave_barrier();
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
					if (takes) {
						const uint32_t k = static_cast<uint32_t>(frank);
						ro = mk(staging_word(lds_brick, k, 0), staging_word(lds_brick, k, 1), staging_word(lds_brick, k, 2));
						rd = mk(staging_word(lds_brick, k, 3), staging_word(lds_brick, k, 4), staging_word(lds_brick, k, 5));
						scolor = mk(staging_word(lds_brick, k, 6), staging_word(lds_brick, k, 7), staging_word(lds_brick, k, 8));
						local_pixel = __float_as_uint(staging_word(lds_brick, k, 9));
						if (DBG) ray_key = __float_as_uint(staging_word(lds_brick, k, 10));
						shadow = true;
						pstate = P_HELPER;
						need_setup = true;
					}
				}
			}
			BM_REGION("C.generate");
			if (state == ST_NEED) {
				if (pstate == P_GEN) {
					if (s >= s_end) {
						// item finished: write the accumulator back (or add this sample's share) and wait for the next one
						if (HELP) {
							// (every event has been added already)
						} else if (atomic_acc) {
							float* a = reinterpret_cast<float*>(accum + local_pixel);
							pixel_atomic_add(a + 0, acc.x); pixel_atomic_add(a + 1, acc.y); pixel_atomic_add(a + 2, acc.z); pixel_atomic_add(a + 3, acc.w);
						} else {
							pixel_store(accum + local_pixel, acc);
						}
						if (DBG && dbg && !ray_digest) { // (ray-digest records were added ray by ray)
							u32_in_global_memory* d = (u32_in_global_memory*)(dbg + static_cast<size_t>(local_pixel) * 8);
							if (sample_items) {
								// one item = one sample: the pixel's record becomes an order-independent digest -- the SUMS (mod 2^32) of
								// the per-sample path hashes, ray counts and cell counts (the caller zeroes the buffer); the first-hit
								// record is that of sample 0, written by the one item that traced it
								if (s_end == 1) { d[0] = d0; d[1] = d1; d[2] = d2; d[3] = d3; }
								record_atomic_add((uint32_t*)(d + 4), hseg); record_atomic_add((uint32_t*)(d + 5), hsh); record_atomic_add((uint32_t*)(d + 6), next | (nsh << 16));
								record_atomic_add((uint32_t*)(d + 7), tally.index_loads - loads0);
							} else {
								d[0] = d0; d[1] = d1; d[2] = d2; d[3] = d3; d[4] = hseg; d[5] = hsh; d[6] = next | (nsh << 16);
								d[7] = tally.index_loads - loads0;
							}
						}
						state = ST_IDLE;
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
				BM_MARK(3, t_sub); // pixel hand-back + primary ray
			}
			BM_REGION("C.set-up");
			if (need_setup) {
				if (shadow) r.n = mk(0.f, 0.f, 0.f); // connect passes a zeroed normal (kernel.cu:338)
				if (DBG) {
					// digest key of the ray about to start: its path's sample and segment.  An extend ray is segment `bounces` (already counted up
					// for a bounce ray), a shadow ray belongs to the segment that drew it (`bounces` is counted up after connect -- or was, by the
					// owner, when it gave the ray away: a helper got the key with the ray)
					if (!(HELP && pstate == P_HELPER)) ray_key = (static_cast<uint32_t>(s) << 8) | static_cast<uint32_t>(bounces);
					ray_loads0 = tally.index_loads;
				}
				pstate = shadow ? ((HELP && pstate == P_HELPER) ? P_HELPER : P_SHD_DONE) : P_EXT_DONE;
				const int st = ray_setup<DBG>(sc, ro, rd, r, tally);
				state = (st == ST_NEED && shadow) ? ST_CONN : st;
			}
			BM_MARK(4, t_sub); // ray set-up
		} else if (phase == 1) {
			BM_REGION("B.candidate");
			if (BM_PRIO) __builtin_amdgcn_s_setprio(BM_PRIO_B);
			if (BM_TIMED) { runsB++; lanesB += nB; }
			// ================= phase B: resolve non-empty cells (index word, LoD / 8^3 bitmask DDA, streaming request)
#ifdef BM_SHARE_PROBE
			// profiling build (-DBM_PHASE_TIMING -DBM_SHARE_PROBE, tools/share_probe.py): how often do lanes of one candidate pass test the
			// SAME brick cell -- what broadcasting one staged brick across the wavefront (the north-star's LDS sentence) could share?
			// det[5] = candidate passes, det[6] = those in which at least two lanes are in the same cell, det[7] = lanes whose cell a
			// lower lane of the pass tests as well (their brick fetch + staging is what a broadcast would save)
			{
				const bool cand = state == ST_CAND;
				const uint32_t cell_id = r.p - r.field_off;
				bool dup = false;
				for (int l = 0; l < 63; ++l) {
					const uint32_t other = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(cell_id), l));
					const bool other_cand = (__ballot(cand) >> l) & 1ull;
					dup = dup || (cand && other_cand && lane > l && other == cell_id);
				}
				const int n_dup = __popcll(__ballot(dup));
				if (__ballot(cand) != 0ull) { det[5] += 1; det[6] += n_dup ? 1 : 0; det[7] += static_cast<unsigned long long>(n_dup); }
			}
#endif
#ifdef BM_PHASE_TIMING
			uint32_t walk_trips = 0;
			if (state == ST_CAND) {
				int st = process_candidate<DBG>(sc, fc.campos, r, info, tally, lds_brick, &walk_trips);
#else
			if (state == ST_CAND) {
				int st = process_candidate<DBG>(sc, fc.campos, r, info, tally, lds_brick);
#endif
				state = (st == ST_NEED && shadow) ? ST_CONN : st;
			}
#if defined(BM_PHASE_TIMING) && !defined(BM_SHARE_PROBE)
			{ // longest 8^3 walk of this pass (= its loop length) and the lanes' own walk lengths
				uint32_t mx = walk_trips, sum = walk_trips;
				for (int off = 32; off > 0; off >>= 1) { const uint32_t o = __shfl_xor(mx, off, 64); mx = o > mx ? o : mx; sum += __shfl_xor(sum, off, 64); }
				if (mx) { det[5] += 1; det[6] += mx; det[7] += sum; }
			}
#endif
		} else {
			// ================= phase A: brick-grid walk; lanes that reach a non-empty cell or leave the grid wait.
			// Walking lanes are of two kinds: ST_JUMP lanes have an empty cube of BM_JUMP_MIN cells or more ahead and cross
			// it with one exact jump (jump.h; about four single moves' worth of instructions, ~1.5 cube edges of progress),
			// ST_OUTER lanes are close to the surface.  With enough jumpers in the wave every walking lane takes the jump
			// pass (a cube of edge 1-3 is crossed just the same, and a jump with n = 1 is exactly one move, valid in any
			// cell); otherwise the lanes near the surface make a few single moves and the jumpers wait for company.
			if (BM_PRIO) __builtin_amdgcn_s_setprio(BM_PRIO_A);
			const int nO = nA - nJ;
			BM_REGION("A.jump");
			if (nJ * BM_JUMP_RATIO >= nO) {
				int walkers = nA;
#pragma unroll 1
				for (int pass = 0; pass < BM_JUMP_PASSES; ++pass) {
					if (BM_TIMED) { runsJ++; lanesJ += walkers; }
					if (state == ST_JUMP || state == ST_OUTER) {
						int st;
						if (!(r.cube & kCubeNoJump)) st = field_jump<DBG>(sc, r, tally);
						else st = field_step<DBG>(sc, r, tally); // tmax outside the range jump.h handles (first move of a ray that starts on a cell face)
						state = (st == ST_NEED && shadow) ? ST_CONN : st;
					}
					if (BM_JUMP_PASSES > 1) { // another pass right away while most of the walkers are still walking (saves a scheduler round)
						const int still = __popcll(__ballot(state == ST_JUMP) | __ballot(state == ST_OUTER));
						if (still * BM_JUMP_KEEP_DIV < walkers * BM_JUMP_KEEP_NUM || still == 0) break;
						walkers = still;
					}
				}
			} else {
				BM_REGION("A.single");
#pragma unroll 1
				for (int k = 0; k < BM_STEPS_PER_ROUND; ++k) {
					if (BM_TIMED) { runsA++; lanesA += __popcll(__ballot(state == ST_OUTER)); }
					if (state == ST_OUTER) {
						const int st = field_step<DBG>(sc, r, tally);
						state = (st == ST_NEED && shadow) ? ST_CONN : st;
					}
				}
			}
		}
		BM_REGION("loop-end");
		if (BM_TIMED) {
			const unsigned long long dt = __builtin_amdgcn_s_memtime() - t_phase;
			if (phase == 0) cycA += dt; else if (phase == 1) cycB += dt; else cycC += dt;
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
		const unsigned long long t_end = __builtin_amdgcn_s_memtime();
		cycD = t_dry ? t_end - t_dry : 0ull; // "connect" slot: time from the wave's last (failed) refill to its exit = the drain
		const unsigned long long cy[8] = {cycA, cycB, cycC, cycD, t_end - t_begin, runsJ, lanesJ, 1ull};
		for (int k = 0; k < 8; ++k) atomicAdd(&counters->cycles[k], cy[k]);
		for (int k = 0; k < 8; ++k) atomicAdd(&counters->detail[k], det[k]);
	}
}
#undef fc

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
	const uint32_t slot = sc.pool_base[sci] + (word & kIndexBits);
	arena_rw[(static_cast<size_t>(slot) << 4) + w] = bricks_queue[(static_cast<size_t>(i) << 4) + w];
	if (w == 0) sc.index_grid[(static_cast<size_t>(sci) << 12) + local] = word; // plain store: clears unloaded + requested
}

// Pool growth (Scene.cpp:231-251): copy the resident bricks of every grown pool to its new region and publish the new
// base.  One workgroup per pool; runs on the load stream ahead of the scatter of the batch that made the pools grow.
__global__ void move_pools(const PoolMove* __restrict__ moves, uint32_t* __restrict__ arena, uint32_t* __restrict__ pool_base) {
	const PoolMove m = moves[blockIdx.x];
	const uint4* src = reinterpret_cast<const uint4*>(arena + (static_cast<size_t>(m.src) << 4));
	uint4* dst = reinterpret_cast<uint4*>(arena + (static_cast<size_t>(m.dst) << 4));
	const uint32_t n = m.src == m.dst ? 0u : m.count * 4u; // 16-byte pieces
	for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
	if (threadIdx.x == 0) pool_base[m.supercell] = m.dst;
}

// Overlapped request servicing: copy the ring's count and its first min(count, capacity) entries into the pinned host mirror,
// straight from the device (the host buffer is mapped): what travels over PCIe is what was requested, not the ring's capacity
// (the count is only known on the device at this point -- the host never waits for the frame; Scene.cpp:200-210 reads both
// with blocking copies).  One small grid on the load stream, behind the frames that may still append to the ring.
__global__ void snapshot_ring(const int* __restrict__ queue, const uint32_t* __restrict__ count, int* __restrict__ host_positions, uint32_t* __restrict__ host_count,
							  uint32_t capacity) {
	const uint32_t n = min(*count, capacity);
	const uint32_t words = n * 3u;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < words; i += gridDim.x * blockDim.x) host_positions[i] = queue[i];
	if (blockIdx.x == 0 && threadIdx.x == 0) *host_count = *count; // (the caller clamps it, like kernel.cu:409 / Scene.cpp:203)
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
// resident 256-thread workgroups per compute unit of one instantiation (asked once per instantiation)
template <bool DBG, bool XCD, bool HELP, int RING>
static int occupancy_of() {
	static const int cached = [] {
		int n = 0;
		const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trace_paths<DBG, XCD, HELP, RING>, 256, 0);
		return e == hipSuccess && n > 0 ? n : 1;
	}();
	return cached;
}
// run `f` with the four instantiation choices as compile-time constants (ring: 0 = one frame, 1 = frame ring, 2 = uniform frame ring)
template <class F>
static auto with_instantiation(bool instrumented, bool xcd, bool help, int ring, F&& f) {
	using R0 = std::integral_constant<int, 0>; using R1 = std::integral_constant<int, 1>; using R2 = std::integral_constant<int, 2>;
	auto pick = [&](auto d, auto x, auto h) { return ring == 2 ? f(d, x, h, R2{}) : (ring == 1 ? f(d, x, h, R1{}) : f(d, x, h, R0{})); };
	auto pick_h = [&](auto d, auto x) { return help ? pick(d, x, std::true_type{}) : pick(d, x, std::false_type{}); };
	auto pick_x = [&](auto d) { return xcd ? pick_h(d, std::true_type{}) : pick_h(d, std::false_type{}); };
	return instrumented ? pick_x(std::true_type{}) : pick_x(std::false_type{});
}
int trace_blocks_per_cu(bool instrumented, bool xcd, bool help, int ring) {
	return with_instantiation(instrumented, xcd, help, ring, [](auto d, auto x, auto h, auto r) { return occupancy_of<decltype(d)::value, decltype(x)::value, decltype(h)::value, decltype(r)::value>(); });
}

// Persistent launch: exactly as many 256-thread workgroups as the device keeps resident (compute_units x
// blocks per CU); the waves pull 4x4-pixel chunks from the ticket counters behind work_counter -- one block of kWorkCounterBytes
// per frame of the launch (fc_dev[0], fc_dev[1], ... up to the entry with frames_after == 0), all zero at launch.  fc = the host
// copy of the first frame's constants (fc_dev[0]).
void launch_trace(const DeviceScene& sc, const FrameConstants& fc, const FrameConstants* fc_dev, DeviceCounters* counters,
				  uint32_t* work_counter, bool instrumented, int compute_units, int blocks_per_cu_cap, hipStream_t stream) {
	const long long chunks = static_cast<long long>(fc.tiles_x) * fc.tiles_y * 16;
	if (chunks <= 0) return;
	const bool xcd = fc.xcd_handout != 0, help = fc.helpers != 0;
	const int ring = fc.frames_after > 0 ? (fc.ring_uniform ? 2 : 1) : 0;
	int per_cu = trace_blocks_per_cu(instrumented, xcd, help, ring); // what THIS instantiation keeps resident
	if (blocks_per_cu_cap > 0 && per_cu > blocks_per_cu_cap) per_cu = blocks_per_cu_cap;
	const long long resident_blocks = static_cast<long long>(compute_units) * per_cu;
	// never more waves than a frame has 64-item groups: an item is a pixel, or ONE sample of a pixel with (chunk, sample) items -- a
	// 1/8 shard of a 1080p frame at 8 spp is 276 480 pixels but 2.2 M items, and sizing its launch by pixels left 40 % of the
	// GPU's wave slots empty (1080 of 1792 workgroups: 1.16 -> 0.95 ms per shard step).  A launch of several frames may start a
	// second frame's worth of waves: those that find the first frame's counters used up go straight on to the next one.
	const long long items = chunks * 16 * ((fc.flags & 4u /* BM_FLAG_SAMPLE_ITEMS */) ? (fc.spp > 0 ? fc.spp : 1) : 1);
	long long blocks = (items + 255) / 256;
	if (ring) blocks *= 2;
	if (blocks > resident_blocks) blocks = resident_blocks;
	const dim3 grid(static_cast<unsigned>(blocks)), block(256);
#ifdef BM_PHASE_TIMING
	DeviceCounters* const plain_counters = counters; // profiling build: the plain kernel reports its phase timers too
#else
	DeviceCounters* const plain_counters = nullptr;
#endif
	DeviceCounters* const cnt = instrumented ? counters : plain_counters;
	with_instantiation(instrumented, xcd, help, ring, [&](auto d, auto x, auto h, auto r) {
		hipLaunchKernelGGL((trace_paths<decltype(d)::value, decltype(x)::value, decltype(h)::value, decltype(r)::value>), grid, block, 0, stream, sc, fc_dev, cnt, work_counter);
		return 0;
	});
}

void launch_upload(const DeviceScene& sc, const uint32_t* bricks_queue, const uint32_t* indices_queue, uint32_t* arena, uint32_t count,
				   hipStream_t stream) {
	if (count == 0) return;
	const int per_block = 256 / 16;
	hipLaunchKernelGGL(upload_bricks, dim3((count + per_block - 1) / per_block), dim3(256), 0, stream, sc, bricks_queue, indices_queue, arena, count);
}

void launch_pool_moves(const PoolMove* moves, uint32_t count, uint32_t* arena, uint32_t* pool_base, hipStream_t stream) {
	if (count == 0) return;
	hipLaunchKernelGGL(move_pools, dim3(count), dim3(256), 0, stream, moves, arena, pool_base);
}

void launch_snapshot_ring(const int* queue, const uint32_t* count, int* host_positions, uint32_t* host_count, uint32_t capacity, hipStream_t stream) {
	hipLaunchKernelGGL(snapshot_ring, dim3(32), dim3(256), 0, stream, queue, count, host_positions, host_count, capacity);
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
