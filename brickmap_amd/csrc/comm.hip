// comm.hip -- the multi-GPU exchange of the path-trace job behind the C-ABI (include/brickmap.h "multi-GPU").
//
// The reference is single-GPU (no collective anywhere; src/main.cpp:89 computes `multi_gpu` and never uses it).
// Here every GPU holds a full scene replica and renders the interleaved row bands of the frame that belong to its rank
// (bm_frame_params band_rows / shard_rank / shard_count) into a PACKED local buffer; one exchange per frame brings the bands
// to the root:
//     ncclGroupStart;  root: ncclRecv from every peer straight into one stacked buffer;  peers: ncclSend;  ncclGroupEnd
//     + one kernel on the root that assembles the frame: row y <- (its rank, its packed row)
// xGMI is point-to-point, so every peer's band travels on its own link into the root (4K float4: 132.7 MB / 8 = 16.6 MB per
// peer, ~0.11 ms at ~153 GB/s); no all-reduce, no ring.  Everything is enqueued on the caller's stream: the gather is ordered
// behind the frame that produced the bands and ahead of whatever the caller queues next.
//
// RCCL is bound at run time (dlopen of librccl.so.1 on the first bm_comm_* call), not at link time: PyTorch ships its own
// librccl.so under the same SONAME, and a hard DT_NEEDED here would make whichever of the two libraries is loaded first the
// RCCL of the whole process.  Bound lazily, a process that already holds an RCCL (torch.distributed, or a C++ host linked
// against /opt/rocm/lib/librccl.so) shares it, and a single-GPU process never loads one.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/brickmap.h"
#include "scene.h"

namespace bm {
namespace {

struct Rccl {
	void* lib = nullptr;
	decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
	decltype(&ncclCommInitRank) CommInitRank = nullptr;
	decltype(&ncclCommDestroy) CommDestroy = nullptr;
	decltype(&ncclGroupStart) GroupStart = nullptr;
	decltype(&ncclGroupEnd) GroupEnd = nullptr;
	decltype(&ncclSend) Send = nullptr;
	decltype(&ncclRecv) Recv = nullptr;
	decltype(&ncclReduce) Reduce = nullptr;
	decltype(&ncclAllReduce) AllReduce = nullptr;
	decltype(&ncclGetErrorString) GetErrorString = nullptr;
	decltype(&ncclCommCount) CommCount = nullptr;       // optional: bm_comm_info asks the library itself where it can
	decltype(&ncclCommUserRank) CommUserRank = nullptr; // optional
};

Rccl bind_rccl() {
	Rccl api;
	// BM_RCCL_LIBRARY names a specific build of the library (a site's own RCCL; the tests' host-staged stand-in that lets
	// several ranks share one GPU, tests/fake_rccl.cpp)
	const char* names[] = {std::getenv("BM_RCCL_LIBRARY"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
	for (const char* n : names) {
		if (!n || !*n) continue;
		api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
		if (api.lib) break;
	}
	if (api.lib) {
#define BM_SYM(name) api.name = reinterpret_cast<decltype(api.name)>(dlsym(api.lib, "nccl" #name))
		BM_SYM(GetUniqueId); BM_SYM(CommInitRank); BM_SYM(CommDestroy); BM_SYM(GroupStart); BM_SYM(GroupEnd);
		BM_SYM(Send); BM_SYM(Recv); BM_SYM(Reduce); BM_SYM(AllReduce); BM_SYM(GetErrorString); BM_SYM(CommCount); BM_SYM(CommUserRank);
#undef BM_SYM
		if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.GroupStart || !api.GroupEnd || !api.Send || !api.Recv || !api.Reduce ||
			!api.AllReduce || !api.GetErrorString) {
			dlclose(api.lib);
			api.lib = nullptr;
		}
	}
	return api;
}
Rccl* rccl() {
	static Rccl api = bind_rccl(); // (a function-local static: bound once, also when several host threads make communicators at the same time)
	return api.lib ? &api : nullptr;
}

} // namespace
int nccl_fail(ncclResult_t r, const char* what) {
	Rccl* R = rccl();
	set_error(std::string("RCCL: ") + what + ": " + (R ? R->GetErrorString(r) : "library not loaded"));
	return 20000 + static_cast<int>(r);
}
#define BM_NCCL(expr)                                          \
	do {                                                       \
		const ncclResult_t bm_r_ = (expr);                     \
		if (bm_r_ != ncclSuccess) return ::bm::nccl_fail(bm_r_, #expr); \
	} while (0)

namespace {
// rows of the frame that belong to `rank` (bm_local_rows for that shard)
int shard_rows(int height, int band_rows, int rank, int world) {
	int n = 0;
	for (int y0 = 0; y0 < height; y0 += band_rows)
		if ((y0 / band_rows) % world == rank) n += (height - y0 < band_rows) ? height - y0 : band_rows;
	return n;
}

// frame row y <- packed row ((y / band) / world) * band + y % band of rank (y / band) % world; the root's own rows come from
// its packed buffer, everybody else's from the stacked receive buffer.  blockIdx.y = frame of the batch (bm_gather_frames): every
// rank's packed buffer holds its rows of frame 0, then of frame 1, ... (rows_r x width each), and the stacked buffer holds rank r's
// whole batch at r * count * max_rows * width -- as it was sent, tightly packed.
__device__ __forceinline__ int rows_of_rank(int height, int band_rows, int rank, int world) { // bm_local_rows in closed form
	const int full_bands = height / band_rows, tail = height % band_rows;
	int rows = rank < full_bands ? ((full_bands - 1 - rank) / world + 1) * band_rows : 0;
	if (tail && full_bands % world == rank) rows += tail;
	return rows;
}
__global__ void assemble_frame(const float4* __restrict__ own, const float4* __restrict__ stacked, float4* __restrict__ frame, int height, int width, int band_rows,
							   int world, int me, int max_rows, int count) {
	const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
	if (i >= static_cast<long long>(height) * width) return;
	const int k = static_cast<int>(blockIdx.y);
	const int y = static_cast<int>(i / width), x = static_cast<int>(i - static_cast<long long>(y) * width);
	const int band = y / band_rows, r = band % world, lrow = (band / world) * band_rows + y % band_rows;
	const size_t rows_r = static_cast<size_t>(rows_of_rank(height, band_rows, r, world));
	const float4* src = (r == me ? own : stacked + static_cast<size_t>(r) * count * max_rows * width) + static_cast<size_t>(k) * rows_r * width;
	frame[static_cast<size_t>(k) * height * width + i] = src[static_cast<size_t>(lrow) * width + x];
}

// bm_probe_streams: one wave that does nothing for `ticks` of the constant 100 MHz clock
__global__ void spin_kernel(long long ticks) {
	const long long t0 = wall_clock64();
	while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

} // namespace
} // namespace bm

struct bm_comm {
	int device = 0, rank = 0, world = 1;
	ncclComm_t comm = nullptr;
	float* stacked = nullptr; // root: world x max_rows x width float4, grown on demand
	size_t stacked_bytes = 0;
	int* word = nullptr;      // barrier / self-test scratch (4 ints)
};

using bm::set_error;

extern "C" {

int bm_comm_unique_id(void* id128) {
	if (!id128) { set_error("null argument"); return BM_EINVAL; }
	bm::Rccl* R = bm::rccl();
	if (!R) { set_error("RCCL library (librccl.so.1) not found"); return BM_ESTATE; }
	static_assert(sizeof(ncclUniqueId) == BM_COMM_ID_BYTES, "bm_comm id size");
	ncclUniqueId id;
	BM_NCCL(R->GetUniqueId(&id));
	std::memcpy(id128, &id, sizeof id);
	return 0;
}

int bm_comm_create(int device, int rank, int world, const void* id128, bm_comm** out) {
	if (!out || !id128 || world < 1 || rank < 0 || rank >= world) { set_error("bad argument"); return BM_EINVAL; }
	*out = nullptr;
	bm::Rccl* R = bm::rccl();
	if (!R) { set_error("RCCL library (librccl.so.1) not found"); return BM_ESTATE; }
	BM_HIP(hipSetDevice(device));
	bm_comm* c = new (std::nothrow) bm_comm;
	if (!c) { set_error("out of host memory"); return BM_EINVAL; }
	c->device = device; c->rank = rank; c->world = world;
	ncclUniqueId id;
	std::memcpy(&id, id128, sizeof id);
	const ncclResult_t r = R->CommInitRank(&c->comm, world, id, rank);
	if (r != ncclSuccess) { delete c; return bm::nccl_fail(r, "ncclCommInitRank"); }
	if (hipMalloc(reinterpret_cast<void**>(&c->word), 4 * sizeof(int)) != hipSuccess) {
		R->CommDestroy(c->comm);
		delete c;
		set_error("out of device memory");
		return BM_EINVAL;
	}
	*out = c;
	return 0;
}

void bm_comm_destroy(bm_comm* c) {
	if (!c) return;
	(void)hipSetDevice(c->device);
	(void)hipDeviceSynchronize();
	if (bm::Rccl* R = bm::rccl()) if (c->comm) R->CommDestroy(c->comm);
	if (c->stacked) (void)hipFree(c->stacked);
	if (c->word) (void)hipFree(c->word);
	delete c;
}

int bm_comm_available(void) { return bm::rccl() ? 1 : 0; }

int bm_comm_info(bm_comm* c, int* rank, int* world) {
	if (!c) { set_error("null communicator"); return BM_EINVAL; }
	// what the LIBRARY says about the communicator (ncclCommUserRank / ncclCommCount), so that "did RCCL see N ranks" has an answer
	// that does not come from the caller's own arguments; a transport without those two entry points reports what it was created with
	bm::Rccl* R = bm::rccl();
	int r = c->rank, w = c->world;
	if (R && R->CommCount && R->CommUserRank && c->comm) {
		BM_NCCL(R->CommUserRank(c->comm, &r));
		BM_NCCL(R->CommCount(c->comm, &w));
	}
	if (rank) *rank = r;
	if (world) *world = w;
	return 0;
}

int bm_gather_frame(bm_comm* c, const float* packed_dev, float* frame_dev, int height, int width, int band_rows, int root, void* hip_stream) {
	return bm_gather_frames(c, packed_dev, frame_dev, 1, height, width, band_rows, root, hip_stream);
}

int bm_gather_frames(bm_comm* c, const float* packed_dev, float* frames_dev, int count, int height, int width, int band_rows, int root, void* hip_stream) {
	if (!c) { set_error("null communicator"); return BM_EINVAL; }
	if (!packed_dev || count < 1 || count > 256 || height <= 0 || width <= 0 || band_rows <= 0 || root < 0 || root >= c->world) { set_error("bad argument"); return BM_EINVAL; }
	if (c->rank == root && !frames_dev) { set_error("the root needs a frame buffer"); return BM_EINVAL; }
	bm::Rccl* R = bm::rccl();
	hipStream_t stream = static_cast<hipStream_t>(hip_stream);
	BM_HIP(hipSetDevice(c->device));
	int max_rows = 0;
	for (int r = 0; r < c->world; ++r) { const int n = bm::shard_rows(height, band_rows, r, c->world); if (n > max_rows) max_rows = n; }
	const size_t row_floats = static_cast<size_t>(width) * 4;
	const size_t slot_floats = static_cast<size_t>(count) * max_rows * row_floats; // one rank's batch in the stacked buffer
	if (c->rank == root && c->world > 1) {
		const size_t need = static_cast<size_t>(c->world) * slot_floats * sizeof(float);
		if (c->stacked_bytes < need) {
			if (c->stacked) { BM_HIP(hipStreamSynchronize(stream)); BM_HIP(hipFree(c->stacked)); c->stacked = nullptr; c->stacked_bytes = 0; }
			BM_HIP(hipMalloc(reinterpret_cast<void**>(&c->stacked), need));
			c->stacked_bytes = need;
		}
	}
	if (c->world > 1) {
		// ONE message per peer for the whole batch (its rows of frame 0, of frame 1, ...: the packed buffers of a batch are one
		// allocation), all of them in one group: every peer's bands travel on its own xGMI link into the root.
		// (an error inside the group must not leave RCCL in group mode: the group is always closed, the first error is reported)
		BM_NCCL(R->GroupStart());
		ncclResult_t first = ncclSuccess;
		if (c->rank == root) {
			for (int r = 0; r < c->world && first == ncclSuccess; ++r) {
				if (r == root) continue;
				const size_t n = static_cast<size_t>(count) * bm::shard_rows(height, band_rows, r, c->world) * row_floats;
				if (n) first = R->Recv(c->stacked + static_cast<size_t>(r) * slot_floats, n, ncclFloat, r, c->comm, stream);
			}
		} else {
			const size_t n = static_cast<size_t>(count) * bm::shard_rows(height, band_rows, c->rank, c->world) * row_floats;
			if (n) first = R->Send(packed_dev, n, ncclFloat, root, c->comm, stream);
		}
		const ncclResult_t closed = R->GroupEnd();
		if (first != ncclSuccess) return bm::nccl_fail(first, "ncclSend / ncclRecv of the packed row bands");
		if (closed != ncclSuccess) return bm::nccl_fail(closed, "ncclGroupEnd");
	}
	if (c->rank == root) {
		const long long n = static_cast<long long>(height) * width;
		hipLaunchKernelGGL(bm::assemble_frame, dim3(static_cast<unsigned>((n + 255) / 256), static_cast<unsigned>(count)), dim3(256), 0, stream,
						   reinterpret_cast<const float4*>(packed_dev), reinterpret_cast<const float4*>(c->stacked), reinterpret_cast<float4*>(frames_dev), height, width,
						   band_rows, c->world, c->rank, max_rows, count);
		BM_HIP(hipGetLastError());
	}
	return 0;
}

int bm_reduce_frame(bm_comm* c, const float* in_dev, float* out_dev, int64_t n_floats, int root, void* hip_stream) {
	if (!c) { set_error("null communicator"); return BM_EINVAL; }
	if (!in_dev || n_floats < 0 || root < 0 || root >= c->world || (c->rank == root && !out_dev)) { set_error("bad argument"); return BM_EINVAL; }
	bm::Rccl* R = bm::rccl();
	BM_HIP(hipSetDevice(c->device));
	BM_NCCL(R->Reduce(in_dev, out_dev, static_cast<size_t>(n_floats), ncclFloat, ncclSum, root, c->comm, static_cast<hipStream_t>(hip_stream)));
	return 0;
}

int bm_comm_barrier(bm_comm* c, void* hip_stream) {
	if (!c) { set_error("null communicator"); return BM_EINVAL; }
	bm::Rccl* R = bm::rccl();
	hipStream_t stream = static_cast<hipStream_t>(hip_stream);
	BM_HIP(hipSetDevice(c->device));
	BM_HIP(hipMemsetAsync(c->word, 0, sizeof(int), stream));
	BM_NCCL(R->AllReduce(c->word, c->word, 1, ncclInt, ncclSum, c->comm, stream));
	BM_HIP(hipStreamSynchronize(stream));
	return 0;
}

int bm_probe_streams(int device, int count, void** streams_out) {
	if (!streams_out || count < 1 || count > 4) { set_error("bad argument"); return BM_EINVAL; }
	BM_HIP(hipSetDevice(device));
	const int pool_n = count + 3;
	hipStream_t pool[8] = {};
	for (int i = 0; i < pool_n; ++i) {
		if (hipStreamCreateWithFlags(&pool[i], hipStreamNonBlocking) != hipSuccess) {
			for (int k = 0; k < i; ++k) (void)hipStreamDestroy(pool[k]);
			set_error("hipStreamCreate failed");
			return BM_ESTATE;
		}
	}
	// every combination of `count` candidates: how long do `count` spin kernels (one wave each, ~0.2 ms) take when started together?
	double best_s = 0.0;
	unsigned best_mask = 0u;
	for (unsigned mask = 0; mask < (1u << pool_n); ++mask) {
		if (__builtin_popcount(mask) != count) continue;
		double fastest = 0.0;
		for (int rep = 0; rep < 3; ++rep) {
			(void)hipDeviceSynchronize();
			const auto t0 = std::chrono::steady_clock::now();
			for (int i = 0; i < pool_n; ++i)
				if (mask & (1u << i)) hipLaunchKernelGGL(bm::spin_kernel, dim3(1), dim3(64), 0, pool[i], 20000ll); // 100 MHz ticks
			(void)hipDeviceSynchronize();
			const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
			if (rep == 0 || s < fastest) fastest = s;
		}
		if (best_mask == 0u || fastest < best_s * 0.9) { best_s = fastest; best_mask = mask; } // (a later combination must be clearly better)
	}
	int n = 0;
	for (int i = 0; i < pool_n; ++i) {
		if (best_mask & (1u << i)) streams_out[n++] = pool[i];
		else (void)hipStreamDestroy(pool[i]);
	}
	BM_HIP(hipGetLastError());
	return 0;
}

void bm_release_streams(int count, void** streams) {
	if (!streams) return;
	for (int i = 0; i < count; ++i)
		if (streams[i]) { (void)hipStreamDestroy(static_cast<hipStream_t>(streams[i])); streams[i] = nullptr; }
}

int bm_debug_assemble_frame(int device, const float* own_packed_dev, const float* stacked_dev, float* frame_dev, int height, int width, int band_rows, int world,
							int me, int max_rows, void* hip_stream) {
	if (!own_packed_dev || !frame_dev || height <= 0 || width <= 0 || band_rows <= 0 || world < 1 || me < 0 || me >= world || (world > 1 && !stacked_dev)) {
		set_error("bad argument");
		return BM_EINVAL;
	}
	BM_HIP(hipSetDevice(device));
	const long long n = static_cast<long long>(height) * width;
	hipLaunchKernelGGL(bm::assemble_frame, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(hip_stream),
					   reinterpret_cast<const float4*>(own_packed_dev), reinterpret_cast<const float4*>(stacked_dev), reinterpret_cast<float4*>(frame_dev), height, width,
					   band_rows, world, me, max_rows, 1);
	BM_HIP(hipGetLastError());
	return 0;
}

int bm_comm_selftest(bm_comm* c, void* hip_stream) {
	if (!c) { set_error("null communicator"); return BM_EINVAL; }
	bm::Rccl* R = bm::rccl();
	hipStream_t stream = static_cast<hipStream_t>(hip_stream);
	BM_HIP(hipSetDevice(c->device));
	// (a) a grouped send / receive round the ring of ranks (to itself when there is one rank): the transport bm_gather_frame uses
	const int next = (c->rank + 1) % c->world, prev = (c->rank + c->world - 1) % c->world;
	const int token = 0x5EED0000 + c->rank;
	BM_HIP(hipMemcpyAsync(c->word, &token, sizeof(int), hipMemcpyHostToDevice, stream));
	BM_HIP(hipMemsetAsync(c->word + 1, 0, sizeof(int), stream));
	BM_NCCL(R->GroupStart());
	ncclResult_t first = R->Send(c->word, 1, ncclInt, next, c->comm, stream);
	if (first == ncclSuccess) first = R->Recv(c->word + 1, 1, ncclInt, prev, c->comm, stream);
	const ncclResult_t closed = R->GroupEnd();
	if (first != ncclSuccess) return bm::nccl_fail(first, "ncclSend / ncclRecv round the ring");
	if (closed != ncclSuccess) return bm::nccl_fail(closed, "ncclGroupEnd");
	// (b) a sum over all ranks
	const int one = 1;
	BM_HIP(hipMemcpyAsync(c->word + 2, &one, sizeof(int), hipMemcpyHostToDevice, stream));
	BM_NCCL(R->AllReduce(c->word + 2, c->word + 3, 1, ncclInt, ncclSum, c->comm, stream));
	int got[4] = {0, 0, 0, 0};
	BM_HIP(hipMemcpyAsync(got, c->word, sizeof got, hipMemcpyDeviceToHost, stream));
	BM_HIP(hipStreamSynchronize(stream));
	if (got[1] != 0x5EED0000 + prev || got[3] != c->world) {
		set_error("RCCL self-test: wrong data (send/recv " + std::to_string(got[1]) + ", all-reduce " + std::to_string(got[3]) + ")");
		return BM_ESTATE;
	}
	return 0;
}

} // extern "C"
