// fake_rccl.cpp -- TEST INFRASTRUCTURE: a stand-in for the ten RCCL entry points csrc/comm.hip binds at run time, so that the
// multi-rank code paths of the C-ABI exchange (bm_comm_create / bm_gather_frame / bm_reduce_frame / bm_comm_barrier /
// bm_comm_selftest: per-rank counts, offsets into the root's stacked buffer, roots other than 0, ragged shards) can run with
// several ranks ON ONE GPU -- real RCCL refuses two ranks on one device ("Duplicate GPU detected").  Transport: a POSIX
// shared-memory segment with one mailbox per (source, destination) pair, staged through the host; every operation completes
// before the call returns (the stream is synchronised first), which is a legal, if slow, implementation of the API.
// What it does NOT test is RCCL itself: grouping semantics and the xGMI transport stay unexercised without a multi-GPU node.
// Built by tests/test_gpu_dist.py into a temporary directory and selected with BM_RCCL_LIBRARY; never shipped, never linked.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

extern "C" {
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclInt = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
typedef struct { char internal[128]; } ncclUniqueId;
struct FakeComm;
typedef FakeComm* ncclComm_t;
}

namespace {
constexpr int kMaxRanks = 4;
constexpr size_t kBoxBytes = 1u << 20; // payload of one mailbox (16 MiB of shared memory in all: test frames are small)
struct Mailbox {
	std::atomic<uint64_t> sent, taken; // messages written by the source / consumed by the destination
	uint64_t bytes;
	unsigned char data[kBoxBytes];
};
struct Segment {
	Mailbox box[kMaxRanks][kMaxRanks]; // [source][destination]
};
struct Op { bool send; void* ptr; size_t bytes; int peer; hipStream_t stream; };
size_t type_size(ncclDataType_t t) { return t == ncclFloat || t == ncclInt32 || t == ncclUint32 ? 4 : (t == ncclInt8 || t == ncclUint8 ? 1 : 8); }
bool wait_until(std::atomic<uint64_t>& a, uint64_t v) {
	const auto t0 = std::chrono::steady_clock::now();
	while (a.load(std::memory_order_acquire) < v) {
		if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) return false;
		std::this_thread::sleep_for(std::chrono::microseconds(50));
	}
	return true;
}
} // namespace

struct FakeComm {
	int rank = 0, world = 1;
	Segment* seg = nullptr;
	char name[64] = {0};
	int depth = 0;
	std::vector<Op> ops;
};

namespace {
ncclResult_t run(FakeComm* c, std::vector<Op>& ops) {
	std::vector<unsigned char> host;
	// A message larger than a mailbox travels in pieces of kBoxBytes, each with the sent / taken handshake (both sides cut it the same
	// way).  Sends run first, so within one group a pair of ranks must not exchange multi-piece messages in BOTH directions (the
	// product's gather does not: peers send, the root receives; the self-test's ring carries four bytes).
	for (const Op& o : ops) { // sends first: they only need the mailbox to be free
		if (!o.send) continue;
		Mailbox& b = c->seg->box[c->rank][o.peer];
		if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
		size_t done = 0;
		do {
			const size_t piece = std::min(kBoxBytes, o.bytes - done);
			if (!wait_until(b.taken, b.sent.load())) return ncclSystemError; // previous piece consumed
			if (piece && hipMemcpy(b.data, static_cast<const unsigned char*>(o.ptr) + done, piece, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
			b.bytes = o.bytes; // the whole message's size: what the receiver checks
			b.sent.fetch_add(1, std::memory_order_release);
			done += piece;
		} while (done < o.bytes);
	}
	for (const Op& o : ops) {
		if (o.send) continue;
		Mailbox& b = c->seg->box[o.peer][c->rank];
		if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
		size_t done = 0;
		do {
			const size_t piece = std::min(kBoxBytes, o.bytes - done);
			if (!wait_until(b.sent, b.taken.load() + 1)) return ncclSystemError;
			if (b.bytes != o.bytes) { std::fprintf(stderr, "fake_rccl: rank %d expected %zu bytes from %d, got %llu\n", c->rank, o.bytes, o.peer, (unsigned long long)b.bytes); return ncclInvalidArgument; }
			if (piece && hipMemcpy(static_cast<unsigned char*>(o.ptr) + done, b.data, piece, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
			b.taken.fetch_add(1, std::memory_order_release);
			done += piece;
		} while (done < o.bytes);
	}
	ops.clear();
	return ncclSuccess;
}
ncclResult_t queue(FakeComm* c, Op o) {
	c->ops.push_back(o);
	return c->depth > 0 ? ncclSuccess : run(c, c->ops);
}
FakeComm* g_group_comm = nullptr; // the communicator the open group belongs to (one communicator per process in the tests)
int g_depth = 0;
} // namespace

extern "C" {
#define FAKE_API __attribute__((visibility("default")))
FAKE_API ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
	std::memset(id, 0, sizeof *id);
	std::snprintf(id->internal, sizeof id->internal, "/bm_fake_rccl_%d_%lld", (int)getpid(), (long long)std::chrono::steady_clock::now().time_since_epoch().count());
	return ncclSuccess;
}
FAKE_API ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
	if (nranks > kMaxRanks) return ncclInvalidArgument;
	FakeComm* c = new FakeComm;
	c->rank = rank; c->world = nranks;
	std::strncpy(c->name, id.internal, sizeof c->name - 1);
	int fd = -1;
	for (int tries = 0; tries < 2000 && fd < 0; ++tries) {
		fd = rank == 0 ? shm_open(c->name, O_CREAT | O_RDWR, 0600) : shm_open(c->name, O_RDWR, 0600);
		if (fd < 0) std::this_thread::sleep_for(std::chrono::milliseconds(5));
	}
	if (fd < 0) { delete c; return ncclSystemError; }
	if (rank == 0 && ftruncate(fd, sizeof(Segment)) != 0) { close(fd); delete c; return ncclSystemError; }
	if (rank != 0) { // wait for rank 0 to size the segment (it is zero-filled: all counters start at 0)
		for (int tries = 0; tries < 2000; ++tries) { off_t n = lseek(fd, 0, SEEK_END); if (n >= (off_t)sizeof(Segment)) break; std::this_thread::sleep_for(std::chrono::milliseconds(5)); }
	}
	void* p = mmap(nullptr, sizeof(Segment), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
	close(fd);
	if (p == MAP_FAILED) { delete c; return ncclSystemError; }
	c->seg = static_cast<Segment*>(p);
	*comm = c;
	return ncclSuccess;
}
FAKE_API ncclResult_t ncclCommDestroy(ncclComm_t c) {
	if (!c) return ncclSuccess;
	if (c->seg) munmap(c->seg, sizeof(Segment));
	if (c->rank == 0) shm_unlink(c->name);
	delete c;
	return ncclSuccess;
}
FAKE_API ncclResult_t ncclGroupStart() { g_depth++; return ncclSuccess; }
FAKE_API ncclResult_t ncclGroupEnd() {
	if (--g_depth > 0 || !g_group_comm) return ncclSuccess;
	FakeComm* c = g_group_comm;
	g_group_comm = nullptr;
	c->depth = 0;
	return run(c, c->ops);
}
FAKE_API ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s) {
	if (g_depth > 0) { g_group_comm = c; c->depth = g_depth; }
	return queue(c, Op{true, const_cast<void*>(buf), count * type_size(t), peer, s});
}
FAKE_API ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s) {
	if (g_depth > 0) { g_group_comm = c; c->depth = g_depth; }
	return queue(c, Op{false, buf, count * type_size(t), peer, s});
}
// sum-reductions (float / int32) through rank `root`: everybody sends, the root adds on the host and (all-reduce) sends back
static ncclResult_t reduce_impl(const void* in, void* out, size_t count, ncclDataType_t t, int root, ncclComm_t c, hipStream_t s, bool all) {
	const size_t bytes = count * type_size(t);
	if (t != ncclFloat && t != ncclInt32) return ncclInvalidArgument;
	if (c->rank != root) {
		std::vector<Op> ops{Op{true, const_cast<void*>(in), bytes, root, s}};
		if (ncclResult_t r = run(c, ops)) return r;
		if (all) { std::vector<Op> back{Op{false, out, bytes, root, s}}; return run(c, back); }
		return ncclSuccess;
	}
	std::vector<unsigned char> acc(bytes), tmp(bytes);
	if (hipStreamSynchronize(s) != hipSuccess || hipMemcpy(acc.data(), in, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
	void* scratch = nullptr;
	if (hipMalloc(&scratch, bytes ? bytes : 4) != hipSuccess) return ncclUnhandledCudaError;
	for (int r = 0; r < c->world; ++r) {
		if (r == root) continue;
		std::vector<Op> ops{Op{false, scratch, bytes, r, s}};
		if (ncclResult_t e = run(c, ops)) { hipFree(scratch); return e; }
		hipMemcpy(tmp.data(), scratch, bytes, hipMemcpyDeviceToHost);
		for (size_t i = 0; i < count; ++i) {
			if (t == ncclFloat) reinterpret_cast<float*>(acc.data())[i] += reinterpret_cast<float*>(tmp.data())[i];
			else reinterpret_cast<int*>(acc.data())[i] += reinterpret_cast<int*>(tmp.data())[i];
		}
	}
	hipMemcpy(out, acc.data(), bytes, hipMemcpyHostToDevice);
	if (all) {
		hipMemcpy(scratch, acc.data(), bytes, hipMemcpyHostToDevice);
		for (int r = 0; r < c->world; ++r) {
			if (r == root) continue;
			std::vector<Op> ops{Op{true, scratch, bytes, r, s}};
			if (ncclResult_t e = run(c, ops)) { hipFree(scratch); return e; }
		}
	}
	hipFree(scratch);
	return ncclSuccess;
}
FAKE_API ncclResult_t ncclReduce(const void* in, void* out, size_t count, ncclDataType_t t, ncclRedOp_t, int root, ncclComm_t c, hipStream_t s) {
	return reduce_impl(in, out, count, t, root, c, s, false);
}
FAKE_API ncclResult_t ncclAllReduce(const void* in, void* out, size_t count, ncclDataType_t t, ncclRedOp_t, ncclComm_t c, hipStream_t s) {
	return reduce_impl(in, out, count, t, 0, c, s, true);
}
FAKE_API const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake_rccl error"; }
}
