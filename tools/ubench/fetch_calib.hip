// fetch_calib.hip -- calibrates rocprofv3's memory-side counters (FETCH_SIZE, TCC_MISS, TCC_EA0_RDREQ*) on the access
// patterns of the brickmap walk: known unique bytes per dispatch, so that "HBM bytes by counters" has a measured factor.
//   stream16        16 B per lane, fully coalesced (the guide's control: FETCH_SIZE reports half of these bytes)
//   gather_b1_l128  ONE byte per lane, every lane in a different 128-byte line, each line of the buffer touched once
//   gather_b1_l64   ONE byte per lane, every lane in a different 64-byte half line, each half touched once
//   gather_b4_l128  one dword per lane (an index word), every lane in a different 128-byte line
//   gather_rec64    one 64-byte record per lane (a brick: four dwordx4 by one lane), every record touched once
//   rows_b1         one byte per lane, consecutive lanes 1 byte apart in rows of 64 (a coherent wave reading the cube field)
// over buffers of 64 MiB (fits the 256 MiB Infinity Cache; the second launch of each kernel finds it warm), 1 GiB and
// 8 GiB.  Prints one line per dispatch in launch order; tools/fetch_calib.py joins them with the counter CSVs.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/fetch_calib.hip -o scratch/fetch_calib
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// bijection on [0, n), n a power of two: odd multiplier + offset, then an xor-shift scramble that is its own inverse domain
__device__ __forceinline__ uint64_t perm(uint64_t i, uint64_t n) {
	uint64_t x = (i * 0x9E3779B97F4A7C15ull + 0x7F4A7C15ull) & (n - 1);
	x ^= x >> 13; // bijective on n-bit values (upper bits only flow down)
	x = (x * 0xD6E8FEB86659FD93ull) & (n - 1);
	x ^= x >> 11;
	return x & (n - 1);
}

__global__ void stream16(const uint4* __restrict__ p, uint64_t n16, uint32_t* sink) {
	uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
	const uint64_t stride = uint64_t(gridDim.x) * blockDim.x;
	uint32_t acc = 0;
	for (; i < n16; i += stride) { const uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
	if (acc == 0x12345678u) *sink = acc;
}
template <int GRAN, int BYTES>
__global__ void gather(const uint8_t* __restrict__ p, uint64_t units, uint32_t* sink) {
	uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
	const uint64_t stride = uint64_t(gridDim.x) * blockDim.x;
	uint32_t acc = 0;
	for (; i < units; i += stride) {
		const uint64_t u = perm(i, units);
		const uint32_t within = (uint32_t(u * 2654435761u) >> 8) & (GRAN - 1) & ~(BYTES - 1);
		const uint8_t* a = p + u * GRAN + within;
		if (BYTES == 1) acc += *a;
		else acc += *reinterpret_cast<const uint32_t*>(a);
	}
	if (acc == 0x12345678u) *sink = acc;
}
__global__ void gather_rec64(const uint8_t* __restrict__ p, uint64_t units, uint32_t* sink) {
	uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
	const uint64_t stride = uint64_t(gridDim.x) * blockDim.x;
	uint32_t acc = 0;
	for (; i < units; i += stride) {
		const uint4* a = reinterpret_cast<const uint4*>(p + perm(i, units) * 64);
		const uint4 q0 = a[0], q1 = a[1], q2 = a[2], q3 = a[3];
		acc += q0.x ^ q1.y ^ q2.z ^ q3.w;
	}
	if (acc == 0x12345678u) *sink = acc;
}
// a wave reads 64 consecutive bytes of one row (one 64-byte half line per wave-load); rows are visited in scrambled order
__global__ void rows_b1(const uint8_t* __restrict__ p, uint64_t rows, uint32_t* sink) {
	const uint64_t wave = (uint64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
	const uint64_t waves = (uint64_t(gridDim.x) * blockDim.x) >> 6;
	const uint32_t lane = threadIdx.x & 63;
	uint32_t acc = 0;
	for (uint64_t r = wave; r < rows; r += waves) acc += p[perm(r, rows) * 64 + lane];
	if (acc == 0x12345678u) *sink = acc;
}

int main(int argc, char** argv) {
	const int reps = argc > 1 ? atoi(argv[1]) : 2;
	const uint64_t sizes[3] = {64ull << 20, 1ull << 30, 8ull << 30};
	uint8_t* buf = nullptr;
	uint32_t* sink = nullptr;
	CK(hipMalloc(&buf, sizes[2]));
	CK(hipMalloc(&sink, 4));
	CK(hipMemset(buf, 1, sizes[2]));
	CK(hipDeviceSynchronize());
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	int dispatch = 0;
	const int block = 256, grid = 256 * 8 * 4; // persistent-ish: 8192 workgroups, grid-stride loops
	for (int si = 0; si < 3; ++si) {
		const uint64_t B = sizes[si];
		for (int k = 0; k < 6; ++k) {
			for (int rep = 0; rep < reps; ++rep) {
				const char* name = "";
				uint64_t useful = 0, lines128 = 0, halves64 = 0;
				CK(hipEventRecord(e0));
				switch (k) {
				case 0: name = "stream16"; stream16<<<grid, block>>>(reinterpret_cast<const uint4*>(buf), B / 16, sink); useful = B; lines128 = B / 128; halves64 = B / 64; break;
				case 1: name = "gather_b1_l128"; gather<128, 1><<<grid, block>>>(buf, B / 128, sink); useful = B / 128; lines128 = B / 128; halves64 = B / 128; break;
				case 2: name = "gather_b1_l64"; gather<64, 1><<<grid, block>>>(buf, B / 64, sink); useful = B / 64; lines128 = B / 128; halves64 = B / 64; break;
				case 3: name = "gather_b4_l128"; gather<128, 4><<<grid, block>>>(buf, B / 128, sink); useful = B / 128 * 4; lines128 = B / 128; halves64 = B / 128; break;
				case 4: name = "gather_rec64"; gather_rec64<<<grid, block>>>(buf, B / 64, sink); useful = B; lines128 = B / 128; halves64 = B / 64; break;
				case 5: name = "rows_b1"; rows_b1<<<grid, block>>>(buf, B / 64, sink); useful = B; lines128 = B / 128; halves64 = B / 64; break;
				}
				CK(hipGetLastError());
				CK(hipEventRecord(e1));
				CK(hipEventSynchronize(e1));
				float ms = 0;
				CK(hipEventElapsedTime(&ms, e0, e1));
				printf("DISPATCH %d %s buffer_MiB %llu rep %d useful_bytes %llu lines128 %llu halves64 %llu ms %.4f\n", dispatch++, name,
					   (unsigned long long)(B >> 20), rep, (unsigned long long)useful, (unsigned long long)lines128, (unsigned long long)halves64, ms);
			}
		}
	}
	return 0;
}
