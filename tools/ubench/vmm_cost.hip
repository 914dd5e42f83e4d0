// vmm_cost.hip -- host time of growing a mapped arena the way scene.cpp does: hipMemCreate + hipMemMap of a chunk behind the
// mapped part + hipMemSetAccess over the whole mapped range, while a kernel is running on the already-mapped part.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void spin(unsigned* p, size_t n, int rounds) {
	size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
	unsigned v = 0;
	for (int r = 0; r < rounds; ++r) for (size_t k = i; k < n; k += (size_t)gridDim.x * blockDim.x) v += p[k];
	if (v == 0x12345u) p[0] = v;
}
int main() {
	hipSetDevice(0);
	hipMemAllocationProp prop{};
	prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
	hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
	const size_t MB = 1 << 20, total = 16384 * MB;
	void* va = nullptr;
	if (hipMemAddressReserve(&va, total, 0, nullptr, 0) != hipSuccess) return 1;
	char* base = (char*)va;
	size_t mapped = 0;
	for (size_t sz = 4 * MB; mapped + sz <= 8192 * MB; sz = mapped) {
		if (mapped) spin<<<1024, 256>>>((unsigned*)base, mapped / 4 > (64 * MB) ? 64 * MB : mapped / 4, 20); // frames in flight
		auto t0 = std::chrono::steady_clock::now();
		hipMemGenericAllocationHandle_t h;
		hipError_t e1 = hipMemCreate(&h, sz, &prop, 0);
		auto t1 = std::chrono::steady_clock::now();
		hipError_t e2 = hipMemMap(base + mapped, sz, 0, h, 0);
		auto t2 = std::chrono::steady_clock::now();
		hipError_t e3 = hipMemSetAccess(base, mapped + sz, &acc, 1);
		auto t3 = std::chrono::steady_clock::now();
		auto us = [](auto a, auto b) { return std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
		printf("grow %6zu -> %6zu MiB: create %6ld us, map %6ld us, set access (whole range) %6ld us  [%s %s %s]\n", mapped / MB, (mapped + sz) / MB, (long)us(t0, t1), (long)us(t1, t2), (long)us(t2, t3),
			   hipGetErrorString(e1), hipGetErrorString(e2), hipGetErrorString(e3));
		mapped += sz;
		hipDeviceSynchronize();
	}
	return 0;
}
