// vmm_probe.hip -- which hipMemMap / hipMemSetAccess patterns this runtime accepts (grow-in-place arena, scene.cpp).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define TRY(x) do { hipError_t e_ = (x); printf("  %-70s -> %s\n", #x, e_ == hipSuccess ? "ok" : hipGetErrorString(e_)); if (e_ != hipSuccess) (void)hipGetLastError(); } while (0)
__global__ void touch(unsigned* p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = (unsigned)i; }
int main() {
	int dev = 0, vmm = 0;
	hipSetDevice(dev);
	hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, dev);
	hipMemAllocationProp prop{};
	prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
	size_t gmin = 0, grec = 0;
	hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum);
	hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended);
	printf("vmm %d granularity min %zu recommended %zu\n", vmm, gmin, grec);
	hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
	const size_t MB = 1 << 20;
	for (int variant = 0; variant < 4; ++variant) {
		printf("variant %d\n", variant);
		void* va = nullptr; const size_t total = 256 * MB;
		TRY(hipMemAddressReserve(&va, total, 0, nullptr, 0));
		char* base = (char*)va;
		hipMemGenericAllocationHandle_t h0, h1, h2;
		TRY(hipMemCreate(&h0, 4 * MB, &prop, 0));
		TRY(hipMemMap(base, 4 * MB, 0, h0, 0));
		TRY(hipMemSetAccess(base, 4 * MB, &acc, 1));
		touch<<<1024, 256>>>((unsigned*)base, MB); // a kernel in flight while the range grows
		TRY(hipMemCreate(&h1, 4 * MB, &prop, 0));
		TRY(hipMemMap(base + 4 * MB, 4 * MB, 0, h1, 0));
		if (variant == 0) TRY(hipMemSetAccess(base + 4 * MB, 4 * MB, &acc, 1));       // the new chunk only
		if (variant == 1) TRY(hipMemSetAccess(base, 8 * MB, &acc, 1));                 // the whole mapped range
		if (variant == 2) { TRY(hipDeviceSynchronize()); TRY(hipMemSetAccess(base + 4 * MB, 4 * MB, &acc, 1)); }
		if (variant == 3) { TRY(hipMemSetAccess(base + 4 * MB, 4 * MB, &acc, 1)); }
		TRY(hipMemCreate(&h2, 8 * MB, &prop, 0));
		TRY(hipMemMap(base + 8 * MB, 8 * MB, 0, h2, 0));
		if (variant == 1) TRY(hipMemSetAccess(base, 16 * MB, &acc, 1)); else TRY(hipMemSetAccess(base + 8 * MB, 8 * MB, &acc, 1));
		touch<<<(unsigned)(4 * MB / 256), 256>>>((unsigned*)base, 4 * MB);
		TRY(hipDeviceSynchronize());
		unsigned v = 0;
		TRY(hipMemcpy(&v, base + 12 * MB, 4, hipMemcpyDeviceToHost));
		printf("  value at 12 MiB: %u (want %u)\n", v, (unsigned)(3 * MB));
		TRY(hipMemUnmap(base, 4 * MB)); TRY(hipMemUnmap(base + 4 * MB, 4 * MB)); TRY(hipMemUnmap(base + 8 * MB, 8 * MB));
		TRY(hipMemRelease(h0)); TRY(hipMemRelease(h1)); TRY(hipMemRelease(h2));
		TRY(hipMemAddressFree(va, total));
	}
	return 0;
}
