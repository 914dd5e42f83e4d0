// vmm_probe2.hip -- repeatability of grow-in-place mappings: many reserve / map / grow / free cycles, several styles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void touch(unsigned* p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = (unsigned)i; }
int main(int argc, char** argv) {
	const int style = argc > 1 ? atoi(argv[1]) : 0; // 0: per-chunk SetAccess, free VA; 1: whole-range SetAccess; 2: per-chunk, never free VA; 3: per-chunk, sync before SetAccess; 4: uniform 4 MiB chunks
	hipSetDevice(0);
	hipMemAllocationProp prop{};
	prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
	hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
	const size_t MB = 1 << 20;
	int fails = 0, fallback_ok = 0;
	for (int it = 0; it < 24; ++it) {
		void* va = nullptr; const size_t total = 1024 * MB;
		if (hipMemAddressReserve(&va, total, 0, nullptr, 0) != hipSuccess) { printf("reserve failed it %d\n", it); return 1; }
		char* base = (char*)va;
		std::vector<hipMemGenericAllocationHandle_t> hs; std::vector<size_t> offs, sizes;
		size_t mapped = 0;
		for (int g = 0; g < 6; ++g) {
			size_t sz = (style == 4) ? 4 * MB : (mapped == 0 ? 4 * MB : mapped);
			hipMemGenericAllocationHandle_t h;
			if (hipMemCreate(&h, sz, &prop, 0) != hipSuccess) { printf("create failed\n"); return 1; }
			if (hipMemMap(base + mapped, sz, 0, h, 0) != hipSuccess) { printf("map failed it %d g %d\n", it, g); return 1; }
			if (style == 3) hipDeviceSynchronize();
			hipError_t e = (style == 1) ? hipMemSetAccess(base, mapped + sz, &acc, 1) : hipMemSetAccess(base + mapped, sz, &acc, 1);
			if (e != hipSuccess) {
				(void)hipGetLastError();
				fails++;
				hipError_t e2 = hipMemSetAccess(base, mapped + sz, &acc, 1);
				hipError_t e3 = hipMemSetAccess(base + mapped, sz, &acc, 1);
				printf("it %d grow %d (offset %zu MiB size %zu MiB va %p): SetAccess %s; whole-range retry %s; same retry %s\n", it, g, mapped / MB, sz / MB, va, hipGetErrorString(e),
					   hipGetErrorString(e2), hipGetErrorString(e3));
				(void)hipGetLastError();
				if (e2 == hipSuccess || e3 == hipSuccess) fallback_ok++;
			}
			hs.push_back(h); offs.push_back(mapped); sizes.push_back(sz);
			mapped += sz;
			touch<<<(unsigned)(mapped / 4 / 256), 256>>>((unsigned*)base, mapped / 4); // kernel over everything mapped so far, in flight during the next growth
		}
		hipError_t es = hipDeviceSynchronize();
		if (es != hipSuccess) { printf("it %d: sync %s\n", it, hipGetErrorString(es)); return 1; }
		for (size_t k = 0; k < hs.size(); ++k) { hipMemUnmap(base + offs[k], sizes[k]); hipMemRelease(hs[k]); }
		if (style != 2) hipMemAddressFree(va, total);
	}
	printf("style %d: %d SetAccess failures in 24 x 6 growths, %d recovered by a retry\n", style, fails, fallback_ok);
	return 0;
}
