// valu_rates.hip -- issue cost of the VALU instructions the walk loops are made of, on gfx950.
// Each kernel runs ITER iterations of 16 independent instances of one instruction per wave; the grid fills every
// SIMD with W waves.  Reported: SIMD cycles per wave-instruction = W waves share one SIMD's issue.
//   build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rates.hip -o scratch/valu_rates   run: scratch/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(X) X X X X X X X X X X X X X X X X
constexpr int ITER = 4096;

#define KERNEL(name, ASM)                                                                                   \
	__global__ __launch_bounds__(256) void name(float* out, float a, float b) {                             \
		float v0 = a + threadIdx.x, v1 = b, v2 = a * 2, v3 = b * 3;                                        \
		unsigned u0 = threadIdx.x, u1 = blockIdx.x + 7;                                                    \
		unsigned long long w = threadIdx.x * 0x100000001ull;                                               \
		for (int i = 0; i < ITER; ++i) { REP16(asm volatile(ASM : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(u0), "+v"(u1), "+v"(w) : : "vcc", "scc", "s10", "s11");) } \
		out[blockIdx.x * 256 + threadIdx.x] = v0 + v1 + v2 + v3 + u0 + u1 + (float)w;                      \
	}

KERNEL(k_add_f32, "v_add_f32 %0, %1, %0\n")
KERNEL(k_fma_f32, "v_fma_f32 %0, %1, %2, %0\n")
KERNEL(k_pk_add_f32, "v_pk_add_f32 %6, %6, %6\n")
KERNEL(k_add_u32, "v_add_u32 %4, %5, %4\n")
KERNEL(k_add3_u32, "v_add3_u32 %4, %5, %4, %5\n")
KERNEL(k_and_or, "v_and_or_b32 %4, %5, %4, %5\n")
KERNEL(k_bfe, "v_bfe_u32 %4, %4, 2, 9\n")
KERNEL(k_mul24, "v_mul_u32_u24 %4, %5, %4\n")
KERNEL(k_mul_lo, "v_mul_lo_u32 %4, %5, %4\n")
KERNEL(k_lshr64, "v_lshrrev_b64 %6, %4, %6\n")
KERNEL(k_cmp_vcc, "v_cmp_lt_f32 vcc, %0, %1\n")
KERNEL(k_cmp_sgpr, "v_cmp_lt_f32 s[10:11], %0, %1\n")
KERNEL(k_cndmask_vcc, "v_cndmask_b32 %0, %1, %2, vcc\n")
KERNEL(k_cndmask_sgpr, "v_cndmask_b32 %0, %1, %0, s[10:11]\n")
KERNEL(k_cmp_cnd, "v_cmp_lt_f32 vcc, %0, %1\nv_cndmask_b32 %2, %3, %2, vcc\n")
KERNEL(k_bitop3, "v_bitop3_b32 %4, %4, %5, %4 bitop3:0x48\n")
KERNEL(k_mov, "v_mov_b32 %0, %1\n")
KERNEL(k_salu, "s_and_b64 s[10:11], s[10:11], vcc\n")
KERNEL(k_and, "v_and_b32 %4, %5, %4\n")
KERNEL(k_lshr32, "v_lshrrev_b32 %4, 3, %4\n")
KERNEL(k_lshl_or, "v_lshl_or_b32 %4, %5, 3, %4\n")
KERNEL(k_lshl_add, "v_lshl_add_u32 %4, %5, 3, %4\n")
KERNEL(k_mad24, "v_mad_u32_u24 %4, %5, %4, %5\n")
KERNEL(k_mul_f32, "v_mul_f32 %0, %1, %0\n")
KERNEL(k_min_f32, "v_min_f32 %0, %1, %0\n")
KERNEL(k_min3_f32, "v_min3_f32 %0, %1, %2, %0\n")
KERNEL(k_cmp_eq_u32, "v_cmp_eq_u32 vcc, %4, %5\n")
KERNEL(k_cvt, "v_cvt_f32_i32 %0, %4\n")
KERNEL(k_rcp, "v_rcp_f32 %0, %0\n")
KERNEL(k_bcnt, "v_bcnt_u32_b32 %4, %5, %4\n")
KERNEL(k_fma_f64, "v_fma_f64 %6, %6, %6, %6\n")
KERNEL(k_swap, "v_swap_b32 %0, %1\n")
KERNEL(k_swap2, "v_swap_b32 %0, %1\nv_swap_b32 %2, %3\n")
KERNEL(k_mov3, "v_mov_b32 %2, %0\nv_mov_b32 %0, %1\nv_mov_b32 %1, %2\n")
KERNEL(k_cnd2, "v_cndmask_b32 %2, %0, %1, vcc\nv_cndmask_b32 %1, %1, %0, vcc\nv_mov_b32 %0, %2\n")
KERNEL(k_readlane, "v_readlane_b32 s10, %0, 3\n")
KERNEL(k_writelane, "v_writelane_b32 %0, s10, 3\n")

KERNEL(k_sub_f32, "v_sub_f32 %0, %1, %0\n")
KERNEL(k_max_f32, "v_max_f32 %0, %1, %0\n")
KERNEL(k_fmac_f32, "v_fmac_f32 %0, %1, %2\n")
KERNEL(k_sub_u32, "v_sub_u32 %4, %5, %4\n")
KERNEL(k_or, "v_or_b32 %4, %5, %4\n")
KERNEL(k_xor, "v_xor_b32 %4, %5, %4\n")
KERNEL(k_lshl32, "v_lshlrev_b32 %4, 3, %4\n")
KERNEL(k_ashr32, "v_ashrrev_i32 %4, 3, %4\n")
KERNEL(k_min_u32, "v_min_u32 %4, %5, %4\n")
KERNEL(k_max_i32, "v_max_i32 %4, %5, %4\n")
KERNEL(k_mul_i24, "v_mul_i32_i24 %4, %5, %4\n")
KERNEL(k_bfi, "v_bfi_b32 %4, %5, %4, %5\n")
KERNEL(k_perm, "v_perm_b32 %4, %5, %4, %5\n")
KERNEL(k_alignbit, "v_alignbit_b32 %4, %5, %4, 7\n")
KERNEL(k_add_co, "v_add_co_u32 %4, vcc, %5, %4\n")
KERNEL(k_addc, "v_addc_co_u32 %4, vcc, %5, %4, vcc\n")
KERNEL(k_cvt_u32_f32, "v_cvt_u32_f32 %4, %0\n")
KERNEL(k_cvt_f32_u32, "v_cvt_f32_u32 %0, %4\n")
KERNEL(k_trunc, "v_trunc_f32 %0, %0\n")
KERNEL(k_floor, "v_floor_f32 %0, %0\n")
KERNEL(k_sqrt, "v_sqrt_f32 %0, %0\n")
KERNEL(k_rsq, "v_rsq_f32 %0, %0\n")
KERNEL(k_exp, "v_exp_f32 %0, %0\n")
KERNEL(k_log, "v_log_f32 %0, %0\n")
KERNEL(k_sin, "v_sin_f32 %0, %0\n")
KERNEL(k_mul_hi, "v_mul_hi_u32 %4, %5, %4\n")
KERNEL(k_mad_u64, "v_mad_u64_u32 %6, vcc, %4, %5, %6\n")
KERNEL(k_ldexp, "v_ldexp_f32 %0, %0, %4\n")
KERNEL(k_frexp_m, "v_frexp_mant_f32 %0, %0\n")
KERNEL(k_pk_mul_f32, "v_pk_mul_f32 %6, %6, %6\n")
KERNEL(k_pk_fma_f32, "v_pk_fma_f32 %6, %6, %6, %6\n")
KERNEL(k_mbcnt, "v_mbcnt_lo_u32_b32 %4, %5, %4\n")
KERNEL(k_dpp_mov, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
KERNEL(k_dpp_add, "v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n")
KERNEL(k_sdwa, "v_add_u32_sdwa %4, %5, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n")
KERNEL(k_cmp_lt_i32, "v_cmp_lt_i32 vcc, %4, %5\n")
KERNEL(k_cmp_class, "v_cmp_class_f32 vcc, %0, %4\n")
KERNEL(k_cmpx, "v_cmpx_le_u32 exec, %4, %4\n")
KERNEL(k_cnd_vcc_dep, "v_cmp_lt_f32 vcc, %0, %1\nv_cndmask_b32 %2, %3, %2, vcc\nv_cndmask_b32 %3, %2, %3, vcc\nv_cndmask_b32 %0, %0, %1, vcc\n")
KERNEL(k_add_mul_mix, "v_add_f32 %0, %1, %0\nv_bfe_u32 %4, %4, 2, 9\n")
KERNEL(k_add_and_mix, "v_add_f32 %0, %1, %0\nv_and_b32 %4, %5, %4\n")
KERNEL(k_salu_add, "s_add_u32 s10, s10, s11\n")
KERNEL(k_salu_valu_mix, "s_add_u32 s10, s10, s11\nv_add_f32 %0, %1, %0\n")
KERNEL(k_salu_bfe_mix, "s_add_u32 s10, s10, s11\nv_bfe_u32 %4, %4, 2, 9\n")
KERNEL(k_sub_lit, "v_subrev_u32 %4, 0x12345, %4\n")
KERNEL(k_add_lit_f32, "v_add_f32 %0, 0x3f8ccccd, %0\n")
KERNEL(k_mul_legacy, "v_mul_legacy_f32 %0, %1, %0\n")
KERNEL(k_med3, "v_med3_f32 %0, %1, %2, %0\n")
KERNEL(k_fma_lit, "v_fmaak_f32 %0, %1, %0, 0x3f8ccccd\n")
KERNEL(k_cvt_i32_f32, "v_cvt_i32_f32 %4, %0\n")
KERNEL(k_ffbh, "v_ffbh_u32 %4, %4\n")
KERNEL(k_ffbl, "v_ffbl_b32 %4, %4\n")
KERNEL(k_not, "v_not_b32 %4, %4\n")
KERNEL(k_bfrev, "v_bfrev_b32 %4, %4\n")
KERNEL(k_readfirst, "v_readfirstlane_b32 s10, %0\n")
KERNEL(k_lshl64, "v_lshlrev_b64 %6, %4, %6\n")
KERNEL(k_add_u32_e64, "v_add_u32_e64 %4, %5, %4\n")
KERNEL(k_add_f32_e64, "v_add_f32_e64 %0, %1, -%0\n")
KERNEL(k_add_f32_abs, "v_add_f32_e64 %0, |%1|, %0\n")
KERNEL(k_mul_f32_sgpr, "v_mul_f32 %0, s10, %0\n")
KERNEL(k_add_u32_sgpr, "v_add_u32 %4, s10, %4\n")
KERNEL(k_add_u32_inl, "v_add_u32 %4, 5, %4\n")


// the same v_add_f32 / v_bfe_u32 loops with part of the wave masked off (an ordinary divergent `if`, so that the compiler
// keeps EXEC consistent): does an instruction cost less when half of its 64 lanes are inactive?
#define KERNEL_EXEC(name, ASM, COND)                                                                        \
	__global__ __launch_bounds__(256) void name(float* out, float a, float b) {                             \
		float v0 = a + threadIdx.x, v1 = b, v2 = a * 2, v3 = b * 3;                                        \
		unsigned u0 = threadIdx.x, u1 = blockIdx.x + 7;                                                    \
		unsigned long long w = threadIdx.x * 0x100000001ull;                                               \
		const unsigned lane = threadIdx.x & 63u;                                                           \
		if (COND) {                                                                                        \
			for (int i = 0; i < ITER; ++i) { REP16(asm volatile(ASM : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(u0), "+v"(u1), "+v"(w) : : "vcc", "scc", "s10", "s11");) } \
		}                                                                                                  \
		out[blockIdx.x * 256 + threadIdx.x] = v0 + v1 + v2 + v3 + u0 + u1 + (float)w;                      \
	}
KERNEL_EXEC(k_add_f32_lo32, "v_add_f32 %0, %1, %0\n", lane < 32u)
KERNEL_EXEC(k_add_f32_lo16, "v_add_f32 %0, %1, %0\n", lane < 16u)
KERNEL_EXEC(k_add_f32_even, "v_add_f32 %0, %1, %0\n", (lane & 1u) == 0u)
KERNEL_EXEC(k_bfe_lo32, "v_bfe_u32 %4, %4, 2, 9\n", lane < 32u)
KERNEL_EXEC(k_bfe_lo16, "v_bfe_u32 %4, %4, 2, 9\n", lane < 16u)
KERNEL_EXEC(k_bfe_one, "v_bfe_u32 %4, %4, 2, 9\n", lane == 0u)
KERNEL_EXEC(k_cmp_lo32, "v_cmp_lt_f32 vcc, %0, %1\n", lane < 32u)
KERNEL_EXEC(k_rcp_lo32, "v_rcp_f32 %0, %0\n", lane < 32u)
KERNEL_EXEC(k_rcp_lo16, "v_rcp_f32 %0, %0\n", lane < 16u)

template <typename K>
static void run(const char* name, K kernel, int per_iter, float* out, int cus) {
	for (int waves : {4, 8}) {
		const int blocks = cus * waves; // 256 threads = 4 waves = one per SIMD
		hipEvent_t e0, e1;
		hipEventCreate(&e0); hipEventCreate(&e1);
		hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, 1.0f, 2.0f);
		hipEventRecord(e0);
		hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, 1.0f, 2.0f);
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		float ms = 0;
		hipEventElapsedTime(&ms, e0, e1);
		const double insts_per_simd = static_cast<double>(ITER) * 16 * per_iter * waves;
		const double cycles = ms * 1e-3 * 2.4e9; // nominal 2.4 GHz
		std::printf("%-14s %d waves/SIMD: %.2f SIMD-cycles per wave-instruction\n", name, waves, cycles / insts_per_simd);
		hipEventDestroy(e0); hipEventDestroy(e1);
	}
}

int main() {
	hipDeviceProp_t prop;
	hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount;
	float* out;
	hipMalloc(&out, sizeof(float) * 256 * cus * 8);
#define RUN(k, n) run(#k, k, n, out, cus)
	RUN(k_add_f32, 1); RUN(k_fma_f32, 1); RUN(k_pk_add_f32, 1); RUN(k_add_u32, 1); RUN(k_add3_u32, 1); RUN(k_and_or, 1); RUN(k_bfe, 1);
	RUN(k_mul24, 1); RUN(k_mul_lo, 1); RUN(k_lshr64, 1); RUN(k_cmp_vcc, 1); RUN(k_cmp_sgpr, 1); RUN(k_cndmask_vcc, 1); RUN(k_cndmask_sgpr, 1);
	RUN(k_cmp_cnd, 2); RUN(k_bitop3, 1); RUN(k_mov, 1); RUN(k_salu, 1);
	RUN(k_and, 1); RUN(k_lshr32, 1); RUN(k_lshl_or, 1); RUN(k_lshl_add, 1); RUN(k_mad24, 1); RUN(k_mul_f32, 1); RUN(k_min_f32, 1); RUN(k_min3_f32, 1);
	RUN(k_cmp_eq_u32, 1); RUN(k_cvt, 1); RUN(k_rcp, 1); RUN(k_bcnt, 1); RUN(k_fma_f64, 1);
	RUN(k_swap, 1); RUN(k_swap2, 2); RUN(k_mov3, 3); RUN(k_cnd2, 3); RUN(k_readlane, 1); RUN(k_writelane, 1);
	RUN(k_sub_f32,1); RUN(k_max_f32,1); RUN(k_fmac_f32,1); RUN(k_sub_u32,1); RUN(k_or,1); RUN(k_xor,1); RUN(k_lshl32,1); RUN(k_ashr32,1); RUN(k_min_u32,1); RUN(k_max_i32,1);
	RUN(k_mul_i24,1); RUN(k_bfi,1); RUN(k_perm,1); RUN(k_alignbit,1); RUN(k_add_co,1); RUN(k_addc,1); RUN(k_cvt_u32_f32,1); RUN(k_cvt_f32_u32,1); RUN(k_trunc,1); RUN(k_floor,1);
	RUN(k_sqrt,1); RUN(k_rsq,1); RUN(k_exp,1); RUN(k_log,1); RUN(k_sin,1); RUN(k_mul_hi,1); RUN(k_mad_u64,1); RUN(k_ldexp,1); RUN(k_frexp_m,1); RUN(k_pk_mul_f32,1); RUN(k_pk_fma_f32,1);
	RUN(k_mbcnt,1); RUN(k_dpp_mov,1); RUN(k_dpp_add,1); RUN(k_sdwa,1); RUN(k_cmp_lt_i32,1); RUN(k_cmp_class,1); RUN(k_cmpx,1); RUN(k_cnd_vcc_dep,4); RUN(k_add_mul_mix,2); RUN(k_add_and_mix,2);
	RUN(k_salu_add,1); RUN(k_salu_valu_mix,2); RUN(k_salu_bfe_mix,2); RUN(k_sub_lit,1); RUN(k_add_lit_f32,1); RUN(k_mul_legacy,1); RUN(k_med3,1); RUN(k_fma_lit,1); RUN(k_cvt_i32_f32,1);
	RUN(k_ffbh,1); RUN(k_ffbl,1); RUN(k_not,1); RUN(k_bfrev,1); RUN(k_readfirst,1); RUN(k_lshl64,1); RUN(k_add_u32_e64,1); RUN(k_add_f32_e64,1); RUN(k_add_f32_abs,1);
	RUN(k_add_f32_lo32,1); RUN(k_add_f32_lo16,1); RUN(k_add_f32_even,1); RUN(k_bfe_lo32,1); RUN(k_bfe_lo16,1); RUN(k_bfe_one,1); RUN(k_cmp_lo32,1); RUN(k_rcp_lo32,1); RUN(k_rcp_lo16,1);
	RUN(k_mul_f32_sgpr,1); RUN(k_add_u32_sgpr,1); RUN(k_add_u32_inl,1);
	return 0;
}
