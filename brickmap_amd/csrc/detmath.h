// detmath.h -- deterministic sin/cos for ray-direction sampling (product side).
//
// Numeric contract of this project (DESIGN.md "Numeric contract"): ray geometry uses only
// IEEE + - * / sqrt and this sincos, compiled with -ffp-contract=off, so that the HIP kernel
// and the CPU oracle agree bit-for-bit on every hit.  Spec (all operations in fp32, in exactly this order):
//   k = (int)(x * (2/pi) + (x >= 0 ? 0.5 : -0.5));                       (truncation)
//   r = ((x - k*C1) - k*C2) - k*C3;        C1 + C2 + C3 = pi/2, C1 and C2 short enough that k*C is exact
//   z = r*r;
//   sin r = ((((S3*z + S2)*z + S1)*z)*r) + r;      cos r = ((((K3*z + K2)*z + K1)*z)*z - 0.5*z) + 1;
//   quadrant fix-up by k & 3.
// Error <= 1.5 ulp on the sampling domain |x| <= 2 pi (tests/test_oracle_units.py checks |x| <= 7).
// The reference calls cos()/sin() of CUDA's libdevice here (kernel.cu:102,296; sunsky.cu:183), which are
// themselves only specified to a couple of ulp.  (An earlier version evaluated fdlibm's double-precision kernels and
// rounded: 124 fp64 instructions per shade, 3.5 % of the frame, for accuracy the reference does not have.)
#pragma once

#if defined(__HIPCC__)
#define BM_HD __host__ __device__ inline
#else
#define BM_HD inline
#endif

namespace bm {

BM_HD void det_sincos(float x, float& s_out, float& c_out) {
	const float kTwoOverPi = 0.6366197723675814f;
	const float kC1 = 1.5703125f, kC2 = 4.837512969970703125e-4f, kC3 = 7.54978995489188e-8f;
	const float kS1 = -1.6666654611e-1f, kS2 = 8.3321608736e-3f, kS3 = -1.9515295891e-4f;
	const float kK1 = 4.166664568298827e-2f, kK2 = -1.388731625493765e-3f, kK3 = 2.443315711809948e-5f;
	const int k = static_cast<int>(x * kTwoOverPi + (x >= 0.0f ? 0.5f : -0.5f));
	const float kf = static_cast<float>(k);
	const float r = ((x - kf * kC1) - kf * kC2) - kf * kC3;
	const float z = r * r;
	const float sr = ((((kS3 * z + kS2) * z + kS1) * z) * r) + r;
	const float cr = ((((kK3 * z + kK2) * z + kK1) * z) * z - 0.5f * z) + 1.0f;
	float s, c;
	switch (k & 3) {
	case 0: s = sr; c = cr; break;
	case 1: s = cr; c = -sr; break;
	case 2: s = -sr; c = -cr; break;
	default: s = -cr; c = sr; break;
	}
	s_out = s;
	c_out = c;
}

} // namespace bm
