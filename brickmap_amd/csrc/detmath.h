// detmath.h -- deterministic sin/cos for ray-direction sampling (product side).
//
// Numeric contract of this project (DESIGN.md "Numeric contract"): ray geometry uses only
// IEEE + - * / sqrt and this sincos, compiled with -ffp-contract=off, so that the HIP kernel
// and the CPU oracle agree bit-for-bit on every hit.  Spec:
//   xd = (double)x;  k = (int)(xd*(2/pi) + (xd >= 0 ? 0.5 : -0.5));
//   r  = (xd - k*PIO2_HI) - k*PIO2_LO;
//   sin(r), cos(r) by the fdlibm kernel polynomials in double; quadrant fix-up by k & 3;
//   results rounded to float.
// The reference calls cos()/sin() of CUDA's libdevice here (kernel.cu:102,296; sunsky.cu:183),
// which are themselves only specified to a couple of ulp.
#pragma once

#if defined(__HIPCC__)
#define BM_HD __host__ __device__ inline
#else
#define BM_HD inline
#endif

namespace bm {

BM_HD void det_sincos(float x, float& s_out, float& c_out) {
	const double kTwoOverPi = 6.36619772367581382433e-01;
	const double kPio2Hi = 1.57079632679489655800e+00;
	const double kPio2Lo = 6.12323399573676603587e-17;
	const double xd = static_cast<double>(x);
	const int k = static_cast<int>(xd * kTwoOverPi + (xd >= 0.0 ? 0.5 : -0.5));
	const double kd = static_cast<double>(k);
	const double r = (xd - kd * kPio2Hi) - kd * kPio2Lo;
	const double z = r * r;
	const double ps = -1.66666666666666324348e-01 +
					  z * (8.33333333332248946124e-03 +
						   z * (-1.98412698298579493134e-04 +
								z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10))));
	const double sr = r + (r * z) * ps;
	const double pc = 4.16666666666666019037e-02 +
					  z * (-1.38888888888741095749e-03 +
						   z * (2.48015872894767294178e-05 +
								z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
	const double cr = (1.0 - 0.5 * z) + (z * z) * pc;
	double s, c;
	switch (k & 3) {
	case 0: s = sr; c = cr; break;
	case 1: s = cr; c = -sr; break;
	case 2: s = -sr; c = -cr; break;
	default: s = -cr; c = sr; break;
	}
	s_out = static_cast<float>(s);
	c_out = static_cast<float>(c);
}

} // namespace bm
