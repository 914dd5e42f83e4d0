// oracle/ref_simplex_shim.cpp -- TEST INFRASTRUCTURE ONLY.
// extern "C" doorway into the REAL reference class SimplexNoise (reference
// src/SimplexNoise.h:17-55), linked against the reference's own SimplexNoise.cpp by
// oracle/Makefile `make ref`.  Used to pin oracle.c's noise restatement bit-exactly and to
// generate tests/golden/noise_ref.npz (tests/golden/make_noise_golden.py).
#include <cstddef>
#include "SimplexNoise.h"

extern "C" {
// Scene.cpp:45 constructs SimplexNoise(1.f, 1.f, 2.f, 0.5f); Scene.cpp:53 calls fractal(8, x, y).
float ref_fractal2(int octaves, float x, float y) {
	SimplexNoise noise(1.f, 1.f, 2.f, 0.5f);
	return noise.fractal(static_cast<size_t>(octaves), x, y);
}
float ref_noise2(float x, float y) { return SimplexNoise::noise(x, y); }
void ref_fractal2_grid(int octaves, int n, const float* xs, const float* ys, float* out) {
	SimplexNoise noise(1.f, 1.f, 2.f, 0.5f);
	for (int i = 0; i < n; i++) out[i] = noise.fractal(static_cast<size_t>(octaves), xs[i], ys[i]);
}
}
