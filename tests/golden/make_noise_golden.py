#!/usr/bin/env python3
"""Generates tests/golden/noise_ref.npz from the REAL reference SimplexNoise.cpp.

Runs only where /root/reference exists (the build container): oracle/Makefile `make ref` compiles
the reference's own src/SimplexNoise.cpp into oracle/_ref/libref_simplex.so, and this script
records its outputs -- inputs and expected outputs only, no reference source -- so that the
oracle's restatement stays pinned on machines that have neither the reference nor that library.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

oracle.build()
R = oracle.ref_lib()
assert R is not None, "oracle/_ref/libref_simplex.so missing: run `make -C oracle ref` where /root/reference exists"
rng = np.random.default_rng(20250614)
# (a) the exact coordinates Scene.cpp:53 feeds the generator: (world voxel)/2048 on a strided lattice of a 4096^2 world
gx, gy = np.meshgrid(np.arange(0, 4096, 67, dtype=np.float32), np.arange(0, 4096, 61, dtype=np.float32))
xs = np.concatenate([(gx.ravel() / np.float32(2048.0)), (rng.random(2048) * 8 - 4).astype(np.float32)]).astype(np.float32)
ys = np.concatenate([(gy.ravel() / np.float32(2048.0)), (rng.random(2048) * 8 - 4).astype(np.float32)]).astype(np.float32)
fr = np.zeros_like(xs)
R.ref_fractal2_grid(8, xs.size, xs.ctypes.data, ys.ctypes.data, fr.ctypes.data)
n2 = np.array([R.ref_noise2(float(x), float(y)) for x, y in zip(xs, ys)], dtype=np.float32)
out = os.path.join(ROOT, "tests", "golden", "noise_ref.npz")
np.savez_compressed(out, xs=xs, ys=ys, fractal8=fr, noise2=n2)
print("wrote", out, xs.size, "points")
