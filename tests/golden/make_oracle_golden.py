#!/usr/bin/env python3
"""Generates the oracle-produced regression fixtures (hit records + radiance) under tests/golden/.

These are outputs of oracle/oracle.c (the CPU restatement), not of the reference itself: they
freeze the oracle's behaviour at the point where it was pinned against the reference-derived
known answers (tests/test_oracle_pins.py), so that a later edit of oracle.c or of the HIP
kernel cannot drift silently.  Inputs are fully described by the arrays stored with them.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

oracle.build()
G = os.path.join(ROOT, "tests", "golden")


def camera(grid, h=0.8, v=-0.5):
    return (grid / 2, grid / 8, 0.8 * grid), oracle.camera_direction(h, v)


# BASELINE config 1: 256x256, 1 spp, primary-ray DDA into one 16^3 superchunk (128^3 voxels)
w = oracle.World(128, 128)
w.reset_device(True)
pos, d = camera(128)
acc, dbg, cnt, _ = w.render(oracle.make_camera(pos, d), oracle.make_frame(256, 256, spp=1, max_bounces=0, primary_only=1))
np.savez_compressed(os.path.join(G, "config1_primary.npz"), grid=128, width=256, height=256, cam_pos=np.float32(pos), cam_dir=d,
                    hits=dbg[..., :4].copy(), accum=acc, counters=np.array([cnt[k] for k in oracle.COUNTER_NAMES], np.uint64))

# 4-segment paths, 2 spp, ragged image (not a multiple of the 16x16 tile), 256^3 world
w = oracle.World(256, 256)
w.reset_device(True)
pos, d = camera(256)
acc, dbg, cnt, _ = w.render(oracle.make_camera(pos, d), oracle.make_frame(100, 70, spp=2, max_bounces=3))
np.savez_compressed(os.path.join(G, "path4_small.npz"), grid=256, width=100, height=70, spp=2, max_bounces=3, cam_pos=np.float32(pos),
                    cam_dir=d, dbg=dbg, accum=acc, counters=np.array([cnt[k] for k in oracle.COUNTER_NAMES], np.uint64),
                    world_hash=np.uint64(w.hash()))
print("wrote fixtures; world hash", hex(w.hash()))
