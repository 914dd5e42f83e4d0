"""CPU tests of the exact multi-cell walk (brickmap_amd/csrc/jump.h) and of the octant cube field it relies on.

jump.h is plain C++ shared by the device kernels and tests/jump_check.cpp, which replays millions of jumps against
the reference's one-cell-at-a-time stepping (src/voxel.cuh:249-258): identical tmax bit patterns, step counts and
final axis.  The cube field (Scene / bm_host_cube_field) is checked against a brute-force definition on the oracle's
world: a jump may only skip cells that the field promises to be empty.
"""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_jump_replays_reference_stepping(tmp_path):
    exe = tmp_path / "jump_check"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", str(exe), os.path.join(ROOT, "tests", "jump_check.cpp")])
    r = subprocess.run([str(exe), "60000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "failures 0" in r.stdout
    jumps = int(r.stdout.split()[1])
    assert jumps > 1_000_000  # the run did exercise the code


def test_cube_field_matches_bruteforce(bm, orc):
    G = 128  # one supercell: 16^3 brick cells
    field = bm.host_cube_field(G, G)
    n = G // 8
    assert field.shape == (8, n + 2, n + 2, n + 2)
    world = orc.World(G, G)
    occ = np.zeros((n, n, n), bool)  # [z, y, x]
    occ[:] = (world.sc_indices(0) != 0).reshape(16, 16, 16)
    assert 0 < occ.sum() < occ.size
    for o in range(8):
        plane = field[o]
        border = np.ones_like(plane, bool)
        border[1:-1, 1:-1, 1:-1] = False
        assert (plane[border] == 255).all()
        inner = plane[1:-1, 1:-1, 1:-1]
        assert ((inner == 0) == occ).all()
        dx, dy, dz = (-1 if o & 1 else 1), (-1 if o & 2 else 1), (-1 if o & 4 else 1)
        rng = np.random.default_rng(o)
        for _ in range(400):
            x, y, z = (int(v) for v in rng.integers(0, n, 3))
            if occ[z, y, x]:
                continue
            e = 0  # brute force: the largest cube of empty in-grid cells anchored at (x, y, z) along (dx, dy, dz)
            while True:
                m = e + 1
                xs = sorted((x, x + dx * (m - 1)))
                ys = sorted((y, y + dy * (m - 1)))
                zs = sorted((z, z + dz * (m - 1)))
                if xs[0] < 0 or ys[0] < 0 or zs[0] < 0 or xs[1] >= n or ys[1] >= n or zs[1] >= n:
                    break
                if occ[zs[0]:zs[1] + 1, ys[0]:ys[1] + 1, xs[0]:xs[1] + 1].any():
                    break
                e = m
            assert inner[z, y, x] == min(e, 254), (o, x, y, z)
