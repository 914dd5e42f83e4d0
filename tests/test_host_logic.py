"""Host-side logic of the product (no GPU): the C-ABI loads and exports every declared symbol,
row sharding, the CPU world builder against the oracle, and the layering rules."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "brickmap.h")).read()
    return sorted(set(re.findall(r"BM_API\s+[\w\s\*]+?\b(bm_\w+)\s*\(", text)))


def test_cabi_exports_every_declared_symbol(bm):
    import ctypes as C
    from brickmap_amd import _lib
    names = declared_symbols()
    assert len(names) >= 30
    L = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), f"{n} is declared in include/brickmap.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "python bindings and header disagree"


def test_error_reporting_without_gpu(bm):
    from brickmap_amd import _lib
    L = _lib.load()
    assert L.bm_scene_get_info(None, None) == 10001  # BM_EINVAL, message set, no abort
    assert b"null scene" in L.bm_last_error_string()
    import ctypes as C
    out = np.zeros((128, 128), np.float32)
    assert L.bm_host_column_heights(100, 128, 0, 0, out.ctypes.data) == 10001  # not a multiple of 128


def test_comm_argument_errors_without_gpu(bm, tmp_path):
    """The exchange's entry points refuse bad arguments before they touch RCCL or a device, and a missing RCCL library is an
    error code with a message, not a crash (checked in a child process: the binding is made once per process)."""
    import ctypes as C
    from brickmap_amd import _lib
    L = _lib.load()
    h = C.c_void_p()
    idbuf = (C.c_ubyte * 128)()
    assert L.bm_comm_create(0, 2, 2, idbuf, C.byref(h)) == 10001 and b"bad argument" in L.bm_last_error_string()   # rank outside the world
    assert L.bm_comm_create(0, 0, 0, idbuf, C.byref(h)) == 10001
    assert L.bm_comm_create(0, 0, 1, None, C.byref(h)) == 10001
    assert L.bm_comm_unique_id(None) == 10001
    assert L.bm_gather_frame(None, None, None, 16, 16, 16, 0, None) == 10001 and b"null communicator" in L.bm_last_error_string()
    assert L.bm_reduce_frame(None, None, None, 0, 0, None) == 10001
    assert L.bm_comm_barrier(None, None) == 10001 and L.bm_comm_selftest(None, None) == 10001
    assert L.bm_comm_info(None, None, None) == 10001
    L.bm_comm_destroy(None)  # a no-op
    code = ("import ctypes as C, sys; sys.path.insert(0, %r); from brickmap_amd import _lib; L = _lib.load(); b = (C.c_ubyte * 128)(); "
            "r = L.bm_comm_unique_id(b); print(r, L.bm_last_error_string().decode())" % ROOT)
    env = dict(os.environ, BM_RCCL_LIBRARY=str(tmp_path / "no_such_rccl.so"), LD_LIBRARY_PATH="")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    # either the named library is missing and a system RCCL was found instead (an id is made: 0), or nothing binds: BM_ESTATE with a message
    assert out.stdout.startswith("0 ") or ("10002" in out.stdout and "not found" in out.stdout), out.stdout


def test_local_rows_matches_shard_rows(bm):
    for H in (1, 15, 16, 17, 70, 1080, 2160):
        for band in (1, 7, 16, 32):
            for world in (1, 2, 3, 8):
                owned = [bm.dist.shard_rows(H, band, r, world) for r in range(world)]
                assert sorted(np.concatenate(owned).tolist()) == list(range(H))
                for r in range(world):
                    got = bm.local_rows(bm.FrameParams(64, H, band_rows=band, shard_rank=r, shard_count=world))
                    assert got == len(owned[r])


def test_camera_update_matches_oracle(bm, orc):
    for h, v in [(0.8, -0.5), (0.0, 0.0), (-3.1, 1.2), (10.0, -1.5)]:
        cam = bm.Camera(horizontal_angle=h, vertical_angle=v).update()
        assert np.array_equal(np.float32(cam.direction), orc.camera_direction(h, v))
    c = bm.Camera()
    assert c.position == (512.0, 512.0, 300.0) and c.up == (0.0, 0.0, 1.0) and c.focalDistance == 1.0 and c.lensRadius == 0.0


def test_world_builder_matches_oracle(bm, orc):
    """brickmap_amd/csrc/world.cpp (product) against oracle.c (checker): bit-identical heights,
    index words and bricks, on a cubic and on a non-cubic world."""
    for (gs, gh) in [(256, 256), (384, 128)]:
        w = orc.World(gs, gh)
        sxy, sz_n = gs // 128, gh // 128
        for sz in range(sz_n):
            for sy in range(sxy):
                for sx in range(sxy):
                    sc = sx + sy * sxy + sz * sxy * sxy
                    idx, bricks = bm.host_generate_supercell(gs, gh, sx, sy, sz)
                    assert np.array_equal(idx, w.sc_indices(sc))
                    assert np.array_equal(bricks, w.sc_bricks(sc))
        h = bm.host_column_heights(gs, gh, sxy - 1, 0)
        assert np.array_equal(h.view(np.uint32), w.column_heights(sxy - 1, 0).view(np.uint32))


def test_product_never_touches_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use oracle/."""
    pkg = os.path.join(ROOT, "brickmap_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip", ".hpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in text and "from oracle" not in text and "liboracle" not in text, f
                assert "oracle/" not in text.replace("oracle/oracle.c orc_counters", ""), f
    inc = open(os.path.join(ROOT, "include", "brickmap.h")).read()
    assert "liboracle" not in inc


def test_no_reference_reads_at_runtime():
    """/root/reference does not exist on the GPU box: product, bench and smoke never open it."""
    for rel in ["bench.py", "__graft_entry__.py", "brickmap_amd/_lib.py", "brickmap_amd/host.py", "brickmap_amd/dist.py"]:
        assert "/root/reference" not in open(os.path.join(ROOT, rel)).read(), rel


def test_cpp_mirror_example_builds():
    """include/brickmap.hpp + examples/headless_main.cpp (the reference-shaped C++ driver) compile and link
    against the in-tree library with plain g++ (no device needed to build)."""
    import subprocess
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "examples")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert os.path.exists(os.path.join(ROOT, "examples", "headless_main"))


def test_cpp_mirror_header_compiles_standalone(tmp_path):
    """include/brickmap.hpp + the headless example are plain C++17 against the C-ABI header: no HIP, no torch."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    src = os.path.join(ROOT, "examples", "headless_main.cpp")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", src])
    # the C-ABI header itself is C: it must also compile as C11
    probe = tmp_path / "probe.c"
    probe.write_text('#include "brickmap.h"\nint main(void) { bm_frame_params p; (void)p; return sizeof(bm_camera) ? 0 : 1; }\n')
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(probe)])


def test_bench_helpers_run_without_a_gpu():
    """bench.py's host-side helpers: workload table, CPU discovery, and the committed PMC summaries roofline.traffic cites."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b.workload("config2") == (1920, 1080, 1, 3, 8, False)  # BASELINE.json configs[1]
    assert b.workload("config3")[5] and not b.workload("config5")[5]
    model, physical, logical = b.host_cpu()
    assert 1 <= physical <= logical and 1 <= b.cpu_quota() <= logical and isinstance(model, str)
    for wl in ("config2", "config3", "config5"):
        t = b.pmc_traffic(wl)
        assert t["traffic_per_step"] and t["traffic_per_step"] > 0 and f"_pmc_summary_{wl}.json" in t["traffic_source"]
        lim = b.limiter_of(wl, 7)
        assert lim["limiter"] and "7 waves per SIMD" in lim["limiter"] and f"_pmc_summary_{wl}.json" in lim["limiter_source"]
    assert b.limiter_of("config1") == {"limiter": None, "limiter_source": None}  # no committed PMC summary: no claim
    assert b.pmc_traffic("no-such-workload") == {"traffic_per_step": None, "valu_lanes": None, "valu_insts": None, "traffic_source": None}
    f = b.counter_figures("config2", 1.0e-3, 20)
    assert 0 < f["frac_by_counters"] < 1 and 10 < f["valu_lanes"] < 64 and f["valu_insts"] > 1e8 and f["traffic"] == 20 * f["traffic_per_step"]
    assert set(f["traffic_age"]) == {"summary", "collected_at_commit", "head", "commits_behind_head"} and f["traffic_age"]["summary"] in f["traffic_source"]
    # steps are cut into launches as evenly as possible (bench.py issue / the frame ring)
    assert b.batches(3, 20, 20) == [(3, 20)] and b.batches(0, 7, 5) == [(0, 4), (4, 3)] and b.batches(0, 0, 5) == [] and b.batches(0, 300, 256) == [(0, 150), (150, 150)]


def test_frame_plan_and_ticket_limits():
    """bm_frame_plan_of (host only): what the library decides for a frame.  ADVICE r05: (chunk, sample) items are the library's own choice
    only where their tickets fit the 32-bit hand-out -- a default-flag frame at very high spp falls back to pixel items (which carry no
    spp factor) instead of being refused; a caller who ASKS for sample items that do not fit is refused; unknown flags are refused."""
    import brickmap_amd as bm
    p1 = bm.frame_plan(bm.FrameParams(1920, 1080, spp=1, max_bounces=3))
    assert (p1["helpers"], p1["sample_items"], p1["ordered"], p1["xcd_handout"], p1["refill_min"], p1["instrumented"]) == (1, 0, 0, 0, 24, 0)
    assert p1["refill_min_in_ring"] == 32 and bm.frame_plan(bm.FrameParams(64, 64, spp=2, flags=bm.BM_FLAG_ORDERED))["refill_min_in_ring"] == 8  # (ordered: as a lone frame)
    p4 = bm.frame_plan(bm.FrameParams(1920, 1080, spp=4, max_bounces=3))
    assert p4["sample_items"] == 1 and p4["flags"] & bm.BM_FLAG_SAMPLE_ITEMS
    for W, H, spp in ((1920, 1080, 5000), (3840, 2160, 1000), (7680, 4320, 300)):  # (beyond the limits the advisor computed: 4100 / 950 / 250)
        hi = bm.frame_plan(bm.FrameParams(W, H, spp=spp, max_bounces=3))
        assert hi["sample_items"] == 0 and hi["helpers"] == 1 and not hi["flags"] & bm.BM_FLAG_SAMPLE_ITEMS
        with pytest.raises(bm.BrickmapError, match="ticket"):
            bm.frame_plan(bm.FrameParams(W, H, spp=spp, max_bounces=3, flags=bm.BM_FLAG_SAMPLE_ITEMS))
    assert bm.frame_plan(bm.FrameParams(3840, 2160, spp=4, max_bounces=7), hit_records=True)["ordered"] == 1  # hit records: chains in path order
    d = bm.frame_plan(bm.FrameParams(3840, 2160, spp=4, max_bounces=7, flags=bm.BM_FLAG_RAY_DIGEST), hit_records=True)
    assert (d["ordered"], d["helpers"], d["sample_items"], d["xcd_handout"], d["instrumented"]) == (0, 1, 1, 1, 1)  # ... unless the ray digest is asked for
    assert bm.frame_plan(bm.FrameParams(64, 64, spp=2, flags=bm.BM_FLAG_ORDERED))["ordered"] == 1
    with pytest.raises(bm.BrickmapError, match="unknown frame flag"):
        bm.frame_plan(bm.FrameParams(64, 64, flags=8))  # the retired K-slot schedule's bit
    with pytest.raises(bm.BrickmapError, match="RAY_DIGEST"):
        bm.frame_plan(bm.FrameParams(64, 64, spp=20000, max_bounces=3, flags=bm.BM_FLAG_RAY_DIGEST))
    assert bm.tuning_overrides() == {}


def test_retired_experiments_stay_out_of_the_kernel_sources():
    """tools/variants/*.patch are experiment RECORDS, exact against the round-4 tree (commit 683c02e; tools/variants/README.md): the
    kernel sources hold the product path only -- none of the retired experiment macros."""
    import glob
    patches = sorted(glob.glob(os.path.join(ROOT, "tools", "variants", "*.patch")))
    assert len(patches) >= 9
    retired = ("BM_JUMP_BINADES", "BM_LOD_PRETEST", "BM_NT_BRICKS", "BM_FIELD_BLOCKED", "BM_XCD_TILES", "BM_CMP3", "BM_B_STEP", "BM_ARGMAX", "BM_FLAG_TWIN", "coarse_field",
               "BM_FLAG_KSLOT", "trace_k", "OVERLAY")
    for f in glob.glob(os.path.join(ROOT, "brickmap_amd", "csrc", "*.h*")) + glob.glob(os.path.join(ROOT, "brickmap_amd", "csrc", "*.cpp")):
        text = open(f).read()
        for macro in retired:
            assert macro not in text, (os.path.basename(f), macro)


def test_variant_patches_apply_to_their_base_commits(tmp_path):
    """Every tools/variants/*.patch is listed in BASES.txt with the commit it is exact against, and applies to an export of that
    commit (the records must not go stale unnoticed).  Needs the repository's history: skipped on a bare snapshot."""
    import glob
    import subprocess
    if not os.path.isdir(os.path.join(ROOT, ".git")):
        pytest.skip("no git history in this snapshot")
    bases = {}
    for line in open(os.path.join(ROOT, "tools", "variants", "BASES.txt")):
        if line.strip() and not line.startswith("#"):
            name, commit = line.split()
            bases[name] = commit
    patches = sorted(os.path.basename(p) for p in glob.glob(os.path.join(ROOT, "tools", "variants", "*.patch")))
    assert patches == sorted(bases), "every patch has a base commit on record"
    for commit in sorted(set(bases.values())):
        if subprocess.run(["git", "cat-file", "-e", commit + "^{commit}"], cwd=ROOT).returncode != 0:
            pytest.skip(f"commit {commit} is not in this clone")
        tree = tmp_path / commit
        tree.mkdir()
        tar = subprocess.run(["git", "archive", commit, "brickmap_amd", "include", "tests", "tools", "bench.py"], cwd=ROOT, capture_output=True, check=True).stdout
        subprocess.run(["tar", "-x", "-C", str(tree)], input=tar, check=True)
        for name in patches:
            if bases[name] == commit:
                r = subprocess.run(["git", "apply", "--check", os.path.join(ROOT, "tools", "variants", name)], cwd=str(tree), capture_output=True, text=True)
                assert r.returncode == 0, (name, commit, r.stderr[-500:])


def test_scheduler_simulator_builds_and_runs(tmp_path, orc):
    """tools/sim/sched_sim.cpp (the model behind docs/HISTORY.md 5.5) stays buildable against the product's jump.h / world.cpp and the
    oracle, and on a sampled frame more paths per lane mean fuller passes."""
    import re
    import subprocess
    exe = tmp_path / "sched_sim"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "brickmap_amd", "csrc"), os.path.join(ROOT, "tools", "sim", "sched_sim.cpp"),
                           os.path.join(ROOT, "brickmap_amd", "csrc", "world.cpp"), "-L" + os.path.join(ROOT, "oracle"), "-l:liboracle.so",
                           "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-lpthread", "-o", str(exe)])
    out = subprocess.run([str(exe), "tiles=16", "cpi=3.0", "policy=2", "sweep=1:5:0:0:0:0:0:0:1", "sweep=3:3:0:0:20:20:30:40:1.3"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lanes = [float(x) for x in re.findall(r"C passes [0-9.]+M at ([0-9.]+) lanes", out.stdout)]
    paths = [int(x) for x in re.findall(r"paths (\d+) rays", out.stdout)]
    assert len(lanes) == 2 and len(paths) == 2 and paths[0] == paths[1] > 100000
    assert lanes[1] > lanes[0] + 5.0  # three paths per lane: shade passes are much fuller than with one


def test_bench_self_launch_command(monkeypatch):
    """`python bench.py --gpus N` without a launcher re-runs itself under torch.distributed.run: one rank per GPU, static rendezvous
    on 127.0.0.1 (the container's hostname may not resolve), the original arguments, HSA_ENABLE_IPC_MODE_LEGACY=0 for RCCL."""
    import importlib
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "5", "--warmup", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    try:
        bench.main()
        raise AssertionError("bench.main() should have exited through the launcher")
    except SystemExit as e:
        assert e.code == 7  # the launcher's exit code is the command's
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-6:] == ["--gpus", "4", "--steps", "5", "--warmup", "2"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # with WORLD_SIZE set (a rank started by a launcher) nothing is re-launched: main() goes on to import torch and the product
    seen.clear()
    monkeypatch.setenv("WORLD_SIZE", "3")
    try:
        bench.main()
    except SystemExit as e:
        assert "WORLD_SIZE=3" in str(e.code)
    except Exception:
        pass  # (no GPU here: anything after the launch decision may fail)
    assert not seen


def test_division_magic_is_exact():
    """The walk keeps a ray's cell as the byte offset of its cube-field entry and divides it by the slice pitch at candidates with one
    multiply-high (traverse.h cell_coords): floor(n / d) == (n * magic >> 32) >> shift must hold for EVERY offset n < 2^30.  Checked
    for the slice pitches of the BASELINE worlds and awkward divisors: all multiples of d and their neighbours (where a rounded-down
    reciprocal fails first), plus random offsets."""
    import ctypes as C
    from brickmap_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(5)
    for d in [(18 << 5), (34 << 6), (130 << 8), (258 << 9), (514 << 10), (50 << 6), 3, 5, 7, 641, 65537, (1 << 20), (1 << 23) - 1, 1000003]:
        magic, shift = C.c_uint32(0), C.c_int(0)
        assert L.bm_debug_division_magic(d, C.byref(magic), C.byref(shift)) == 0
        k = np.arange(0, (1 << 30) // d + 1, dtype=np.uint64)
        if len(k) > 2_000_000:
            k = np.unique(np.concatenate([k[:500_000], k[-500_000:], rng.integers(0, len(k), 1_000_000).astype(np.uint64)]))
        n = np.concatenate([k * d, k * d + 1, (k + 1) * d - 1, rng.integers(0, 1 << 30, 1_000_000).astype(np.uint64)])
        n = n[n < (1 << 30)]
        got = ((n * np.uint64(magic.value)) >> np.uint64(32)) >> np.uint64(shift.value)
        assert np.array_equal(got, n // np.uint64(d)), d
    assert L.bm_debug_division_magic(2, C.byref(magic), C.byref(shift)) != 0
