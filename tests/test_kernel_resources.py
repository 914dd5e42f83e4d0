"""Guard rails for the production kernel's compiled shape (CPU only: hipcc cross-compiles gfx950 without a GPU).

bm::trace_paths<false, ...> is bound by vector-instruction issue at 7 waves per SIMD.  Two things
silently cost 5-10 % and have both happened: a register budget one step too high (one wave per SIMD less, or spills), and the backend
linearising the scheduler loop's scalar branches again, which keeps every lane's state in two register sets and copies
one onto the other around every pass (a few hundred extra v_mov; a small change to the loop's control flow is enough).
"""
import collections
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "brickmap_amd", "csrc")
# trace_paths<instrumented = false, XCD-aware hand-out, helper lanes, frame ring>: the twelve production instantiations (the default of
# production frames is <false, *, true, *>; BM_FLAG_ORDERED frames run <false, *, false, *>; frame ring: 0 = a launch of one frame, 1 =
# several frames, waves change frame when idle, 2 = several frames of a uniform launch -- bm_render_frames)
KERNELS = tuple("_ZN2bm11trace_pathsILb0ELb%dELb%dELi%dE" % (x, h, r) for r in (0, 2, 1) for h in (1, 0) for x in (0, 1))

import pytest


@pytest.fixture(scope="module")
def listing():
    """the compiler's resource report and .s listing of csrc/trace.hip (one compile for all cases)"""
    subprocess.check_call(["make", "-s", "-C", CSRC, "asm"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return os.path.join(CSRC, "build")


@pytest.mark.parametrize("KERNEL", KERNELS)
def test_production_kernel_keeps_its_register_budget_and_its_shape(KERNEL, listing):
    usage = open(os.path.join(CSRC, "build", "resource_usage.txt")).read()
    block = usage[usage.index(KERNEL):]
    block = block[:block.index("Function Name", 10)] if "Function Name" in block[10:] else block

    def field(name):
        return int(re.search(name + r": (\d+)", block).group(1))

    ring = int(KERNEL[-2])
    helpers = KERNEL[:-4].endswith("Lb1E")  # the default of production frames; the ordered instantiations keep an accumulator (4 more registers)
    assert field("VGPRs") <= 72 and field(r"Occupancy \[waves/SIMD\]") == 7, "more than 72 VGPRs: 6 waves per SIMD instead of 7"
    # what is spilled at seven waves are two loop-invariant constants of a cold branch (rays that start outside the world) in the
    # helper-lane instantiations; the ordered ones spill two more
    # (the ordered instantiation of a multi-frame launch -- a verification path, never the timed one -- spills one word more)
    assert field("VGPRs Spill") <= ((2 if ring != 1 else 3) if helpers else 5) and field(r"ScratchSize \[bytes/lane\]") <= ((8 if ring != 1 else 12) if helpers else (20 if ring else 16))
    assert field(r"LDS Size \[bytes/block\]") == 16384
    # the scheduler loop runs at the limit of the scalar file: a spilled scalar is a v_readlane / v_writelane in the hot loop.  The
    # frame ring's extra loop-carried scalar costs a few (which is why single-frame launches have an instantiation of their own):
    # its first version -- frame count, first frame's buffers and the round budget live across the loop -- had 25 and was 4 % slower
    # (and a test of the frame's constants in the loop's exit condition -- a scalar load and a wait in every scheduler round -- was its 2.6 %:
    # whether there is a next frame is settled when the ticket counters run dry)
    assert field("SGPRs Spill") <= {0: 12, 2: 18, 1: 20}[ring]
    lines = open(os.path.join(CSRC, "build", "trace-hip-amdgcn-amd-amdhsa-gfx950.s")).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith(KERNEL) and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    ops = collections.Counter(l.split(";")[0].split()[0] for l in lines[start:end] if l.startswith("\t") and l.split(";")[0].strip())
    valu = sum(c for o, c in ops.items() if o.startswith("v_"))
    movs = sum(c for o, c in ops.items() if o.startswith("v_mov_b"))
    packed = sum(c for o, c in ops.items() if o.startswith("v_pk_"))
    assert movs <= 275, f"{movs} register copies in {valu} VALU instructions: the scheduler loop was structurized again (tools/isa_movs.py)"
    flat = sum(c for o, c in ops.items() if o.startswith("flat_"))
    assert flat == 0, "flat_* memory instructions: a buffer pointer read from the frame's constants is used as a generic pointer (trace.hip pixel_atomic_add)"
    assert packed == 0, "packed fp32 operations: SLP vectorisation is back (they cost two plain operations each and pair registers)"
    assert valu <= 2100, f"{valu} VALU instructions"
