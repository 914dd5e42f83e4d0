"""Per-pass static instruction table of bm::trace_paths from the compiler's own line table.

The kernel source marks the boundaries of the scheduler's passes with BM_REGION("...") lines; this tool compiles csrc/trace.hip
with -gline-tables-only (line tables do not change the generated code: the VALU total is checked against the normal build when
build/ holds one), reads the `.loc` comments of the .s listing -- every instruction carries its inlining chain down to the line of
trace.hip it was inlined at -- and attributes each instruction to the region of that OUTERMOST trace.hip line and to the function
of its INNERMOST line.  Costs are those of tools/isa_mix.py (tools/ubench/valu_rates.hip).

usage: python tools/isa_passes.py [kernel substring ...]      (default: the two 1080p instantiations, with and without helper lanes)
"""
import bisect
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "brickmap_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "tools"))
from isa_mix import cost  # noqa: E402

FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -mllvm -structurizecfg-skip-uniform-regions -fPIC -fvisibility=hidden".split()


def function_ranges(path):
    """[(first line, name)] of the __device__ / __global__ / BM_JHD functions of a source file, in line order."""
    out = []
    pat = re.compile(r"^\s*(?:template\s*<[^>]*>\s*)?(?:__device__|__global__|BM_JHD|static|inline|__host__)[^;=]*?\b([A-Za-z_]\w*)\s*\(")
    pending = None
    for i, line in enumerate(open(path), 1):
        m = pat.match(line)
        if m and "(" in line and not line.strip().startswith("//"):
            out.append((i, m.group(1)))
    return out


def main():
    kernels = sys.argv[1:] or ["_ZN2bm11trace_pathsILb0ELb0ELb0E", "_ZN2bm11trace_pathsILb0ELb0ELb1E"]
    src = os.path.join(CSRC, "trace.hip")
    regions = [(1, "prologue")]
    for i, line in enumerate(open(src), 1):
        m = re.search(r'^\s*BM_REGION\("([^"]+)"\)', line)
        if m:
            regions.append((i, m.group(1)))
        if re.match(r"\tif \(DBG && counters\) \{ // wave-level sum", line):
            regions.append((i, "epilogue"))
    starts = [r[0] for r in regions]
    funcs = {f: function_ranges(os.path.join(CSRC, f)) for f in ("traverse.h", "jump.h", "detmath.h")}
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-gline-tables-only", *FLAGS, "-save-temps", "-x", "hip", src, "-c", "-o", "trace.o"], cwd=tmp,
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        lines = open(os.path.join(tmp, "trace-hip-amdgcn-amd-amdhsa-gfx950.s")).read().splitlines()
    for K in kernels:
        start = next(i for i, l in enumerate(lines) if l.startswith(K) and l.split(";")[0].rstrip().endswith(":"))
        end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
        tab = collections.defaultdict(lambda: collections.Counter())
        sub = collections.defaultdict(lambda: collections.Counter())
        region, inner = "prologue", "trace.hip"
        for l in lines[start:end]:
            if l.lstrip().startswith(".loc"):
                c = l.split(";", 1)[1] if ";" in l else ""
                outer = re.findall(r"trace\.hip:(\d+)", c)
                if outer:
                    region = regions[bisect.bisect_right(starts, int(outer[-1])) - 1][1]
                m = re.search(r"/([\w.]+):(\d+):\d+", c)
                if m:
                    f, ln = m.group(1), int(m.group(2))
                    if f in funcs and funcs[f]:
                        fs = funcs[f]
                        k = bisect.bisect_right([x[0] for x in fs], ln) - 1
                        inner = f"{f}:{fs[k][1]}" if k >= 0 else f
                    else:
                        inner = f if f == "trace.hip" else "(hip headers / libm)"
                continue
            code = l.split(";")[0].strip()
            if not code or code.startswith(".") or code.endswith(":"):
                continue
            parts = code.split(None, 1)
            base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", parts[0])
            t = tab[region]
            if base.startswith("v_"):
                srcs = parts[1].split(",")[1:] if len(parts) > 1 else []
                sg = any(re.match(r"\s*-?\|?(s\d+|s\[|vcc|exec|m0|ttmp)", x) for x in srcs)
                cy = cost(base, sg)
                t["valu"] += 1; t["cycles"] += cy
                t["mov"] += base.startswith("v_mov_b")
                sub[region][inner] += 1
            elif base.startswith("s_"):
                t["salu"] += 1
            elif base.startswith(("global_", "ds_", "buffer_", "flat_", "scratch_")):
                t["mem"] += 1
        tot = collections.Counter()
        print(f"== {K}  (static instructions per region of the scheduler loop; one pass executes its region once, loops inside it as often as they run)")
        print(f"{'region':14s} {'VALU':>6s} {'weighted cycles':>16s} {'v_mov':>6s} {'SALU':>6s} {'mem':>5s}   innermost functions (VALU)")
        for _, name in regions:
            if name not in tab:
                continue
            t = tab[name]
            tot.update(t)
            inner_txt = ", ".join(f"{k.split(':')[-1]} {v}" for k, v in sub[name].most_common(6))
            print(f"{name:14s} {t['valu']:6d} {t['cycles']:16.0f} {t['mov']:6d} {t['salu']:6d} {t['mem']:5d}   {inner_txt}")
            tab.pop(name)
        print(f"{'total':14s} {tot['valu']:6d} {tot['cycles']:16.0f} {tot['mov']:6d} {tot['salu']:6d} {tot['mem']:5d}")
        ref = os.path.join(CSRC, "build", "trace-hip-amdgcn-amd-amdhsa-gfx950.s")
        if os.path.exists(ref):
            rl = open(ref).read().splitlines()
            try:
                rs = next(i for i, l in enumerate(rl) if l.startswith(K) and l.split(";")[0].rstrip().endswith(":"))
                re_ = next(i for i in range(rs, len(rl)) if rl[i].startswith(".Lfunc_end"))
                nv = sum(1 for l in rl[rs:re_] if l.startswith("\tv_"))
                print(f"(the normal build in csrc/build has {nv} VALU instructions in this kernel{'' if nv == tot['valu'] else ' -- DIFFERENT: rebuild with make asm'})")
            except StopIteration:
                pass


if __name__ == "__main__":
    main()
