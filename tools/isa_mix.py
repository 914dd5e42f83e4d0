"""Static instruction mix of a kernel in the compiler's .s output, weighted with the issue costs measured by
tools/ubench/valu_rates.hip (SIMD cycles per wave64 instruction on gfx950): ~2.5 for the plain VOP2 float / integer / logic
operations on VGPR operands, ~4.8 for everything else (min/max, compares, selects, conversions, shifts left, bit-field
and multiply-add forms, any VALU operation with an SGPR source), ~8.6 for the transcendental unit and v_swap.
usage: isa_mix.py <file.s> <kernel name substring> [first_line last_line]"""
import collections
import re
import sys

FAST = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32",
        "v_or_b32", "v_xor_b32", "v_not_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32", "v_mul_legacy_f32", "v_fmaak_f32", "v_fmamk_f32"}
TRANS = {"v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_swap_b32", "v_rcp_iflag_f32"}


def cost(base, sgpr_src):
    if base in TRANS:
        return 8.6
    if base in FAST and not sgpr_src:
        return 2.6
    return 4.8


def main():
    path, name = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and name in l.split(":")[0] and l.rstrip().endswith(")") is False and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    if len(sys.argv) > 4:
        start, end = int(sys.argv[3]) - 1, int(sys.argv[4])
    cnt, cyc, sg = collections.Counter(), collections.Counter(), collections.Counter()
    for line in lines[start:end]:
        line = line.split(";")[0].strip()
        if not line or line.startswith(".") or line.endswith(":"):
            continue
        parts = line.split(None, 1)
        base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", parts[0])
        if not base.startswith("v_") or base.startswith(("v_readlane", "v_readfirstlane")) and False:
            cnt[base] += 1
            continue
        srcs = parts[1].split(",")[1:] if len(parts) > 1 else []
        sgpr_src = any(re.match(r"\s*-?\|?(s\d+|s\[|vcc|exec|m0|ttmp)", x) for x in srcs)
        cnt[base] += 1
        cyc[base] += cost(base, sgpr_src)
        if sgpr_src:
            sg[base] += 1
    valu = sum(c for o, c in cnt.items() if o.startswith("v_"))
    salu = sum(c for o, c in cnt.items() if o.startswith("s_"))
    total = sum(cyc.values())
    print(f"lines {start + 1}-{end}: VALU {valu} ({total:.0f} weighted cycles, {total / max(valu, 1):.2f} per instruction), SALU {salu}, "
          f"VALU with an SGPR source {sum(sg.values())}")
    for o, c in sorted(cyc.items(), key=lambda kv: -kv[1])[:45]:
        print(f"  {o:26s} x{cnt[o]:5d}  {c:8.0f} cycles {100 * c / total:5.1f} %   sgpr-src {sg.get(o, 0)}")


if __name__ == "__main__":
    main()
