"""Basic blocks of a kernel (compiler .s output) with the most v_mov_b32: where the register allocator shuffles state.
usage: isa_movs.py <file.s> <kernel name substring>"""
import collections, re, sys
lines = open(sys.argv[1]).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sys.argv[2] in l.split(":")[0] and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
blk, movs, tot, first = "entry", collections.Counter(), collections.Counter(), {"entry": start + 1}
for i in range(start, end):
    m = re.match(r"^(\.LBB\d+_\d+):", lines[i])
    if m:
        blk = m.group(1); first[blk] = i + 1
    t = lines[i].split(";")[0].strip()
    if t.startswith("v_"):
        tot[blk] += 1
    if t.startswith("v_mov_b32") or t.startswith("v_mov_b64"):
        movs[blk] += 1
print("total movs", sum(movs.values()), "of", sum(tot.values()), "VALU")
for b, c in movs.most_common(14):
    print(f"{b:12s} line {first[b]:6d}: {c:3d} movs of {tot[b]:3d} VALU")
