"""One kernel source, variants as patches.
Splits the default-off experiment paths out of brickmap_amd/csrc/{traverse.h,trace.hip,scene.cpp}: rewrites the sources without
them and writes one patch per experiment (clean source -> source with that experiment's #if blocks) under tools/variants/.
Run once when an experiment is retired from the tree; tools/build_variants.sh applies the patches again (`+name` arguments).
A small unifdef: evaluates #if / #elif lines that mention ONLY retired macros (at their default values), keeps everything else."""
import difflib, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
FILES = ["brickmap_amd/csrc/traverse.h", "brickmap_amd/csrc/trace.hip", "brickmap_amd/csrc/scene.cpp"]
EXPERIMENTS = {  # name -> {macro: default}
    "jump_binades": {"BM_JUMP_BINADES": 1},
    "lod_pretest": {"BM_LOD_PRETEST": 0},
    "nt_bricks": {"BM_NT_BRICKS": 0},
    "field_blocked": {"BM_FIELD_BLOCKED": 0},
    "cmp3": {"BM_CMP3": 0},
    "xcd_tiles": {"BM_XCD_TILES": 0, "BM_XCD_HWID": 0},
    "b_step": {"BM_B_STEP": 0},
    "argmax": {"BM_ARGMAX_WB": None, "BM_ARGMAX_WC": None},  # (#ifdef-style: undefined by default)
}


def strip(text, retired):
    """Remove the conditional code of the macros in `retired` (dict macro -> default value or None = undefined)."""
    out, stack = [], []  # stack entries: [kind, emitting_before, taken_already, active] ; kind 'keep' = directive kept verbatim
    lines = text.split("\n")
    i = 0

    def active():
        return all(s[3] for s in stack)

    def evaluate(expr):
        names = set(re.findall(r"[A-Za-z_]\w*", expr)) - {"defined"}
        if not names or not names <= set(retired):
            return None
        e = expr
        for m, v in retired.items():
            e = re.sub(r"defined\s*\(\s*%s\s*\)" % m, "1" if v is not None else "0", e)
            e = re.sub(r"\b%s\b" % m, str(v if v is not None else 0), e)
        e = e.replace("&&", " and ").replace("||", " or ").replace("!", " not ")
        return bool(eval(e))

    while i < len(lines):
        line = lines[i]
        s = line.strip()
        m = re.match(r"#\s*(ifndef|ifdef|if|elif|else|endif)\b(.*)", s)
        if not m:
            if active():
                out.append(line)
            i += 1
            continue
        d, rest = m.group(1), m.group(2).split("//")[0].strip()
        if d == "ifndef" and rest in retired and i + 2 < len(lines) and lines[i + 1].strip().startswith("#define " + rest) and lines[i + 2].strip() == "#endif":
            i += 3  # the default-value block of a retired macro
            continue
        if d in ("if", "ifdef", "ifndef"):
            val = None
            if d == "if":
                val = evaluate(rest)
            elif rest in retired:
                val = (retired[rest] is not None) if d == "ifdef" else (retired[rest] is None)
            if val is None:
                stack.append(["keep", None, None, True])
                if active():
                    out.append(line)
            else:
                stack.append(["eval", None, val, val])
        elif d == "elif":
            top = stack[-1]
            if top[0] == "keep":
                if active():
                    out.append(line)
            else:
                val = evaluate(rest)
                if val is None:
                    raise SystemExit("mixed #elif not supported: " + line)
                top[3] = (not top[2]) and val
                top[2] = top[2] or val
        elif d == "else":
            top = stack[-1]
            if top[0] == "keep":
                if active():
                    out.append(line)
            else:
                top[3] = not top[2]
                top[2] = True
        else:  # endif
            top = stack.pop()
            if top[0] == "keep" and active():
                out.append(line)
        i += 1
    assert not stack
    return "\n".join(out)


def main():
    all_retired = {}
    for macros in EXPERIMENTS.values():
        all_retired.update(macros)
    # --from DIR: the sources that still hold the experiments (basename lookup); the tree's own files are then taken as the clean
    # version as they are (hand-tidied after the mechanical split), so the patches are exact against what is committed
    src_dir = sys.argv[sys.argv.index("--from") + 1] if "--from" in sys.argv else None
    sources = {f: open(os.path.join(src_dir, os.path.basename(f)) if src_dir else os.path.join(ROOT, f)).read() for f in FILES}
    clean = {f: (open(os.path.join(ROOT, f)).read() if src_dir else strip(t, all_retired)) for f, t in sources.items()}
    for name, macros in EXPERIMENTS.items():
        others = {m: v for m, v in all_retired.items() if m not in macros}
        patch = []
        for f in FILES:
            with_x = strip(sources[f], others)
            if with_x != clean[f]:
                patch += list(difflib.unified_diff(clean[f].split("\n"), with_x.split("\n"), "a/" + f, "b/" + f, lineterm=""))
        open(os.path.join(ROOT, "tools/variants", name + ".patch"), "w").write("\n".join(patch) + "\n")
        print(name, len(patch), "patch lines")
    if "--write" in sys.argv:
        for f, t in clean.items():
            open(os.path.join(ROOT, f), "w").write(t)
        print("sources rewritten")


if __name__ == "__main__":
    main()
