"""Re-base the retired-experiment patches after the kernel sources moved: every tools/variants/*.patch that no longer applies to HEAD
but applied to <base> (default HEAD~1) is applied there in a scratch worktree, its files are three-way merged onto HEAD's
(git merge-file), and the patch is written again as the difference to HEAD.  Conflicts are reported and left for a hand merge.
usage: python tools/variants/rebase_patches.py [base-commit]"""
import glob, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
base = sys.argv[1] if len(sys.argv) > 1 else "HEAD~1"


def git(*a, cwd=ROOT, check=True):
    return subprocess.run(["git", *a], cwd=cwd, capture_output=True, text=True, check=check)


assert not git("status", "--porcelain", "--untracked-files=no").stdout.strip(), "commit or stash first: the patches are written against HEAD"
wt = tempfile.mkdtemp(prefix="bm_rebase_")
git("worktree", "add", "--detach", wt, base)
try:
    for p in sorted(glob.glob(os.path.join(ROOT, "tools", "variants", "*.patch"))):
        name = os.path.basename(p)
        if git("apply", "--check", p, check=False).returncode == 0:
            continue
        git("checkout", "-q", ".", cwd=wt); git("clean", "-fdq", cwd=wt)
        if git("apply", p, cwd=wt, check=False).returncode != 0:
            print(f"{name}: does not apply to {base} either -- skipped"); continue
        modified = git("diff", "--name-only", cwd=wt).stdout.split()
        new = git("ls-files", "-o", "--exclude-standard", cwd=wt).stdout.split()
        conflicts = []
        for f in modified:
            ours, theirs = os.path.join(ROOT, f), os.path.join(wt, f)
            basef = tempfile.NamedTemporaryFile(delete=False).name
            open(basef, "w").write(git("show", f"{base}:{f}").stdout)
            r = subprocess.run(["git", "merge-file", "-p", ours, basef, theirs], capture_output=True, text=True)
            os.unlink(basef)
            if r.returncode != 0:
                conflicts.append(f)
            open(ours, "w").write(r.stdout)
        for f in new:
            os.makedirs(os.path.dirname(os.path.join(ROOT, f)) or ROOT, exist_ok=True)
            shutil.copy(os.path.join(wt, f), os.path.join(ROOT, f))
        if new:
            git("add", "-N", *new)
        diff = git("diff", "HEAD").stdout
        if new:
            git("reset", "-q", "--", *new)
            for f in new:
                os.unlink(os.path.join(ROOT, f))
        git("checkout", "-q", "--", *modified)
        if conflicts:
            print(f"{name}: CONFLICTS in {conflicts} -- left as it was"); continue
        open(p, "w").write(diff)
        print(f"{name}: re-based")
finally:
    git("worktree", "remove", "--force", wt)
