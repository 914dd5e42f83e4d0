#!/usr/bin/env python3
"""Run tools/ubench/fetch_calib under rocprofv3 (one --pmc pass per counter group, kernel-trace only) on the GPU box and
print, per dispatch, the counters next to the known unique bytes:  python tools/fetch_calib.py > gpurun_out/r03_fetch_calibration.txt"""
import collections
import csv
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "scratch", "fetch_calib")
GROUPS = ["FETCH_SIZE", "TCC_HIT TCC_MISS TCC_REQ TCC_READ", "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B", "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum",
          "TCC_BUBBLE TCC_EA0_RD_UNCACHED_32B", "TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES", "GRBM_GUI_ACTIVE"]


def main():
    env = dict(os.environ, TMPDIR="/tmp")
    base = subprocess.run([EXE, "2"], capture_output=True, text=True, cwd="/tmp", env=env)
    rows = []
    for line in base.stdout.splitlines():
        if line.startswith("DISPATCH"):
            t = line.split()
            rows.append({"id": int(t[1]), "kernel": t[2], "MiB": int(t[4]), "rep": int(t[6]), "useful": int(t[8]), "lines128": int(t[10]),
                         "halves64": int(t[12]), "ms": float(t[14])})
    if not rows:
        print(base.stdout[-2000:], base.stderr[-2000:])
        raise SystemExit("fetch_calib did not run")
    counters = collections.defaultdict(dict)  # dispatch order -> counter -> value
    for gi, group in enumerate(GROUPS):
        d = f"/tmp/fetch_calib_{gi}"
        shutil.rmtree(d, ignore_errors=True)
        r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc"] + group.split() + ["--output-format", "csv", "-d", d, "-o", "p", "--", EXE, "2"],
                           capture_output=True, text=True, cwd="/tmp", env=env)
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            print(f"# group '{group}': no counter file ({(r.stderr or r.stdout)[-300:].strip()!r})")
            continue
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for f in files:
            for row in csv.DictReader(open(f)):
                if not any(k in row["Kernel_Name"] for k in ("stream16", "gather", "rows_b1")):
                    continue  # the fill kernel of hipMemset and friends are dispatches too
                per[int(row["Dispatch_Id"])][row["Counter_Name"]] += float(row["Counter_Value"])
        assert len(per) in (0, len(rows)), f"group '{group}': {len(per)} profiled dispatches for {len(rows)} launches"
        for order, did in enumerate(sorted(per)):
            counters[order].update(per[did])
    names = sorted({c for v in counters.values() for c in v})
    print("# per dispatch: known bytes vs counters.  FETCH_SIZE in KiB as reported; x64 / x128 columns = counter * 64 B or 128 B over the known unique bytes")
    print("# " + " ".join(names))
    for r in rows:
        c = counters.get(r["id"], {})
        fs = c.get("FETCH_SIZE")
        miss = c.get("TCC_MISS")
        rd = c.get("TCC_EA0_RDREQ", c.get("TCC_EA0_RDREQ_sum"))
        rd32 = c.get("TCC_EA0_RDREQ_32B", c.get("TCC_EA0_RDREQ_32B_sum"))
        line = (f"{r['kernel']:15s} {r['MiB']:5d} MiB rep {r['rep']}  {r['ms']:9.3f} ms  useful {r['useful'] / 1e6:10.1f} MB  unique 128B-lines {r['lines128'] * 128 / 1e6:10.1f} MB  "
                f"unique 64B-halves {r['halves64'] * 64 / 1e6:10.1f} MB |")
        if fs is not None:
            b = fs * 1024
            line += f" FETCH_SIZE {b / 1e6:10.1f} MB = {b / (r['halves64'] * 64):.3f} x halves = {b / (r['lines128'] * 128):.3f} x lines = {b / r['useful']:.3f} x useful, {b / r['ms'] / 1e6:.0f} GB/s |"
        if miss is not None:
            line += f" TCC_MISS {miss / 1e6:.2f} M ({miss / r['halves64']:.3f} per half, {miss / r['lines128']:.3f} per line) REQ {c.get('TCC_REQ', 0) / 1e6:.2f} M HIT {c.get('TCC_HIT', 0) / 1e6:.2f} M |"
        if rd is not None:
            line += f" EA_RDREQ {rd / 1e6:.2f} M ({rd / r['halves64']:.3f} per half) 32B {0 if rd32 is None else rd32 / 1e6:.2f} M |"
        for extra in ("TCC_BUBBLE", "TCC_EA0_RD_UNCACHED_32B", "TCP_TCC_READ_REQ"):
            if extra in c:
                line += f" {extra} {c[extra] / 1e6:.2f} M"
        print(line)


if __name__ == "__main__":
    main()
