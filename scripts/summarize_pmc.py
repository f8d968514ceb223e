#!/usr/bin/env python3
"""Per-kernel table from the passes of scripts/profile_passes.sh.

  python scripts/summarize_pmc.py NAME [--out profiles/rNN_NAME.md] [--title "..."] [--kernels k_spmv,k_dots]

Reads gpurun_out/prof_NAME_stats (rocprofv3 --kernel-trace --stats) and every gpurun_out/prof_NAME_pmc*
(one --pmc group each), and prints / writes a Markdown table: calls, average duration, and the average value per launch
of every collected counter (summed over the counter's dimensions, i.e. XCDs / channels).  Derived columns:
  HBM bytes/launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024   (gfx950 correction, /opt/skills/guides/MI355X_MICROARCH.md "HBM")
  L2 hit rate      = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
The kernel-stats CSV is copied next to the table (rNN_NAME_kernel_stats.csv)."""
import csv
import glob
import os
import shutil
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name: str) -> str:
    name = name.strip()
    if name.startswith("void "):
        name = name[5:]
    name = name.replace("eigenex::(anonymous namespace)::", "").replace("eigenex::", "")
    for sep in ("<", "("):
        if sep in name:
            name = name.split(sep)[0]
    return name.strip()


def load_counters(d):
    """{kernel: {counter: [value per dispatch]}}; values of one dispatch are summed over dimensions"""
    per = defaultdict(lambda: defaultdict(dict))
    files = sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for f in files[-1:]:  # gpurun merges into existing directories: only the newest run of a pass counts
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            disp = row.get("Dispatch_Id") or row.get("Correlation_Id")
            c = row["Counter_Name"]
            per[k][c][disp] = per[k][c].get(disp, 0.0) + float(row["Counter_Value"])
    return {k: {c: list(v.values()) for c, v in cs.items()} for k, cs in per.items()}


def main():
    args = sys.argv[1:]
    name = args[0]
    out = args[args.index("--out") + 1] if "--out" in args else None
    title = args[args.index("--title") + 1] if "--title" in args else name
    only = args[args.index("--kernels") + 1].split(",") if "--kernels" in args else None
    base = os.path.join(ROOT, "gpurun_out")
    stats_files = sorted(glob.glob(os.path.join(base, f"prof_{name}_stats", "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)[-1:]
    stats = {}
    if stats_files:
        for r in csv.DictReader(open(stats_files[0])):
            k = short(r["Name"])
            a = stats.setdefault(k, dict(calls=0, total=0.0))
            a["calls"] += int(r["Calls"])
            a["total"] += float(r["TotalDurationNs"])
    counters = defaultdict(dict)
    groups = []
    for d in sorted(glob.glob(os.path.join(base, f"prof_{name}_pmc*"))):
        if not os.path.isdir(d):
            continue
        got = load_counters(d)
        names = sorted({c for cs in got.values() for c in cs})
        groups.append((os.path.basename(d), names))
        for k, cs in got.items():
            for c, vals in cs.items():
                counters[k][c] = (sum(vals) / len(vals), len(vals))
    allc = []
    for _, names in groups:
        allc += [c for c in names if c not in allc]
    try:
        head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        head = "unknown"
    head = os.environ.get("PROFILED_COMMIT", head)
    lines = [f"# {title}", "",
             f"Source tree: commit `{head}` (+ working tree at the time of the run). rocprofv3 passes, each its own run: "
             f"`--kernel-trace --stats`" + "".join(f"; `--pmc {' '.join(n)}`" for _, n in groups) + ".",
             "Counter values are averages per launch, summed over the counter's instances (XCDs / channels).", ""]
    cols = ["kernel", "calls", "avg us"] + allc
    derived = []
    if "FETCH_SIZE" in allc and "WRITE_SIZE" in allc:
        derived.append("HBM bytes/launch")
    if "TCC_HIT_sum" in allc and "TCC_MISS_sum" in allc:
        derived.append("L2 hit rate")
    cols += derived
    lines.append("| " + " | ".join(cols) + " |")
    lines.append("|" + "---|" * len(cols))
    kernels = sorted(set(stats) | set(counters), key=lambda k: -stats.get(k, dict(total=0))["total"])
    for k in kernels:
        if only and k not in only:
            continue
        s = stats.get(k)
        row = [k, str(s["calls"]) if s else "-", f"{s['total'] / s['calls'] / 1e3:.2f}" if s else "-"]
        cv = counters.get(k, {})
        for c in allc:
            row.append(f"{cv[c][0]:.5g}" if c in cv else "-")
        if "HBM bytes/launch" in derived:
            row.append(f"{(2 * cv['FETCH_SIZE'][0] + cv['WRITE_SIZE'][0]) * 1024:.5g}" if "FETCH_SIZE" in cv and "WRITE_SIZE" in cv else "-")
        if "L2 hit rate" in derived:
            h, m = cv.get("TCC_HIT_sum", (0, 0))[0], cv.get("TCC_MISS_sum", (0, 0))[0]
            row.append(f"{h / (h + m):.3f}" if h + m > 0 else "-")
        lines.append("| " + " | ".join(row) + " |")
    text = "\n".join(lines) + "\n"
    print(text)
    if out:
        outp = os.path.join(ROOT, out)
        open(outp, "w").write(text)
        if stats_files:
            shutil.copy(stats_files[0], outp[:-3] + "_kernel_stats.csv")


if __name__ == "__main__":
    main()
