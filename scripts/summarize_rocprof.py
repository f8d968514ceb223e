#!/usr/bin/env python3
"""Turn rocprofv3 CSV output (gpurun_out/prof_*) into the summaries committed under profiles/.

  python scripts/summarize_rocprof.py ROUND WORKLOAD_KEY STATS_DIR FETCH_DIR WRITE_DIR

Writes profiles/rNN_kernel_stats.csv (verbatim rocprofv3 --kernel-trace --stats summary),
profiles/rNN_summary.md and profiles/hbm_traffic.json (read by bench.py for roofline.traffic).

HBM bytes per launch follow /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE
are collected in separate --pmc passes (they do not fit one pass), are in KiB, and on gfx950
FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read, so
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name: str) -> str:
    """'void eigenex::(anonymous namespace)::k_dots<false, true>(double const*, ...)' -> 'k_dots'"""
    name = name.strip()
    if name.startswith("void "):
        name = name[5:]
    name = name.replace("eigenex::(anonymous namespace)::", "").replace("eigenex::", "")
    for sep in ("<", "("):
        if sep in name:
            name = name.split(sep)[0]
    return name.strip()


def load_counter(d: str, counter: str):
    per = defaultdict(list)
    files = sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for f in files[-1:]:  # gpurun merges into existing directories: only the newest run of a pass counts
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                per[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    return per


def main():
    rnd, key, stats_dir, fetch_dir, write_dir = sys.argv[1:6]
    import subprocess
    try:
        head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
        dirty = bool(subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "cmpt-eigenex_amd/csrc", "bench.py"], text=True).strip())
    except Exception:
        head, dirty = "unknown", False
    if os.environ.get("PROFILED_COMMIT"):  # the summary is written after the run: name the commit that was on the GPU box
        head, dirty = os.environ["PROFILED_COMMIT"], False
    source = f"commit {head}{' + uncommitted changes' if dirty else ''}: kernels and bench.py as profiled"

    out_dir = os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    stats_file = sorted(glob.glob(os.path.join(stats_dir, "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)[-1]
    shutil.copy(stats_file, os.path.join(out_dir, f"{rnd}_kernel_stats.csv"))
    stats = {}
    for r in csv.DictReader(open(stats_file)):
        k = short(r["Name"])
        if k in stats:  # template instantiations of one kernel
            a = stats[k]
            calls = int(a["Calls"]) + int(r["Calls"])
            tot = float(a["TotalDurationNs"]) + float(r["TotalDurationNs"])
            a.update(Calls=str(calls), TotalDurationNs=str(tot), AverageNs=str(tot / calls),
                     Percentage=str(float(a["Percentage"]) + float(r["Percentage"])))
        else:
            stats[k] = dict(r)
    fetch = load_counter(fetch_dir, "FETCH_SIZE")
    write = load_counter(write_dir, "WRITE_SIZE")
    traffic = {}
    lines = [f"# rocprofv3 summary, round {rnd}: {key}", "", f"Source: {source}.", "",
             "Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline`",
             "PMC: separate passes `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` over the same command; "
             "HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 correction of MI355X_MICROARCH.md).", "",
             "| kernel | calls | avg ms | % time | HBM bytes/launch (PMC) | fetch KiB avg (raw) | write KiB avg |", "|---|---|---|---|---|---|---|"]
    for k, r in sorted(stats.items(), key=lambda kv: -float(kv[1]["TotalDurationNs"])):
        f = fetch.get(k, [])
        w = write.get(k, [])
        bpl = None
        if f and w:
            bpl = (2.0 * sum(f) / len(f) + sum(w) / len(w)) * 1024.0
            traffic[k] = {"bytes_per_launch": bpl, "fetch_kib_avg_raw": sum(f) / len(f), "write_kib_avg": sum(w) / len(w),
                          "launches_profiled": len(f)}
        lines.append(f"| {k} | {r['Calls']} | {float(r['AverageNs'])/1e6:.4f} | {r['Percentage']} | "
                     f"{bpl:.4g} |" f" {sum(f)/len(f):.4g} | {sum(w)/len(w):.4g} |" if bpl else
                     f"| {k} | {r['Calls']} | {float(r['AverageNs'])/1e6:.4f} | {r['Percentage']} | - | - | - |")
    open(os.path.join(out_dir, f"{rnd}_summary.md"), "w").write("\n".join(lines) + "\n")
    tfile = os.path.join(out_dir, "hbm_traffic.json")
    allt = json.load(open(tfile)) if os.path.exists(tfile) else {}
    traffic["__source__"] = f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, {source}"
    traffic["__commit__"] = head
    sys.path.insert(0, ROOT)
    from cmpt_eigenex_amd.build import source_fingerprint
    traffic["__sources_sha16__"] = source_fingerprint()  # bench.py recomputes it: roofline.traffic_is_current
    allt[key] = traffic
    json.dump(allt, open(tfile, "w"), indent=1, sort_keys=True)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
