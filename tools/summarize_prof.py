#!/usr/bin/env python3
"""Condense rocprofv3 output under gpurun_out/ into the tracked summaries under profiles/.

    python tools/summarize_prof.py r01 gpurun_out/prof_r1 gpurun_out/pmc_fetch gpurun_out/pmc_write

  profiles/<round>_kernel_stats.csv   copy of rocprofv3 --kernel-trace --stats (per-kernel calls / avg ns)
  profiles/<round>_pmc_summary.json   per kernel: launches, mean FETCH_SIZE / WRITE_SIZE (KB, raw) and the HBM
                                      bytes per launch with the gfx950 correction of MI355X_MICROARCH.md
                                      (FETCH_SIZE counts 128-B requests as 64 B: doubled; WRITE_SIZE exact)
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0][:90]


def pmc(dirname):
    files = glob.glob(os.path.join(dirname, "**", "*_counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(lambda: [0, 0.0])
    counter = None
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"])
                agg[k][0] += 1
                agg[k][1] += float(row["Counter_Value"])
                counter = row["Counter_Name"]
    return counter, agg


def main():
    tag, stats_dir, fetch_dir, write_dir = sys.argv[1:5]
    config = sys.argv[5] if len(sys.argv) > 5 else ("c4" if "_c4_" in tag else "c2")
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    ks = glob.glob(os.path.join(stats_dir, "**", "*_kernel_stats.csv"), recursive=True)
    if ks:
        shutil.copy(ks[0], os.path.join(out, f"{tag}_kernel_stats.csv"))
    _, fa = pmc(fetch_dir)
    _, wa = pmc(write_dir)
    summary = {}
    for k in sorted(set(fa) | set(wa), key=lambda k: -(fa.get(k, [0, 0])[1] + wa.get(k, [0, 0])[1])):
        nf, vf = fa.get(k, [0, 0.0])
        nw, vw = wa.get(k, [0, 0.0])
        n = max(nf, nw)
        if n == 0:
            continue
        f_kb, w_kb = (vf / nf if nf else 0.0), (vw / nw if nw else 0.0)
        summary[k] = {"launches": n, "fetch_size_kb_mean_raw": f_kb, "write_size_kb_mean": w_kb,
                      "hbm_bytes_per_launch_corrected": (2.0 * f_kb + w_kb) * 1024.0}
    with open(os.path.join(out, f"{tag}_pmc_summary.json"), "w") as fh:
        json.dump({"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of "
                           "bench.py --steps 1 --warmup 1 --serial-streams (config " + config + "); FETCH_SIZE doubled per MI355X_MICROARCH.md",
                   "kernels": summary}, fh, indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main()
