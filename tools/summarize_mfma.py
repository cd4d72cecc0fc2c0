#!/usr/bin/env python3
"""Condense a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
GRBM_GUI_ACTIVE --kernel-trace` run into profiles/<tag>_mfma_busy_summary.json.

    python tools/summarize_mfma.py r02_bf16x3 gpurun_out/p_mfma

Per kernel (launch-weighted means): duration, MFMA-busy cycles, the clock the chip held (GRBM_GUI_ACTIVE is summed over
the 8 XCDs: / 8 / duration; reads high on dispatches shorter than ~0.3 ms, MI355X_MICROARCH.md 'DVFS give-back') and the
matrix-pipe utilisation = MFMA-busy cycles / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    return name.replace("void ", "").split("(")[0][:90]


def main():
    tag, d = sys.argv[1:3]
    note_cmd = sys.argv[3] if len(sys.argv) > 3 else ("bench.py --lean --no-prof --serial-streams --steps 1 --warmup 1 "
                                                       "(config c2)")
    cnt = collections.defaultdict(lambda: collections.defaultdict(float))
    nl = collections.defaultdict(set)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            cnt[k][r["Counter_Name"]] += float(r["Counter_Value"])
            nl[k].add(r["Dispatch_Id"])
    dur = collections.defaultdict(float)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[short(r["Kernel_Name"])] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    out = {}
    for k, c in sorted(cnt.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)):
        n = len(nl[k])
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        if busy <= 0 or gui <= 0 or n == 0:
            continue
        t_ns = dur.get(k, 0.0)
        out[k] = {"launches": n, "avg_us": t_ns / n / 1e3 if t_ns else None,
                  "mfma_busy_cycles_per_launch": busy / n,
                  "clock_ghz": (gui / 8.0) / t_ns if t_ns else None,
                  "matrix_pipe_busy_frac": busy / (gui / 8.0 * 1024.0),
                  "sq_wait_any_frac_of_wave_cycles": c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None}
    path = os.path.join(ROOT, "profiles", f"{tag}_mfma_busy_summary.json")
    with open(path, "w") as fh:
        json.dump({"note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY "
                           "GRBM_GUI_ACTIVE --kernel-trace of " + note_cmd + "; profiled passes hold a lower clock than "
                           "un-profiled ones",
                   "kernels": out}, fh, indent=1)
    print("wrote", path)
    for k, v in list(out.items())[:10]:
        print(f"{k[:60]:60s} n={v['launches']:4d} busy {v['matrix_pipe_busy_frac']:.2f} clock {v['clock_ghz'] or 0:.2f} GHz")


if __name__ == "__main__":
    main()
