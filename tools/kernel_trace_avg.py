#!/usr/bin/env python3
"""Per-kernel launch count / average / minimum duration of a rocprofv3 --kernel-trace CSV:  python tools/kernel_trace_avg.py <dir>"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    agg[r["Kernel_Name"].split("(")[0].replace("void ", "")[:50]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if "diga" in k: print(f"{k:52s} n={len(v):3d} avg {sum(v)/len(v)/1e3:9.1f} us  min {min(v)/1e3:9.1f}")
