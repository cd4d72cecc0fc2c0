#!/usr/bin/env python3
"""Per-kernel mean of every counter in the *counter_collection.csv files of a rocprofv3 --pmc run:  python tools/pmc_sum.py <dir>"""
import csv, glob, sys, collections
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); 
        cnt[(k, row["Counter_Name"])] += 1
for k, v in agg.items():
    if "conv" not in k: continue
    print(k, {c: round(x / cnt[(k, c)]) for c, x in v.items()})
