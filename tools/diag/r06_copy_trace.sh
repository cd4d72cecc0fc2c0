#!/bin/bash
# Which HIP API calls are behind the ~365 __amd_rocclr_copyBuffer kernels rocprofv3 lists per C2 step?  HIP runtime trace + kernel trace + memory-copy trace
# (no counters), then count the API calls / copies by name and size.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --hip-runtime-trace --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/p_copy -- python3 $R/bench.py --lean --no-prof --warmup 1 --steps 1 --serial-streams > $R/gpurun_out/p_copy.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
api = collections.Counter()
for f in glob.glob("gpurun_out/p_copy/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        api[r["Function"]] += 1
print("HIP API calls (2 steps):")
for k, v in api.most_common(25):
    print(f"  {k:45s} {v}")
mc = collections.Counter(); sizes = collections.Counter()
for f in glob.glob("gpurun_out/p_copy/**/*memory_copy_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    print("memory copy trace columns:", list(rows[0].keys()) if rows else None, len(rows))
    for r in rows:
        mc[r.get("Direction", "?")] += 1
        sizes[(r.get("Direction", "?"), r.get("Bytes", r.get("Size", "?")))] += 1
print("copies by direction:", dict(mc))
for k, v in sizes.most_common(20):
    print("  ", k, v)
PY
rm -rf gpurun_out/p_copy
