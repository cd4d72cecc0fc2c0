#!/bin/bash
# For every __amd_rocclr_copyBuffer kernel of a serialised C2 step: the kernels dispatched right before / after it (same process, by start time).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/p_cn -- python3 $R/bench.py --lean --no-prof --warmup 1 --steps 1 --serial-streams > $R/gpurun_out/p_cn.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
rows = []
for f in glob.glob("gpurun_out/p_cn/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void ", "").replace("diga::", "")[:48] for r in rows]
n = len(rows)
half = rows[n // 2:]          # the second (timed) step, roughly
off = n // 2
prev = collections.Counter(); nxt = collections.Counter(); sizes = collections.Counter()
for i in range(off, n):
    if "copyBuffer" in names[i]:
        prev[names[i - 1]] += 1
        if i + 1 < n: nxt[names[i + 1]] += 1
        sizes[(rows[i].get("Grid_Size", rows[i].get("Grid_Size_X", "?")), rows[i].get("Workgroup_Size", rows[i].get("Workgroup_Size_X", "?")))] += 1
print("copyBuffer kernels in the second half of the trace:", sum(prev.values()))
print("dispatched right BEFORE them:", prev.most_common(12))
print("dispatched right AFTER them:", nxt.most_common(12))
print("grid / workgroup sizes:", sizes.most_common(6))
PY
rm -rf gpurun_out/p_cn
