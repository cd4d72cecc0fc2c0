#!/bin/bash
# Locate a regression against the round-4 tree: serialised-step kernel stats of both trees on one box, per-kernel totals side by side.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for tree in r04 r05off; do
  d=$R; envs="DIGA_LIB=$R/diga_amd/libdiga_hip${VARIANT}.so"
  if [ $tree = r04 ]; then d=$R/.r04_tree; envs="X=1"; fi
  rm -rf $R/gpurun_out/pr_$tree
  (cd /tmp && export $envs && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pr_$tree -- python3 $d/bench.py --lean --no-prof --warmup 1 --serial-streams --steps 3 > $R/gpurun_out/pr_$tree.log 2>&1)
  find $R/gpurun_out/pr_$tree -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/regress_${tree}_kernel_stats.csv
  rm -rf $R/gpurun_out/pr_$tree
done
python3 - $R/gpurun_out/regress_r04_kernel_stats.csv $R/gpurun_out/regress_r05off_kernel_stats.csv <<'PY'
import csv, sys
def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        d[r["Name"].split("(")[0].replace("void ", "")[:70]] = (int(r["Calls"]), int(r["TotalDurationNs"]) / 4e6)
    return d
a, b = load(sys.argv[1]), load(sys.argv[2])
rows = []
for k in set(a) | set(b):
    ca, ta = a.get(k, (0, 0.0)); cb, tb = b.get(k, (0, 0.0))
    rows.append((tb - ta, k, ca, ta, cb, tb))
print("total ms/step r04 %.1f  r05off %.1f" % (sum(v[1] for v in a.values()), sum(v[1] for v in b.values())))
for dlt, k, ca, ta, cb, tb in sorted(rows, key=lambda r: -abs(r[0]))[:25]:
    print(f"{k:72s} r04 {ca:5d} {ta:8.2f}   r05off {cb:5d} {tb:8.2f}   delta {dlt:+7.2f}")
PY
