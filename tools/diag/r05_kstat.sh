#!/bin/bash
# Serialised-step kernel totals (ms per step) of library variants of THIS tree, side by side:  bash tools/diag/r05_kstat.sh "" _frag1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  rm -rf $R/gpurun_out/ks_$v
  DIGA_LIB=$R/diga_amd/libdiga_hip$v.so rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_$v -- python3 $R/bench.py --lean --no-prof --warmup 1 --serial-streams --steps 3 > /dev/null 2>&1
  find $R/gpurun_out/ks_$v -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/kstat$v.csv
  rm -rf $R/gpurun_out/ks_$v
done
python3 - $R/gpurun_out "$@" <<'PY'
import csv, sys
d, vs = sys.argv[1], sys.argv[2:]
tabs = []
for v in vs:
    t = {}
    for r in csv.DictReader(open(f"{d}/kstat{v}.csv")):
        t[r["Name"].split("(")[0].replace("void ", "")[:60]] = int(r["TotalDurationNs"]) / 4e6
    tabs.append(t)
print("total ms/step: " + "  ".join(f"'{v}' {sum(t.values()):.1f}" for v, t in zip(vs, tabs)))
keys = sorted(set().union(*tabs), key=lambda k: -max(t.get(k, 0) for t in tabs))[:14]
for k in keys:
    print(f"{k:62s} " + "  ".join(f"{t.get(k, 0):8.2f}" for t in tabs))
PY
