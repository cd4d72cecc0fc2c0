#!/bin/bash
# Per-kernel averages (rocprofv3 --kernel-trace --stats) of the Winograd layer l3.conv2 (16 images) for library variants.
#   gpurun -- 'bash tools/diag/r05_wino_kernels.sh "" _occ4 ...'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_wino_kernels.txt
: > $OUT
for v in "$@"; do
  rm -rf $R/gpurun_out/ks_w
  DIGA_LIB=$R/diga_amd/libdiga_hip$v.so rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_w -- python3 $R/tools/bench_conv.py --only l3.conv2 --reps 10 > /dev/null 2>&1
  echo "== variant '$v'" >> $OUT
  f=$(find $R/gpurun_out/ks_w -name "*kernel_stats.csv" | head -1)
  python3 - $f >> $OUT <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "wino" in n or "gemm" in n or "wgrad_dma" in n:
        print(f"{n.split('(')[0].replace('void ', '')[:64]:66s} calls {r['Calls']:>4s} avg {float(r['AverageNs']) / 1e3:8.1f} us  min {float(r['MinNs']) / 1e3:8.1f}")
PY
done
rm -rf $R/gpurun_out/ks_w
cat $OUT
