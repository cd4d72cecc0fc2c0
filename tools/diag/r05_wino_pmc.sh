#!/bin/bash
# HBM-side bytes of the Winograd transform kernels on the l3.conv2 shape (16 images), per library variant: separate --pmc passes for
# FETCH_SIZE and WRITE_SIZE (MI355X_MICROARCH.md: FETCH_SIZE x 2 on gfx950 for wide streaming reads).
#   gpurun -- 'bash tools/diag/r05_wino_pmc.sh "" _rowmajor _noxcd'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_wino_pmc.txt
: > $OUT
for v in "$@"; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/pmc_w
    DIGA_LIB=$R/diga_amd/libdiga_hip$v.so rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/gpurun_out/pmc_w -- python3 $R/tools/bench_conv.py --only l3.conv2 --reps 3 > /dev/null 2>&1
    echo "== variant '$v' $ctr" >> $OUT
    python3 - $R/gpurun_out/pmc_w >> $OUT <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")[:70]
        agg[(k, row["Counter_Name"])] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
for (k, c), v in sorted(agg.items()):
    if "wino" in k or "gemm" in k or "wgrad" in k:
        print(f"{k:72s} {c} mean {v / cnt[(k, c)]:12.0f} (x{cnt[(k, c)]})")
PY
  done
done
rm -rf $R/gpurun_out/pmc_w
cat $OUT
