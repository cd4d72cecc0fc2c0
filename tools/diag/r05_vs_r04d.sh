#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_vs_r04d.txt
: > $OUT
run() { (cd $1 && env $2 python bench.py --lean --steps 12 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"); }
for round in 1 2 3; do
  echo "$round r05 $(run $R "X=1")" | tee -a $OUT
  echo "$round r04 $(run $R/.r04_tree "X=1")" | tee -a $OUT
  echo "$round r05_nofusebn1 $(run $R "DIGA_FUSE_BN1=0")" | tee -a $OUT
  echo "$round r05_noxcd $(run $R "DIGA_LIB=$R/diga_amd/libdiga_hip_noxcd.so")" | tee -a $OUT
  echo "$round r05_nostats $(run $R "DIGA_WINOGRAD_STATS=0")" | tee -a $OUT
done
