#!/bin/bash
# Is anything in the round-5 tree slower than round 4 with the round-5 switches off?  Same box, interleaved.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_vs_r04b.txt
: > $OUT
run() { (cd $1 && env $2 python bench.py --lean --steps 12 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"); }
for round in 1 2 3; do
  echo "$round r05 $(run $R "X=1")" | tee -a $OUT
  echo "$round r04 $(run $R/.r04_tree "X=1")" | tee -a $OUT
  echo "$round r05_switches_off $(run $R "DIGA_LIB=$R/diga_amd/libdiga_hip_r04like.so DIGA_FUSE_BN1=0 DIGA_WINOGRAD_STATS=0")" | tee -a $OUT
  echo "$round r05_lib_only_off $(run $R "DIGA_LIB=$R/diga_amd/libdiga_hip_r04like.so")" | tee -a $OUT
done
