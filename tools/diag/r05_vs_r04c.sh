#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_vs_r04c.txt
: > $OUT
run() { (cd $1 && env $2 python bench.py --lean --steps 12 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"); }
for round in 1 2 3; do
  echo "$round r05_in4 $(run $R "X=1")" | tee -a $OUT
  echo "$round r04 $(run $R/.r04_tree "X=1")" | tee -a $OUT
  echo "$round r05_in2 $(run $R "DIGA_LIB=$R/diga_amd/libdiga_hip_in2.so")" | tee -a $OUT
  echo "$round r05_in2noxcd $(run $R "DIGA_LIB=$R/diga_amd/libdiga_hip_in2noxcd.so")" | tee -a $OUT
done
