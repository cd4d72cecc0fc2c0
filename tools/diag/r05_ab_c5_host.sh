#!/bin/bash
# c5 from its HIP graph: this tree against a second checkout of the Python side (same library), interleaved on one box
#   gpurun -- 'bash tools/diag/r05_ab_c5_host.sh <other_tree_relative_to_repo_root>'
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_ab_c5_host.txt
: > $OUT
run() { (cd $1 && python bench.py --lean --graph --config c5 --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"); }
for round in 1 2 3; do
  echo "$round new $(run $R)" | tee -a $OUT
  echo "$round old $(run $R/$1)" | tee -a $OUT
done
