#!/bin/bash
# This tree against the round-5 tree on ONE box, interleaved (boxes of the pool differ by +-1.5 %).  The round-5 tree is a scratch copy
# next to this one (git-ignored):  mkdir .r05_tree && git archive ad449cd | tar -x -C .r05_tree && (cd .r05_tree && python -m diga_amd.build)
# usage: bash tools/diag/r06_vs_r05.sh [rounds_c2] [rounds_c4] [rounds_c5]
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_vs_r05.txt
: > $OUT
N2=${1:-3}; N4=${2:-0}; N5=${3:-0}
run() { (cd $1 && python bench.py --lean $2 --steps 12 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"); }
for round in $(seq 1 $N2); do
  echo "c2 $round r06 $(run $R "")" | tee -a $OUT
  echo "c2 $round r05 $(run $R/.r05_tree "")" | tee -a $OUT
done
for round in $(seq 1 $N4); do
  echo "c4 $round r06 $(run $R "--config c4")" | tee -a $OUT
  echo "c4 $round r05 $(run $R/.r05_tree "--config c4")" | tee -a $OUT
done
for round in $(seq 1 $N5); do
  echo "c5 $round r06 $(run $R "--config c5")" | tee -a $OUT
  echo "c5 $round r05 $(run $R/.r05_tree "--config c5")" | tee -a $OUT
done
