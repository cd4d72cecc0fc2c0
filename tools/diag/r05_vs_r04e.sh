#!/bin/bash
# This tree against the round-4 tree on ONE box, interleaved.  The round-4 tree is a scratch copy next to this one (git-ignored, removed
# at the end of the round):  mkdir .r04_tree && git archive 2992702 | tar -x -C .r04_tree && (cd .r04_tree && python -m diga_amd.build)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_vs_r04e.txt
: > $OUT
run() { (cd $1 && env $2 python bench.py --lean $3 --steps 12 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"); }
for round in 1 2 3; do
  echo "c2 $round r05 $(run $R "X=1")" | tee -a $OUT
  echo "c2 $round r04 $(run $R/.r04_tree "X=1")" | tee -a $OUT
done
for round in 1 2; do
  echo "c4 $round r05 $(run $R "X=1" "--config c4")" | tee -a $OUT
  echo "c4 $round r04 $(run $R/.r04_tree "X=1" "--config c4")" | tee -a $OUT
done
