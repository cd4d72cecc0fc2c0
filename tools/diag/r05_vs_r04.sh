#!/bin/bash
# Same-box A/B of this tree against the round-4 tree (.r04_tree = `git archive 2992702`, built in place): three interleaved rounds of
# bench.py --lean --steps 12 (C2) and one of the c4 leg.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_vs_r04.txt
: > $OUT
for round in 1 2 3; do
  for tree in r05 r04; do
    d=$R; [ $tree = r04 ] && d=$R/.r04_tree
    ms=$(cd $d && python bench.py --lean --steps 12 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "c2 $round $tree $ms" | tee -a $OUT
  done
done
for tree in r05 r04 r05 r04; do
  d=$R; [ $tree = r04 ] && d=$R/.r04_tree
  ms=$(cd $d && python bench.py --lean --config c4 --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "c4 $tree $ms" | tee -a $OUT
done
