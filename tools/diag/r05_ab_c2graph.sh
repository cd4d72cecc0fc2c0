#!/bin/bash
# C2 (main configuration) eager two-stream step vs the same step replayed from a HIP graph with forked streams, interleaved on one box
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_ab_c2graph.txt
: > $OUT
for round in 1 2; do
  for name in eager graph; do
    flag=""; [ "$name" = graph ] && flag="--graph"
    ms=$(python $R/bench.py --lean $flag --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$round $name $ms" | tee -a $OUT
  done
done
