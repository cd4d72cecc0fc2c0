#!/bin/bash
# c5 from its HIP graph: library variants (name=ENV=VAL ...), interleaved on one box, three rounds
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_ab_c5lib.txt
: > $OUT
for round in 1 2 3; do
  for spec in "$@"; do
    name=${spec%%=*}; envs=${spec#*=}
    if [ "$name" = "$spec" ]; then envs=""; fi
    ms=$(env $envs python $R/bench.py --lean --graph --config c5 --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$round $name $ms" | tee -a $OUT
  done
done
