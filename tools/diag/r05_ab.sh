#!/bin/bash
# Same-box A/B of library variants / host switches on the C2 step (bench.py --lean): two interleaved rounds.
#   gpurun -- 'bash tools/diag/r05_ab.sh "name=ENV=VAL ..." ...'   (name "default" = no override)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_ab.txt
: > $OUT
for round in 1 2 3; do
  for spec in "$@"; do
    name=${spec%%=*}; envs=${spec#*=}
    if [ "$name" = "$spec" ]; then envs=""; fi
    ms=$(env $envs python $R/bench.py --lean --steps 12 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$round $name $ms" | tee -a $OUT
  done
done
