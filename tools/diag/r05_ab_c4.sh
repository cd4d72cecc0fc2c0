#!/bin/bash
# Same-box A/B on the self-training step (config c4): three interleaved rounds.  Specs as r05_ab.sh.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_ab_c4.txt
: > $OUT
for round in 1 2 3; do
  for spec in "$@"; do
    name=${spec%%=*}; envs=${spec#*=}
    if [ "$name" = "$spec" ]; then envs=""; fi
    ms=$(env $envs python $R/bench.py --lean --config c4 --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$round $name $ms" | tee -a $OUT
  done
done
