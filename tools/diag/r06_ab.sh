#!/bin/bash
# Same-box A/B of host switches / bench flags on one configuration (bench.py --lean): interleaved rounds.
#   gpurun -- 'bash tools/diag/r06_ab.sh <rounds> "<bench flags>" "name=ENV=VAL ..." "name2=+--flag" ...'
#   a spec is name=<space-separated ENV=VAL and/or +--bench-flag items>; name alone = defaults
R=$GRAFT_REPO_ROOT
N=$1; shift
BASE=$1; shift
OUT=$R/gpurun_out/r06_ab_$(echo "$BASE" | tr -c 'a-z0-9' '_').txt
: > $OUT
for round in $(seq 1 $N); do
  for spec in "$@"; do
    name=${spec%%=*}; rest=${spec#*=}
    if [ "$name" = "$spec" ]; then rest=""; fi
    envs=""; flags=""
    for item in $rest; do
      case "$item" in +*) flags="$flags ${item#+}";; *) envs="$envs $item";; esac
    done
    line=$(env $envs python $R/bench.py --lean $BASE $flags --steps 12 --warmup 3 2>$R/gpurun_out/r06_ab_last_stderr.txt | tail -1)
    ms=$(echo "$line" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'alloc', round(d.get('peak_mem_gb',{}).get('allocated',0),1), 'resv', round(d.get('peak_mem_gb',{}).get('reserved',0),1))" 2>/dev/null)
    echo "$round $name $ms" | tee -a $OUT
  done
done
