#!/bin/bash
# Refresh of the two-stream kernel stats + overlap timeline after the weight-gradient hold (host-side change only: the serialised
# profiles of tools/profile_round6.sh do not use the side stream and stand).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=r06
B="python3 $R/bench.py --lean --no-prof --warmup 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_f32_ov -- $B --precision f32 --steps 3 > $R/gpurun_out/p5.log 2>&1
cd $R
mkdir -p gpurun_out/profiles_${T}
python3 tools/diag/overlap_timeline.py gpurun_out/p_f32_ov > gpurun_out/profiles_${T}/${T}_f32_overlap_timeline.txt 2>&1
find gpurun_out/p_f32_ov -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/profiles_${T}/${T}_f32_overlapped_kernel_stats.csv
rm -rf gpurun_out/p_f32_ov
cat gpurun_out/profiles_${T}/${T}_f32_overlap_timeline.txt; tail -2 gpurun_out/p5.log
