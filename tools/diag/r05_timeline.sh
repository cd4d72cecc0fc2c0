#!/bin/bash
# Overlap timeline of the default two-stream step (what runs with no matrix-core kernel in flight):  bash tools/diag/r05_timeline.sh [tag]
TAG=${1:-tl}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/p_tl
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/p_tl -- python3 $R/bench.py --lean --no-prof --steps 3 --warmup 2 > $R/gpurun_out/p_tl.log 2>&1
cd $R
python tools/diag/overlap_timeline.py gpurun_out/p_tl > gpurun_out/${TAG}_overlap_timeline.txt 2>&1
rm -rf gpurun_out/p_tl
cat gpurun_out/${TAG}_overlap_timeline.txt
