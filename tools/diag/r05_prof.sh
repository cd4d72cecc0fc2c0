#!/bin/bash
# Round-5 profile of the C2 step: (1) rocprofv3 --kernel-trace --stats of the serialised step -> per-kernel averages,
# (2) kernel trace of the default two-stream step -> tools/diag/overlap_timeline.py (what runs with no matrix-core kernel in flight).
#   gpurun -- 'bash tools/diag/r05_prof.sh [tag]'
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_ser -- python3 $R/bench.py --lean --no-prof --warmup 1 --serial-streams --precision f32 --steps 3 > $R/gpurun_out/p_ser.log 2>&1
cd $R
find gpurun_out/p_ser -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${TAG}_f32_serial_kernel_stats.csv
rm -rf gpurun_out/p_ser
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/p_tl -- python3 $R/bench.py --lean --no-prof --steps 3 --warmup 2 > $R/gpurun_out/p_tl.log 2>&1
cd $R
python tools/diag/overlap_timeline.py gpurun_out/p_tl > gpurun_out/${TAG}_overlap_timeline.txt 2>&1
python tools/kernel_trace_avg.py gpurun_out/p_tl > gpurun_out/${TAG}_overlapped_kernel_avg.txt 2>&1
rm -rf gpurun_out/p_tl
tail -3 gpurun_out/p_ser.log | cut -c1-200
cat gpurun_out/${TAG}_overlap_timeline.txt
