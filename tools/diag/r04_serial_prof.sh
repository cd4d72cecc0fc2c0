cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_f32_stats -- python3 $R/bench.py --lean --no-prof --warmup 1 --serial-streams --precision f32 --steps 3 > $R/gpurun_out/p1.log 2>&1
cd $R
find gpurun_out/p_f32_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r04b_f32_serial_kernel_stats.csv
rm -rf gpurun_out/p_f32_stats
