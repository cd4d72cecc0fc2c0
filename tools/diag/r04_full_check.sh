cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04_gputests.log
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_f32_stats -- python3 $R/bench.py --lean --no-prof --warmup 1 --serial-streams --precision f32 --steps 3 > $R/gpurun_out/p1.log 2>&1
cd $R
find gpurun_out/p_f32_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r04a_f32_serial_kernel_stats.csv
rm -rf gpurun_out/p_f32_stats
python bench.py --lean --steps 10 --warmup 3 > gpurun_out/r04a_lean.log 2>&1
tail -5 gpurun_out/r04_gputests.log; head -25 gpurun_out/r04a_f32_serial_kernel_stats.csv | cut -c1-100,180-260; tail -1 gpurun_out/r04a_lean.log | cut -c1-300
