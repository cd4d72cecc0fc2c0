# round-4 profile of the SegFormer (c5) step: rocprofv3 kernel stats of the eager one-stream step, 3 steps after 1 warm-up
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_c5_stats -- python3 $R/bench.py --lean --no-prof --warmup 1 --config c5 --serial-streams --steps 3 > $R/gpurun_out/p_c5.log 2>&1
cd $R
mkdir -p gpurun_out/profiles_r04
find gpurun_out/p_c5_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/profiles_r04/r04_c5_segformer_serial_kernel_stats.csv
rm -rf gpurun_out/p_c5_stats
tail -1 gpurun_out/p_c5.log | cut -c1-300
head -25 gpurun_out/profiles_r04/r04_c5_segformer_serial_kernel_stats.csv | cut -c1-150
