cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in f4 w2; do
  if [ $v = w2 ]; then export DIGA_LIB=$GRAFT_REPO_ROOT/diga_amd/libdiga_probe_w2.so; fi
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_wino4 -o w4 --output-format csv -- python3 tools/bench_conv.py --only l3.conv2 --reps 10 > gpurun_out/r04_prof_wino4_$v.log 2>&1
  find gpurun_out/prof_wino4 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r04_wino4_l3conv2_kernel_stats_$v.csv
  echo == $v; head -8 gpurun_out/r04_wino4_l3conv2_kernel_stats_$v.csv | cut -c1-60,150-260
  rm -rf gpurun_out/prof_wino4
done
