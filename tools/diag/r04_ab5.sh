cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in base tl2 tl1; do
  if [ $v = base ]; then unset DIGA_LIB; else export DIGA_LIB=$R/diga_amd/libdiga_probe_$v.so; fi
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_ab5 -- python3 $R/bench.py --lean --no-prof --warmup 1 --serial-streams --precision f32 --steps 3 > $R/gpurun_out/p_ab5.log 2>&1
  cd $R
  f=$(find gpurun_out/p_ab5 -name "*kernel_stats.csv" | head -1)
  echo "== $v" >> gpurun_out/r04_ab5.log
  grep -E "winoM_output_epi" $f | awk -F'",' '{print substr($1,1,60), $2}' | cut -c1-120 >> gpurun_out/r04_ab5.log
  rm -rf gpurun_out/p_ab5
  for i in 1 2; do python bench.py --lean --steps 8 --warmup 3 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('step $v', d['ms_per_step'])" >> gpurun_out/r04_ab5.log; done
done
cat gpurun_out/r04_ab5.log
