cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 0 1 0 1; do
  cd $R
  DIGA_FUSE_JUNCTION=$v python bench.py --lean --steps 8 --warmup 3 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('junction fusion $v: step', d['ms_per_step'])" >> gpurun_out/r04_ab6.log
done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_ab6 -- python3 $R/bench.py --lean --no-prof --warmup 1 --serial-streams --precision f32 --steps 3 > $R/gpurun_out/p_ab6.log 2>&1
cd $R
f=$(find gpurun_out/p_ab6 -name "*kernel_stats.csv" | head -1)
grep -E "persistent|affine_apply" $f | awk -F'",' '{print substr($1,1,70), $2}' | cut -c1-140 >> gpurun_out/r04_ab6.log
rm -rf gpurun_out/p_ab6
cat gpurun_out/r04_ab6.log
