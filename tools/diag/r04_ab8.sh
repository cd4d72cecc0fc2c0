cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -q -x -k "bit_identical or winograd or stat" 2>&1 | tail -3 > gpurun_out/r04_ab8.log
for i in 1 2; do python bench.py --lean --steps 8 --warmup 3 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('step', d['ms_per_step'])" >> gpurun_out/r04_ab8.log; done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_ab8 -- python3 $R/bench.py --lean --no-prof --warmup 1 --serial-streams --precision f32 --steps 3 > $R/gpurun_out/p_ab8.log 2>&1
cd $R
f=$(find gpurun_out/p_ab8 -name "*kernel_stats.csv" | head -1)
grep -E "persistent" $f | awk -F'",' '{print substr($1,1,70), $2}' | cut -c1-140 >> gpurun_out/r04_ab8.log
rm -rf gpurun_out/p_ab8
python tools/bench_conv.py --only l3.conv >> gpurun_out/r04_ab8.log 2>&1
cat gpurun_out/r04_ab8.log
