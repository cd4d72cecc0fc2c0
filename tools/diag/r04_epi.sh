cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 1200 python -m pytest tests/test_gpu_bn_box.py tests/test_gpu_bf16x3_parity.py tests/test_gpu_conv.py -m gpu -q -x 2>&1 | tail -3 > gpurun_out/r04_epi_tests.log
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_epi -- python3 $R/bench.py --lean --no-prof --warmup 1 --serial-streams --precision f32 --steps 3 > $R/gpurun_out/p_epi.log 2>&1
cd $R
f=$(find gpurun_out/p_epi -name "*kernel_stats.csv" | head -1)
grep -E "winoM_output_epi|winoM_output_kernel<6|winoM_input_kernel<6" $f | awk -F'",' '{print substr($1,1,70), $2}' | cut -c1-140 > gpurun_out/r04_epi_stats.log
rm -rf gpurun_out/p_epi
for i in 1 2; do python bench.py --lean --steps 8 --warmup 3 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('step', d['ms_per_step'])" >> gpurun_out/r04_epi_stats.log; done
cat gpurun_out/r04_epi_tests.log gpurun_out/r04_epi_stats.log
