cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize_properties.py tests/test_selftrain.py tests/test_gpu_dropin_selftrain.py -m gpu -q 2>&1 | tail -4 > gpurun_out/r04_bw_tests.log
python bench.py --steps 2 --warmup 1 --no-other-precision --no-other-configs --no-cpu-baseline --no-miou > gpurun_out/r04_bw_bench.log 2>&1
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/bench_detail.json'))
for k,v in d['bandwidth_kernels'].items(): print(k, round(v['frac'],3), round(v['avg_launch_ms']*1e3,1),'us')
for k in ('classmix_hist','classmix_paste','upsample_loss'):
    v=d['roofline_other_kernels'][k]; print(k, round(v['frac'],3), round(v['avg_launch_ms']*1e3,1),'us')
PY
cat gpurun_out/r04_bw_tests.log
