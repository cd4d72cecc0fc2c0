cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_bn_box.py tests/test_gpu_model.py tests/test_gpu_fullsize_golden.py tests/test_gpu_dropin_step.py -m gpu -q -x 2>&1 | tail -3 > gpurun_out/r04_ab7.log
for v in 0 1 0 1; do
  DIGA_FUSE_JUNCTION=$v python bench.py --lean --steps 8 --warmup 3 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('junction fusion $v: step', d['ms_per_step'])" >> gpurun_out/r04_ab7.log
done
cat gpurun_out/r04_ab7.log
