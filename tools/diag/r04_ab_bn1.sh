cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_conv.py tests/test_gpu_fullsize_golden.py -m gpu -q -s 2>&1 | grep -E "FAILED|passed|failed|aspp .* y:|aspp .* dw:" | tail -30 > gpurun_out/r04_gputests2.log
for f in 0 1 0 1; do
  DIGA_FUSE_BN1=$f python bench.py --lean --steps 8 --warmup 3 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('FUSE_BN1=$f', d['ms_per_step'])" >> gpurun_out/r04_ab_bn1.log
done
cat gpurun_out/r04_gputests2.log gpurun_out/r04_ab_bn1.log
