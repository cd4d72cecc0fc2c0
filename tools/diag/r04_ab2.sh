cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_mit.py tests/test_gpu_bn_box.py tests/test_gpu_bf16x3_parity.py tests/test_gpu_ddp_step.py tests/test_gpu_fullsize_golden.py -m gpu -q -s 2>&1 | grep -E "FAILED|passed|failed|mit_b|RESNET101 math=0|TINY math=0|logits max err" | tail -30 > gpurun_out/r04_gputests4.log
for v in f2 e4 f2 e4; do
  if [ $v = e4 ]; then export DIGA_LIB=$GRAFT_REPO_ROOT/diga_amd/libdiga_probe_e4.so; else unset DIGA_LIB; fi
  python bench.py --lean --steps 8 --warmup 3 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('EPI=$v', d['ms_per_step'])" >> gpurun_out/r04_ab2.log
done
unset DIGA_LIB
DIGA_STEP_GRAPH=1 python bench.py --lean --steps 8 --warmup 3 2>gpurun_out/r04_graph_c2.err | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('STEP_GRAPH=1', d['ms_per_step'])" >> gpurun_out/r04_ab2.log
cat gpurun_out/r04_gputests4.log gpurun_out/r04_ab2.log; tail -3 gpurun_out/r04_graph_c2.err
