set -x
python -m pytest tests/test_gpu_conv.py -x -q -k "winograd" -s 2>&1 | grep -E "winograd tile|passed|failed|Error|error" | tail -80 > gpurun_out/r04_wino_test.log
for t in 2 4; do
  DIGA_CONV_WINOGRAD_TILE=$t python tools/bench_conv.py --only conv2 > gpurun_out/r04_benchconv_tile$t.log 2>&1
  DIGA_CONV_WINOGRAD_TILE=$t python tools/bench_conv.py --only aspp >> gpurun_out/r04_benchconv_tile$t.log 2>&1
done
for t in 2 4; do
  DIGA_CONV_WINOGRAD_TILE=$t python bench.py --lean --steps 6 --warmup 2 > gpurun_out/r04_lean_tile$t.log 2>&1
done
tail -3 gpurun_out/r04_wino_test.log
