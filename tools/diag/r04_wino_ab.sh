cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_conv.py -x -q -k "winograd or persistent or conv_fwd_bwd" -s 2>&1 | grep -E "winograd tile|passed|failed|Error|error" | tail -120 > gpurun_out/r04_wino_test.log
for t in 4 6; do
  DIGA_CONV_WINOGRAD_TILE=$t python tools/bench_conv.py --only conv2 > gpurun_out/r04_benchconv_tile$t.log 2>&1
  DIGA_CONV_WINOGRAD_TILE=$t python tools/bench_conv.py --only aspp >> gpurun_out/r04_benchconv_tile$t.log 2>&1
done
for t in 4 6 4 6; do
  DIGA_CONV_WINOGRAD_TILE=$t python bench.py --lean --steps 6 --warmup 2 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('tile cap $t', d['ms_per_step'])" >> gpurun_out/r04_lean_tiles.log
done
tail -4 gpurun_out/r04_wino_test.log; cat gpurun_out/r04_lean_tiles.log
