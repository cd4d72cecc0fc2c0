#!/bin/bash
# Weight-gradient input hold (config.wgrad_hold) against record_stream: bit-identity of the stream forms, step time and allocator
# totals for C2 / c4 / split bf16, the per-stream pools, then the stream-structure tests.
R=$GRAFT_REPO_ROOT
cd $R
python tools/diag/r06_hold_race.py 2>&1 | grep hold | tee gpurun_out/r06_hold_race.txt
bash tools/diag/r06_ab.sh 2 "" "hold0=DIGA_WGRAD_HOLD=0" "hold2=DIGA_WGRAD_HOLD=2" "hold4=DIGA_WGRAD_HOLD=4" "hold8=DIGA_WGRAD_HOLD=8" "hold16=DIGA_WGRAD_HOLD=16"
tail -5 gpurun_out/r06_ab_last_stderr.txt
bash tools/diag/r06_ab.sh 1 "--config c4" "hold0=DIGA_WGRAD_HOLD=0" "hold4=DIGA_WGRAD_HOLD=4" "hold8=DIGA_WGRAD_HOLD=8"
bash tools/diag/r06_ab.sh 1 "--precision bf16x3" "hold0=DIGA_WGRAD_HOLD=0" "hold4=DIGA_WGRAD_HOLD=4"
python tools/diag/r06_mem_by_stream.py c2 8 2>&1 | tail -6 | tee gpurun_out/r06_hold_by_stream.txt
python tools/diag/r06_mem_by_stream.py c4 8 2>&1 | tail -7 | tee -a gpurun_out/r06_hold_by_stream.txt
timeout 1200 python -m pytest tests/test_selftrain.py tests/test_gpu_ddp_step.py tests/test_gpu_mit.py tests/test_gpu_trajectory.py -m gpu -x -q 2>&1 | tail -5
