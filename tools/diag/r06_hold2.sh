#!/bin/bash
# After the hold: the stream on/off table again, the c4 forms, the largest batches.
R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests/test_gpu_memory.py -m gpu -x -q -s 2>&1 | tail -8
bash tools/diag/r06_ab.sh 1 "" "default" "no_teacher_stream=DIGA_TEACHER_STREAM=0" "no_wgrad_stream=DIGA_WGRAD_STREAM=0" "serial=DIGA_TEACHER_STREAM=0 DIGA_WGRAD_STREAM=0" "b16=+--batch +16" "b18=+--batch +18" "b20=+--batch +20"
mv gpurun_out/r06_ab__.txt gpurun_out/r06_hold_memory_c2.txt
bash tools/diag/r06_ab.sh 1 "--config c4" "overlap2" "overlap1=DIGA_C4_OVERLAP=1" "overlap0=DIGA_C4_OVERLAP=0" "serial=DIGA_TEACHER_STREAM=0 DIGA_WGRAD_STREAM=0 DIGA_C4_OVERLAP=0" "b12=+--batch +12" "b14=+--batch +14"
mv gpurun_out/r06_ab___config_c4_.txt gpurun_out/r06_hold_memory_c4.txt
