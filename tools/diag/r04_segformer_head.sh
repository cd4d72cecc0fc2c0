#!/bin/bash
# round-4 check of the SegFormer head: its tests, the old MiT / graph tests on the ASPP wiring, and the head's stand-alone timing
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_segformer_head.py -x -q 2>&1 | tail -25
timeout 900 python -m pytest tests/test_gpu_mit.py tests/test_gpu_model.py -x -q -k "segformer or graph_captured" 2>&1 | tail -8
timeout 600 python tools/diag/segformer_head_probe.py 2>&1 | tail -45
