cd "$GRAFT_REPO_ROOT"
timeout 600 python tools/diag/segformer_graph_probe.py 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_segformer_head.py -x -q -k "graph_captured or segformer" 2>&1 | tail -3
