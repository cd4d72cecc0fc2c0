cd "$GRAFT_REPO_ROOT"; timeout 600 python -m pytest tests/test_gpu_segformer_head.py -q 2>&1 | tail -3
