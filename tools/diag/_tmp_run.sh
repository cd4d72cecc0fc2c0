cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_mit.py -x -q 2>&1 | tail -3
timeout 600 python tools/bench_mit_ops.py --cold 2>&1 | grep -i "stage\|dwconv"
