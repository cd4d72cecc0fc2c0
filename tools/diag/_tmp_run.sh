cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_gpu_mit.py -x -q -k dwconv 2>&1 | tail -4
timeout 900 python bench.py --config c5 --steps 10 --warmup 3 --lean 2>&1 | tail -1 | cut -c1-200
