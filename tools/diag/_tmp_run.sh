cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_conv.py tests/test_gpu_bn_box.py tests/test_gpu_norm.py -x -q 2>&1 | tail -3
timeout 900 python bench.py --steps 10 --warmup 3 --lean 2>&1 | tail -1 | cut -c1-200
