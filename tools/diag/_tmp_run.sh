cd "$GRAFT_REPO_ROOT"
timeout 900 python bench.py --config c5 --steps 10 --warmup 3 --lean 2>&1 | tail -1 | cut -c1-200
timeout 600 python tools/diag/segformer_head_probe.py 2>&1 | grep -v "^\-\-\-" | cut -c1-95,200-330 | sed -n 3,30p
