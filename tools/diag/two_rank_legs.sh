# Diagnostic: the two precision legs of a 2-rank (gloo, both ranks on one GPU) bench run under different switches.
run() { DIGA_DDP_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 --batch 2 --size 384 384 --no-other-configs --no-bandwidth-kernels 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['second_precision']['ms_per_step'])"; }
echo "default"; run
