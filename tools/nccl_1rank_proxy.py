#!/usr/bin/env python3
"""One-GPU proxy for the N>1 RCCL path:  python tools/nccl_1rank_proxy.py <B> <crop> [f32|bf16x3]   (e.g. 8 768 f32)."""
# Proxy for the N>1 RCCL path on one GPU: a 1-rank "nccl" process group with the gradient reducer forced on, so that every
# step runs hook-driven RCCL all-reduces (RCCL stream + events) next to the teacher / weight-gradient side streams.
import os, sys, time, random
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
from diga_amd import ddp, synthetic, _lib
ddp.world_size = lambda: 2            # make the reducer (and its hooks) believe there are peers; all-reduce runs on RCCL
from diga_amd.model import seg_model_noaux as sm
from diga_amd.model.model_noaux import SegModel
from diga_amd.train_step import DigaTrainer
_lib.set_conv_math(0 if (len(sys.argv) > 3 and sys.argv[3] == 'f32') else 1)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
student, teacher = SegModel(arch=sm.RESNET101).to(dev), SegModel(arch=sm.RESNET101).to(dev)
teacher.train()
tr = DigaTrainer(student, teacher, rng=random.Random(1))
B, H, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[2])
batch = synthetic.warmup_batch(1234, B, H, W, block=32, device=dev)
for it in range(2):
    out = tr.warmup_step(it, *batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for it in range(2, 6):
    out = tr.warmup_step(it, *batch)
torch.cuda.synchronize()
print(f"nccl-1rank proxy: {1e3 * (time.perf_counter() - t0) / 4:.1f} ms/step  hooks={len(tr.reducer._hooks)} buckets={len(tr.reducer.buckets)} loss={float(out['total']):.4f}", flush=True)
dist.destroy_process_group()

# ---- correctness of the hook-driven RCCL path (buckets packed and launched from the weight-gradient side stream): with
# one real rank the all-reduce is an identity, so 3 steps through it must leave the student bit-identical to 3 steps
# without a reducer whose optimizer carries the same grad_scale = 1/2.
def run(with_reducer):
    dist_ws = (lambda: 2) if with_reducer else (lambda: 1)
    ddp.world_size = dist_ws
    torch.manual_seed(1)
    s, t = SegModel(arch=sm.RESNET101).to(dev), SegModel(arch=sm.RESNET101).to(dev)
    t.train()
    trn = DigaTrainer(s, t, rng=random.Random(2))
    trn.opt.grad_scale = 0.5
    assert (len(trn.reducer._hooks) > 0) == with_reducer
    b = synthetic.warmup_batch(77, 2, 256, 256, block=32, device=dev)
    for it in range(3):
        trn.warmup_step(it, *b)
    torch.cuda.synchronize()
    return {k: v.detach().clone() for k, v in s.named_parameters()}


dist.init_process_group("nccl", rank=0, world_size=1)
a_, b_ = run(True), run(False)
bad = [k for k in a_ if not torch.equal(a_[k], b_[k])]
print("nccl-1rank proxy check: student after 3 steps through hook-driven RCCL buckets", "== plain run (bit for bit)" if not bad else f"DIFFERS in {len(bad)} tensors, e.g. {bad[:3]}", flush=True)
dist.destroy_process_group()
sys.exit(1 if bad else 0)
