"""Worker of tests/test_gpu_ddp_step.py: one rank of a 2-rank data-parallel run of the DiGA steps on ONE GPU over gloo
(the collective layer is backend-agnostic; RCCL needs one GPU per rank).  Runs one warm-up step and one self-training
step on this rank's shard and writes what the parent checks: the student's parameters after each step, the centroid
bank, and the per-rank class sums that went into the rank-major all-gather."""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out_dir):
    os.environ["DIGA_DDP_BACKEND"] = "gloo"
    from diga_amd import _lib, config, ddp
    from diga_amd.calc_centroids import Class_Features
    from diga_amd.model import seg_model_noaux as sm
    from diga_amd.model.model_noaux import SegModel
    from diga_amd.train_step import DigaTrainer
    from oracle import deeplab as od
    from oracle import detweights, synth
    rank, world, local = ddp.init_from_env()
    if os.environ.get("DIGA_TEST_FORCE_STREAMS") == "1":
        # init_from_env puts gloo runs on one stream (gloo's host-synchronous GPU collectives make the side streams slow, not
        # wrong); the parent asks for them anyway to drive the three-stream self-training step under data parallelism
        config.DEFAULTS.teacher_stream = True
        config.DEFAULTS.wgrad_stream = True
    dev = torch.device("cuda", local)
    _lib.set_conv_math(int(os.environ.get("DIGA_TEST_MATH", "0")))

    def make():
        m = SegModel(arch=sm.TINY)
        m.load_state_dict(detweights.state_dict(od.TINY))
        m.final.head[0].p = 0.0
        return m.to(dev)

    student, teacher = make(), make()
    teacher.train()
    tr = DigaTrainer(student, teacher, rng=random.Random(500 + rank))
    assert tr.world == world and len(tr.reducer._hooks) > 0          # hook-driven buckets are live
    sent = []
    orig = ddp.gather_class_sums

    def spy(sums, counts, group=None):
        sent.append((sums.cpu().clone(), counts.cpu().clone()))
        return orig(sums, counts, group)
    ddp.gather_class_sums = spy
    forms = []
    for name in ("_selftrain_rest_overlapped", "_selftrain_tail_overlapped"):
        def wrap(fn, name=name):
            def run(*a, **k):
                forms.append(name)
                return fn(*a, **k)
            return run
        setattr(tr, name, wrap(getattr(tr, name)))
    wb = [t.to(dev) for t in synth.warmup_batch(900 + rank, 2, 96, 128, block=16)]
    log1 = tr.warmup_step(0, *wb)
    torch.cuda.synchronize()
    after1 = {k: v.detach().cpu().clone() for k, v in student.named_parameters()}
    cf = Class_Features(numbers=19)
    g7 = torch.Generator().manual_seed(7)
    cf.objective_vectors = torch.randn((19, 256), generator=g7).to(dev)
    sb = [t.to(dev) for t in synth.selftrain_batch(950 + rank, 2, 96, 128, block=16)]
    log2 = tr.selftrain_step(1, *sb, cf)
    torch.cuda.synchronize()
    after2 = {k: v.detach().cpu().clone() for k, v in student.named_parameters()}
    torch.save({"after1": after1, "after2": after2, "log1": {k: float(v) for k, v in log1.items()},
                "log2": {k: float(v) for k, v in log2.items()}, "cents": cf.objective_vectors.cpu(),
                "nums": cf.objective_vectors_num.cpu(), "sent": sent, "forms": forms}, os.path.join(out_dir, f"rank{rank}.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
