"""Worker of tests/test_gpu_ddp_step.py::test_rccl_single_rank_runs_the_production_reducer_path: ONE rank of a real "nccl" (= RCCL)
process group on the one GPU of the test box, with the gradient reducer forced active (config.ddp_single_rank) -- so that the code a
multi-GPU run executes DOES execute under the driver's GPU test: hook-driven 25 MB buckets launched from inside backward on RCCL's
stream next to the teacher / weight-gradient side streams, gradients living in the bucket slices (grad views), the self-training
step in its three-stream form with the hooks held and the buckets sent per bucket after the join, the packed class-sum all-gather.
With one rank every collective is the identity, so the student after three steps must equal, BIT FOR BIT, the student of the same
three steps without any reducer.  Exits non-zero on any difference.  Nothing touches the GPU before the process group exists."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", LOCAL_WORLD_SIZE="1")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    import torch
    import torch.distributed as dist
    from diga_amd import config, ddp
    rank, world, local = ddp.init_from_env(backend="nccl", single_rank_group=True)
    assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1
    from diga_amd import _lib
    from diga_amd.calc_centroids import Class_Features
    from diga_amd.model import seg_model_noaux as sm
    from diga_amd.model.model_noaux import SegModel
    from diga_amd.train_step import DigaTrainer
    from oracle import deeplab as od
    from oracle import detweights, synth
    dev = torch.device("cuda", local)
    _lib.set_conv_math(0)
    arch_name = os.environ.get("DIGA_TEST_ARCH", "RESNET101")

    def make():
        m = SegModel(arch=getattr(sm, arch_name))
        m.load_state_dict(detweights.state_dict(getattr(od, arch_name)))
        m.final.head[0].p = 0.0
        return m.to(dev)

    calls = {"all_reduce": 0, "all_gather": 0}
    real_ar, real_ag = dist.all_reduce, dist.all_gather_into_tensor

    def count_ar(*a, **k):
        calls["all_reduce"] += 1
        return real_ar(*a, **k)

    def count_ag(*a, **k):
        calls["all_gather"] += 1
        return real_ag(*a, **k)
    dist.all_reduce, dist.all_gather_into_tensor = count_ar, count_ag

    def run(kind, reducer_on, overlap):
        cfg = config.DEFAULTS.replace(ddp_single_rank=reducer_on, c4_overlap=overlap)
        assert cfg.teacher_stream and cfg.wgrad_stream          # RCCL keeps the side streams (gloo runs switch them off)
        student, teacher = make(), make()
        teacher.train()
        tr = DigaTrainer(student, teacher, rng=random.Random(7), config=cfg)
        assert tr.reducer.active == reducer_on and (len(tr.reducer._hooks) > 0) == reducer_on and tr.reducer.as_views == reducer_on
        if reducer_on:
            assert sum(1 for p in student.parameters() if getattr(p, "_diga_grad_view", None) is not None) > 50
        cf = Class_Features(numbers=19)
        cf.objective_vectors = torch.randn((19, 256), generator=synth.gen(7)).to(dev)
        forms = []
        for name in ("_selftrain_rest_overlapped", "_selftrain_tail_overlapped"):
            def wrap(fn, name=name):
                def go(*a, **k):
                    forms.append(name)
                    return fn(*a, **k)
                return go
            setattr(tr, name, wrap(getattr(tr, name)))
        logs = []
        for it in range(3):
            if kind == "warmup":
                batch = [t.to(dev) for t in synth.warmup_batch(600 + it, 2, 128, 128, block=16)]
                out = tr.warmup_step(it, *batch)
            else:
                batch = [t.to(dev) for t in synth.selftrain_batch(700 + it, 2, 128, 128, block=16)]
                out = tr.selftrain_step(it, *batch, cf)
            logs.append({k: float(v) for k, v in out.items()})
        torch.cuda.synchronize()
        if reducer_on:
            # the gradients the optimizer read ARE the bucket slices
            flat_ptrs = {(f.data_ptr(), f.data_ptr() + f.numel() * 4) for f in tr.reducer._flat if f is not None}
            inside = sum(1 for p in student.parameters() if p.grad is not None and any(lo <= p.grad.data_ptr() < hi for lo, hi in flat_ptrs))
            assert inside > 50, inside
        tr.reducer.close()
        return logs, {k: v.detach().clone() for k, v in student.state_dict().items()}, cf.objective_vectors.clone(), forms

    failures = []
    for kind, overlap, want_form in (("warmup", 2, []), ("selftrain", 2, ["_selftrain_rest_overlapped"] * 3),
                                     ("selftrain", 1, ["_selftrain_tail_overlapped"] * 3), ("selftrain", 0, [])):
        before = dict(calls)
        la, sa, ca, fa = run(kind, True, overlap)
        n_ar, n_ag = calls["all_reduce"] - before["all_reduce"], calls["all_gather"] - before["all_gather"]
        lb, sb, cb, fb = run(kind, False, overlap)
        assert fa == want_form and fb == want_form, (kind, overlap, fa, fb)
        assert n_ar >= 3 * 2, (kind, n_ar)                        # several buckets per step went through RCCL
        if kind == "selftrain":
            assert n_ag == 3 * 2, n_ag                            # target + source class sums per step through the RCCL all-gather
        bad = [k for k in sa if not torch.equal(sa[k], sb[k])]
        if bad or la != lb or not torch.equal(ca, cb):
            failures.append((kind, overlap, bad[:3], la[-1], lb[-1]))
        print(f"rccl 1-rank {kind} overlap={overlap}: {n_ar} all-reduces, {n_ag} all-gathers on RCCL; student "
              + ("== plain run bit for bit" if not bad else f"DIFFERS in {len(bad)} tensors"), flush=True)
    dist.destroy_process_group()
    if failures:
        print("FAILED", failures, flush=True)
        sys.exit(1)
    print("rccl single-rank worker OK", flush=True)


if __name__ == "__main__":
    main()
