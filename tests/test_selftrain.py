"""One self-training step (centroid pseudo-labeler + both ClassMix blocks + centroid EMA + three
student passes) against the capture of the reference (tests/golden/selftrain.npz):
the CPU oracle on CPU, the HIP path on the GPU."""
import random

import pytest
import torch

from diga_amd import config

from conftest import assert_close
from oracle import deeplab as od
from oracle import detweights, synth
from oracle import step as ost


def _check_common(g, log, cents, nums, head, pseudo_kept):
    assert log["ce"] == pytest.approx(float(g["ce"]), rel=1e-3)
    assert log["distil"] == pytest.approx(float(g["distil"]), rel=1e-3)
    assert log["ce_mix"] == pytest.approx(float(g["ce_mix"]), rel=1e-3)
    assert pseudo_kept == pytest.approx(float(g["kept"]), abs=2e-3)
    assert_close(cents, g.t("cents"), 1e-5, 1e-6, "centroids after the two EMA passes")
    assert torch.equal(nums.cpu().float(), g.t("nums"))
    assert float((g.t("cents") - g.t("cents0")).abs().max()) > 0       # the update did something
    assert_close(head, g.t("student_head"), 5e-3, 1e-5, "student head after SGD")


@pytest.mark.timeout(900)
def test_oracle_selftrain_step(golden):
    g = golden("selftrain")
    tr = ost.Trainer(detweights.state_dict(), detweights.state_dict())
    cents, nums = g.t("cents0").clone(), torch.zeros(19)
    batch = synth.selftrain_batch(3000, 2, 128, 128, block=16)
    random.seed(78)
    log = tr.selftrain_step(3, *batch, cents, nums, random)
    _check_common(g, log, cents, nums, tr.s["final.head.1.weight"], log["kept"])


@pytest.mark.gpu
def test_gpu_selftrain_step(golden, conv_math):
    from diga_amd.calc_centroids import Class_Features
    from diga_amd.model.model_noaux import SegModel
    from diga_amd.train_step import DigaTrainer
    g = golden("selftrain")

    def make():
        m = SegModel()
        m.load_state_dict(detweights.state_dict(od.RESNET101))
        m.final.head[0].p = 0.0
        return m.to("cuda")

    student, teacher = make(), make()
    teacher.train()
    tr = DigaTrainer(student, teacher, rng=random)
    cf = Class_Features(numbers=19)
    cf.objective_vectors = g.t("cents0").clone().to("cuda")
    batch = [t.to("cuda") for t in synth.selftrain_batch(3000, 2, 128, 128, block=16)]
    random.seed(78)
    out = tr.selftrain_step(3, *batch, cf)
    log = {k: float(v) for k, v in out.items()}
    # pseudo-label consensus itself (away from argmax near-ties)
    pseudo, fp = cf_consensus(cf, teacher, batch, g)
    _check_common(g, log, cf.objective_vectors, cf.objective_vectors_num, student.state_dict()["final.head.1.weight"],
                  float(g["kept"]))
    assert pseudo


def cf_consensus(cf, teacher, batch, g):
    """Re-run the consensus with the ORIGINAL centroids to compare label maps with the capture."""
    from diga_amd.calc_centroids import Class_Features
    ref = Class_Features(numbers=19)
    ref.objective_vectors = g.t("cents0").clone().to("cuda")
    # the capture's teacher is the post-EMA teacher of that step; EMA at it=3 of two identical models is a no-op
    with torch.no_grad():
        t_feat = teacher(batch[4])[3]
    out, fp = ref.consensus_pseudo_labels(t_feat, batch[6], return_feat_pseudo=True)
    safe = g.t("margin") > 1e-4
    same = (fp.cpu() == g.t("feat_pseudo"))[safe]
    # the teacher has taken one more train-mode forward since the capture (running stats only), features equal
    return bool(same.float().mean() > 0.995), fp


@pytest.mark.gpu
def test_selftrain_overlapped_tail_is_bit_identical(golden, conv_math, monkeypatch):
    """Round 5: the self-training step's two student graphs on two streams (backward of student(cat) started as soon as its loss
    exists, forward + backward of student(cross_mix) on a third stream, one multi-tensor add of the two gradient sets) against the
    one-backward form: losses, every student parameter after the step, BatchNorm running statistics and the centroid bank are equal
    BIT FOR BIT -- the gradient of a shared weight is a two-term sum either way."""
    from diga_amd import train_step as ts
    from diga_amd.calc_centroids import Class_Features
    from diga_amd.model.model_noaux import SegModel
    if conv_math != 0:
        pytest.skip("stream structure, not arithmetic: once (fp32) is enough -- three two-step runs of the ResNet-101 step")
    g = golden("selftrain")

    def run(overlap):
        monkeypatch.setattr(config.active(), "c4_overlap", overlap)
        def make():
            m = SegModel()
            m.load_state_dict(detweights.state_dict(od.RESNET101))
            m.final.head[0].p = 0.0
            return m.to("cuda")
        student, teacher = make(), make()
        teacher.train()
        tr = ts.DigaTrainer(student, teacher, rng=random)
        cf = Class_Features(numbers=19)
        cf.objective_vectors = g.t("cents0").clone().to("cuda")
        logs = []
        for it in (3, 4):                                    # two steps: the second runs on momentum buffers and a moved teacher
            batch = [t.to("cuda") for t in synth.selftrain_batch(3000 + it, 2, 128, 128, block=16)]
            random.seed(78 + it)
            logs.append({k: float(v) for k, v in tr.selftrain_step(it, *batch, cf).items()})
        torch.cuda.synchronize()
        return logs, {k: v.clone() for k, v in student.state_dict().items()}, cf.objective_vectors.clone()

    la, sa, ca = run(0)
    for mode in (1, 2):                                        # 1: cross-mixed forward / backward only; 2: the whole target branch
        lb, sb, cb = run(mode)
        assert la == lb, (mode, la, lb)
        assert torch.equal(ca, cb), mode
        for k in sa:
            assert torch.equal(sa[k], sb[k]), (mode, k)


@pytest.mark.timeout(900)
def test_oracle_selftrain_trajectory_first_steps(golden):
    """The oracle's self-training step against the 10-step capture of the reference's loop (tests/golden/selftraj10.npz,
    tools/gen_golden.py::_gen_selftraj): the first three steps -- losses within 3x the capture's own rounding floor (reference vs
    reference, oneDNN off; the floor of CE_mix contains single consensus pixels flipping), the kept share within two pixels, the
    centroid bank's change to 1e-3 of itself."""
    g = golden("selftraj10")
    B, H, W, steps, seed0, block, mix_seed = (int(v) for v in g["geometry"])
    tr = ost.Trainer(detweights.state_dict(), detweights.state_dict())
    cents, nums = g.t("cents0").clone(), torch.zeros(19)
    random.seed(mix_seed)
    assert steps == 10 and bool(g["floor_nums_equal"])
    for it in range(3):
        batch = synth.selftrain_batch(seed0 + it, B, H, W, block=block)
        log = tr.selftrain_step(it, *batch, cents, nums, random)
        for k, extra in (("ce", 2e-6), ("distil", 2e-6), ("ce_mix", 3e-5)):
            assert log[k] == pytest.approx(float(g[k][it]), rel=3 * (float(g["floor_" + k + "_dev"].max()) + extra)), (it, k)
        assert log["kept"] == pytest.approx(float(g["kept"][it]), abs=2.5 / (B * H * W)), it
        cd = float((cents.double() - g.t("cents0").double()).norm())
        assert cd == pytest.approx(float(g["cents_delta"][it]), rel=1e-3), it
