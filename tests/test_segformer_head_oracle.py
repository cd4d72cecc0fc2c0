"""oracle/segformer_head.py (the CPU restatement of the SegFormer decode head) against the capture of the REFERENCE class
(G5/model/networks/segformer_head.py:25-165, imported by tools/gen_golden.py::gen_segformer_head with mmcv's ConvModule stood
in as conv -> BN -> ReLU): logits, the concatenated raw features, input and parameter gradients in train mode on the stage maps of
a 128x96 image, and eval-mode logits on maps of odd sizes."""
import numpy as np
import torch

from oracle import segformer_head as oh
from oracle import synth


def test_state_dict_layout_matches_reference(golden):
    g = golden("segformer_head")
    assert list(oh.state_shapes().keys()) == g["keys"].tolist()


def test_forward_backward_vs_reference(golden):
    g = golden("segformer_head")
    sd = {k: (v.requires_grad_() if v.is_floating_point() and "running" not in k else v) for k, v in oh.state_dict().items()}
    feats = [g.t(f"c{i}").requires_grad_() for i in (1, 2, 3, 4)]
    logits, c_raw, _ = oh.forward(sd, feats, training=True)
    want = g.t("logits")
    assert float((logits.detach() - want).abs().max()) < 2e-5 * float(want.abs().max())
    assert float((c_raw.detach().reshape(-1)[::397] - g.t("c_raw_sample")).abs().max()) < 1e-5
    assert abs(synth.checksum(c_raw) - float(g["c_raw_sum"])) < 1e-4 * abs(float(g["c_raw_sum"])) + 1e-3
    (logits * g.t("probe")).sum().backward()
    for i, f in enumerate(feats):
        w = g.t(f"dc{i + 1}")
        assert float((f.grad - w).abs().max()) < 1e-4 * float(w.abs().max()), i
    for k in [n[2:] for n in g if n.startswith("g_")]:
        name = [n for n in sd if n.replace(".", "_") == k][0]
        step = int(g["gstep_" + k])
        w = g.t("g_" + k)
        got = sd[name].grad.reshape(-1)[::step]
        if name.endswith("proj.bias"):
            # a constant added in front of a train-mode BatchNorm has no effect: these gradients are rounding noise around zero
            assert float(w.abs().max()) < 2e-5 and float(got.abs().max()) < 2e-5, name
            continue
        assert float((got - w).abs().max()) < 2e-4 * float(w.abs().max()) + 1e-7, name
        assert abs(float(sd[name].grad.double().norm()) / float(g["gnorm_" + k]) - 1) < 1e-5, name
    # what the train-mode forward blended into the running buffers (momentum 0.1)
    mean, var = oh.batch_stats(sd, feats)
    rm = 0.9 * sd["linear_fuse.bn.running_mean"] + 0.1 * mean
    rv = 0.9 * sd["linear_fuse.bn.running_var"] + 0.1 * var
    assert torch.allclose(rm, g.t("running_mean"), rtol=1e-4, atol=1e-5)
    assert torch.allclose(rv, g.t("running_var"), rtol=1e-4, atol=1e-5)
    assert int(g["nbt"]) == 1


def test_eval_mode_odd_sizes_vs_reference(golden):
    g = golden("segformer_head")
    sd = oh.state_dict()
    sd["linear_fuse.bn.running_mean"] = g.t("running_mean")
    sd["linear_fuse.bn.running_var"] = g.t("running_var")
    feats = [g.t(f"e{i}") for i in (1, 2, 3, 4)]
    with torch.no_grad():
        logits, _, _ = oh.forward(sd, feats, training=False)
    want = g.t("logits_eval")
    assert tuple(logits.shape) == (1, 19, 25, 19)
    assert float((logits - want).abs().max()) < 2e-5 * float(want.abs().max())
