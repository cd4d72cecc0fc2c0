"""oracle/mit.py (the CPU restatement of the MiT encoder) against captures of the REFERENCE module
(G5/model/networks/MixTransfomer.py, imported by tools/gen_golden.py::gen_mit): features and parameter gradients of mit_b5
on a small non-square input, and features of mit_b1 at the benchmark geometry (768x768: 36864 queries x 576 keys in stage 1)."""
import numpy as np
import pytest
import torch

from oracle import mit as om
from oracle import synth


def test_state_dict_layout_matches_reference(golden):
    g = golden("mit")
    assert list(om.state_shapes(om.MIT_B5).keys()) == g["keys"].tolist()
    assert sum(int(np.prod(s)) for s, _ in om.state_shapes(om.MIT_B5).values()) == 81443008


@pytest.mark.timeout(600)
def test_mit_b5_forward_backward_vs_reference(golden):
    g = golden("mit")
    sd = {k: v.requires_grad_() for k, v in om.state_dict(om.MIT_B5).items()}
    outs = om.forward(sd, g.t("x"), om.MIT_B5)
    for i, o in enumerate(outs):
        want = g.t(f"c{i + 1}")
        assert float((o - want).abs().max()) < 2e-5 * float(want.abs().max()), i
    sum((o * g.t(f"probe{i + 1}")).sum() for i, o in enumerate(outs)).backward()
    norms = np.array([float(sd[k].grad.norm()) for k in g["keys"].tolist()])
    assert np.allclose(norms, g["grad_norms"], rtol=2e-4, atol=1e-7)
    for k in [n[2:] for n in g if n.startswith("g_")]:
        name = [n for n in sd if n.replace(".", "_") == k][0]
        step = int(g["gstep_" + k])
        want = g.t("g_" + k)
        got = sd[name].grad.reshape(-1)[::step]
        assert float((got - want).abs().max()) < 1e-4 * float(want.abs().max()) + 1e-7, name


@pytest.mark.timeout(600)
def test_mit_b1_benchmark_geometry_vs_reference(golden):
    g = golden("mit768")
    x = torch.rand((1, 3, 768, 768), generator=synth.gen(int(g["seed"]))) * 2 - 1
    with torch.no_grad():
        outs = om.forward(om.state_dict(om.MIT_B1), x, om.MIT_B1)
    assert [tuple(o.shape) for o in outs] == [(1, 64, 192, 192), (1, 128, 96, 96), (1, 320, 48, 48), (1, 512, 24, 24)]
    for o, key, step, mx in zip(outs, ("c1_sample", "c2_sample", "c3_sample", "c4_sample"), (211, 53, 7, 3), g["maxs"]):
        assert float((o.reshape(-1)[::step] - g.t(key)).abs().max()) < 2e-5 * float(mx)
    assert np.allclose([float(o.abs().sum()) for o in outs], g["sums"], rtol=1e-5)


@pytest.mark.timeout(900)
def test_mit_b5_benchmark_geometry_vs_reference(golden):
    g = golden("mit768")
    x = torch.rand((1, 3, 768, 768), generator=synth.gen(int(g["seed"]))) * 2 - 1
    with torch.no_grad():
        outs = om.forward(om.state_dict(om.MIT_B5), x, om.MIT_B5)
    for o, key, step, mx in zip(outs, ("b5_c1_sample", "b5_c2_sample", "b5_c3_sample", "b5_c4_sample"), (211, 53, 7, 3), g["b5_maxs"]):
        assert float((o.reshape(-1)[::step] - g.t(key)).abs().max()) < 3e-5 * float(mx)
    assert np.allclose([float(o.abs().sum()) for o in outs], g["b5_sums"], rtol=1e-5)
