"""Translator (ImgEncoder/ImgDecoder): state-dict compatibility on CPU, output parity
with the capture of the reference on the GPU."""
import pytest
import torch

from conftest import assert_close
from oracle import detweights, synth


def _filled(cls, tag):
    m = cls()
    m.load_state_dict({k: detweights.fill(f"translator.{tag}.{k}", tuple(v.shape), "conv" if v.dim() == 4 else "bias")
                       for k, v in m.state_dict().items()})
    return m.eval()


def test_translator_state_dict_keys(golden):
    from diga_amd.model.model_noaux import ImgDecoder, ImgEncoder
    g = golden("translator")
    assert list(ImgEncoder().state_dict().keys()) == g["enc_keys"].tolist()
    assert list(ImgDecoder().state_dict().keys()) == g["dec_keys"].tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("folded", [True, False], ids=["folded_no_grad", "explicit_ops"])
def test_translator_output(golden, conv_math, folded, monkeypatch):
    """dec_s2t(enc_s(x)) against the capture of the reference: with the reflection pads / nearest upsampling / tanh
    folded into the conv kernels and InstanceNorm on the HIP kernels (no_grad, how the DiGA scripts call it), and with
    explicit pad / upsample / tanh ops around the convs (autograd enabled)."""
    from diga_amd import _lib
    from diga_amd.model.model_noaux import ImgDecoder, ImgEncoder
    g = golden("translator")
    enc, dec = _filled(ImgEncoder, "enc").cuda(), _filled(ImgDecoder, "dec").cuda()
    calls = []
    orig = _lib.call

    def counting(name, *a):
        calls.append(name)
        return orig(name, *a)
    monkeypatch.setattr(_lib, "call", counting)
    with (torch.no_grad() if folded else torch.enable_grad()):
        feat = enc(g.t("x").cuda())
        rec = dec(feat)
    # folded: every conv but the 3-channel stem arms the input map (reflect; + upsample for the two decoder blocks; + tanh)
    n_opts = sum(1 for c in calls if c.endswith("_opts"))
    assert (n_opts == 21) == folded, n_opts
    assert calls.count("diga_gn_fwd") == 21                      # every InstanceNorm on the HIP kernels
    assert list(feat.shape) == g["feat_shape"].tolist()
    assert synth.checksum(feat.detach().cpu()) == pytest.approx(float(g["feat_sum"]), rel=1e-3, abs=1e-2)
    assert_close(rec, g.t("rec"), 1e-3, 1e-4, "translated image")
