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
def test_translator_output(golden):
    from diga_amd.model.model_noaux import ImgDecoder, ImgEncoder
    g = golden("translator")
    enc, dec = _filled(ImgEncoder, "enc").cuda(), _filled(ImgDecoder, "dec").cuda()
    with torch.no_grad():
        feat = enc(g.t("x").cuda())
        rec = dec(feat)
    assert list(feat.shape) == g["feat_shape"].tolist()
    assert synth.checksum(feat.cpu()) == pytest.approx(float(g["feat_sum"]), rel=1e-3, abs=1e-2)
    assert_close(rec, g.t("rec"), 1e-3, 1e-4, "translated image")
