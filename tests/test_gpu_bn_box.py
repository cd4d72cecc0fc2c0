"""The residual-gradient hand-off between a BatchNorm and the convolution that reads its output (model/norm.py `_BnFn`,
model/conv.py `bn_box`) for compositions the in-tree Bottleneck never builds (round-2 advisor findings): a ReLU BatchNorm
WITHOUT residual whose output feeds a stride-1 conv and is reused as a later BatchNorm's residual, in both backward
orders; and a second backward pass through a gradient chain must raise instead of returning zeros."""
import pytest
import torch
import torch.nn.functional as F

from oracle import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _bn_ref(x, w, b, residual=None, relu=True):
    y = F.batch_norm(x, None, None, w, b, True, 0.1, 1e-5)
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


def _mods(g, c):
    from diga_amd.model.conv import DigaConv2d
    from diga_amd.model.norm import DigaBatchNorm2d
    convs = [DigaConv2d(c, c, k, padding=k // 2, bias=False) for k in (3, 1, 3)]
    bns = [DigaBatchNorm2d(c) for _ in range(3)]
    for m in convs:
        with torch.no_grad():
            m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / (c * m.kernel_size[0] ** 2)) ** 0.5)
    for m in bns:
        with torch.no_grad():
            m.weight.copy_(1.0 + 0.1 * torch.randn(c, generator=g))
            m.bias.copy_(0.1 * torch.randn(c, generator=g))
        m.weight.requires_grad_(False)
        m.bias.requires_grad_(False)
    return [m.to(DEV).train() for m in convs], [m.to(DEV).train() for m in bns]


@pytest.mark.parametrize("order", ["residual_bn_first", "conv_first"])
def test_residual_reuse_of_a_bn_without_residual(order):
    g = synth.gen(31 if order == "conv_first" else 30)
    c = 64
    convs, bns = _mods(g, c)
    x = torch.randn((2, c, 19, 17), generator=g)
    z = torch.randn((2, c, 19, 17), generator=g)
    p1 = torch.randn((2, c, 19, 17), generator=g)
    p2 = torch.randn((2, c, 19, 17), generator=g)

    def run(xd, zd, conv, bn):
        a = bn[0](conv[0](xd), None, True)                       # ReLU, no residual; feeds conv[1] AND is a residual below
        t = conv[1](a)
        if order == "residual_bn_first":
            y = bn[1](t, a, True)                                # main input derives from the consumer conv
            return (y * p1.to(y.device, y.dtype)).sum()
        # the residual BatchNorm's main input does NOT derive from conv[1]: its backward may run after conv[1]'s
        y = bn[1](conv[2](zd), a, True)
        u = bn[2](t, None, True)
        return (u * p1.to(u.device, u.dtype)).sum() + (y * p2.to(y.device, y.dtype)).sum()

    xd = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_()
    zd = z.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_()
    run(xd, zd, convs, [lambda t, r, relu, m=m: m(t, residual=r, relu=relu) for m in bns]).backward()

    w64 = [m.weight.detach().cpu().double().contiguous().requires_grad_() for m in convs]
    bnp = [(m.weight.detach().cpu().double(), m.bias.detach().cpu().double()) for m in bns]
    xr, zr = x.double().requires_grad_(), z.double().requires_grad_()
    ref_convs = [lambda t, w=w, k=m.kernel_size[0]: F.conv2d(t, w, None, 1, k // 2) for w, m in zip(w64, convs)]
    ref_bns = [lambda t, r, relu, wb=wb: _bn_ref(t, wb[0], wb[1], r, relu) for wb in bnp]
    run(xr, zr, ref_convs, ref_bns).backward()

    def rel(a, b):
        return float((a.detach().cpu().double() - b).abs().max() / b.abs().max())

    assert rel(xd.grad, xr.grad) < 2e-4, rel(xd.grad, xr.grad)
    for i in range(3 if order == "conv_first" else 2):
        assert rel(convs[i].weight.grad, w64[i].grad) < 2e-4, i
    if order == "conv_first":
        assert rel(zd.grad, zr.grad) < 2e-4


def test_second_backward_through_a_gradient_chain_raises():
    g = synth.gen(33)
    # two convs on one BatchNorm output, chained like the ASPP branches
    convs, bns = _mods(g, 64)
    x = torch.randn((1, 64, 9, 9), generator=g).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_()
    a = bns[0](convs[0](x), relu=True)
    chain = {"remaining": 2, "acc": None, "box": getattr(a, "_diga_bn_box", None)}
    y = convs[1](a, chain=chain) + convs[2](a, chain=chain)
    loss = y.sum()
    loss.backward(retain_graph=True)
    assert x.grad is not None and float(x.grad.abs().sum()) > 0
    with pytest.raises(RuntimeError, match="second backward"):
        loss.backward()
