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


def test_deferred_bn1_apply_is_bit_identical(monkeypatch):
    """fp32: bn1 of a bottleneck computes statistics and coefficients only and conv2's Winograd input transform applies
    relu(fma(y1, a, b)) on load (diga_conv2d_winograd_f32_ab; the weight gradient reads the kept transform or re-applies the
    coefficients): output, input gradient and every weight gradient equal the run with the stand-alone apply pass bit for bit."""
    from diga_amd import _lib
    from diga_amd.model import conv as _dconv
    from diga_amd.model import seg_model_noaux as sm
    torch.manual_seed(3)
    blk = sm.Bottleneck(1024, 256, 1, dilation=2).to(DEV).train()
    g = torch.Generator().manual_seed(8)
    x0 = torch.randn((2, 1024, 19, 17), generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    probe = torch.randn((2, 1024, 19, 17), generator=g).to(DEV)
    prev = _lib.get_conv_math()
    _lib.set_conv_math(0)
    calls, real = [], _lib.call
    monkeypatch.setattr(_lib, "call", lambda name, *a: (calls.append(name), real(name, *a))[1])
    try:
        res = []
        for fuse, keep in (("0", "1"), ("1", "1"), ("1", "0")):
            monkeypatch.setattr(_dconv, "FUSE_BN1", fuse == "1")
            monkeypatch.setattr(_dconv, "WINOGRAD_KEEP_V", keep == "1")
            for p_ in blk.parameters():
                p_.grad = None
            sd = {k: v.clone() for k, v in blk.state_dict().items()}
            calls.clear()
            x = x0.clone().requires_grad_()
            y = blk(x)
            (y * probe).sum().backward()
            assert ("diga_conv2d_winograd_f32_ab" in calls) == (fuse == "1"), calls
            assert ("diga_conv2d_wgrad_winograd_f32_ab" in calls) == (fuse == "1"), calls
            res.append([y.detach().clone(), x.grad.clone()] + [getattr(blk, c).weight.grad.clone() for c in ("conv1", "conv2", "conv3")]
                       + [blk.bn1.running_mean.clone(), blk.bn1.running_var.clone()])
            blk.load_state_dict(sd)              # (running statistics back to where they were)
    finally:
        _lib.set_conv_math(prev)
    for other in res[1:]:
        for a, b in zip(res[0], other):
            assert torch.equal(a, b)


def test_fused_residual_junction_is_bit_identical(monkeypatch):
    """fp32: inside a whole-network forward (norm.junction_fusion) bn3 of a bottleneck leaves relu(bn3(y3) + skip) to conv1 of the next
    block, whose persistent GEMM applies it while staging its operand and stores the activated tensor + ReLU mask bits
    (diga_conv2d_junction_f32).  Three chained layer3-width blocks: outputs, input gradient, every weight gradient and the BatchNorm
    running statistics equal the run with stand-alone apply passes bit for bit; a consumer that cannot fuse (here: the chain called
    with a conv1 the GEMM does not take) falls back to the stand-alone pass."""
    from diga_amd import _lib
    from diga_amd.model import norm as dn
    from diga_amd.model import seg_model_noaux as sm
    monkeypatch.setattr(dn, "JUNCTION_FUSION", 2)              # every eligible consumer (the default fuses single-column-tile ones only)
    torch.manual_seed(5)
    blocks = torch.nn.Sequential(*[sm.Bottleneck(1024, 256, 1, dilation=2) for _ in range(3)]).to(DEV).train()
    for b in blocks[:-1]:
        b.defer_out, b.next_conv1 = True, (256, (1, 1))
    g = torch.Generator().manual_seed(9)
    x0 = torch.randn((8, 1024, 97, 97), generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    probe = torch.randn((8, 1024, 97, 97), generator=g).to(DEV)
    prev = _lib.get_conv_math()
    _lib.set_conv_math(0)
    calls, real = [], _lib.call
    monkeypatch.setattr(_lib, "call", lambda name, *a: (calls.append(name), real(name, *a))[1])
    try:
        res = []
        for fused in (False, True):
            for p_ in blocks.parameters():
                p_.grad = None
            sd = {k: v.clone() for k, v in blocks.state_dict().items()}
            calls.clear()
            x = x0.clone().requires_grad_()
            if fused:
                with dn.junction_fusion():
                    y = blocks(x)
            else:
                y = blocks(x)
            (y * probe).sum().backward()
            assert calls.count("diga_conv2d_junction_f32") == (2 if fused else 0), calls.count("diga_conv2d_junction_f32")
            assert calls.count("diga_bn_apply") == 0
            res.append([y.detach().clone(), x.grad.clone()] + [p_.grad.clone() for p_ in blocks.parameters() if p_.grad is not None]
                       + [v.clone() for k, v in blocks.state_dict().items() if "running" in k])
            blocks.load_state_dict(sd)
        for a, b in zip(res[0], res[1]):
            assert torch.equal(a, b)
        # a deferred junction whose consumer cannot fuse: materialised by the stand-alone pass, same bits
        calls.clear()
        with dn.junction_fusion():
            mid = blocks[0](x0)
            assert getattr(mid, "_diga_lazy_junction", None) is not None and not mid._diga_lazy_junction["filled"]
            narrow = sm.Bottleneck(1024, 16, 1, dilation=2).to(DEV).train()          # conv1 into 16 channels: not a persistent-GEMM shape
            narrow(mid)
        assert calls.count("diga_bn_apply") == 1 and mid._diga_lazy_junction["filled"]
        with torch.no_grad():
            want = blocks[0](x0)
        assert torch.equal(mid.detach(), want)
    finally:
        _lib.set_conv_math(prev)
