"""Normalisation / pooling modules of the DeepLabV2 model on the HIP kernels of csrc/norm.hip.

Drop-in subclasses (same parameters, buffers and state-dict keys) of the torch modules the reference uses
in G5/model/seg_model_noaux.py: nn.BatchNorm2d (frozen affine, batch statistics in train mode; fused with
the following ReLU and the residual add), nn.GroupNorm(32) (fused with ReLU / the Dropout2d channel
scale, able to write straight into a channel slice of the ASPP concat buffer), SEBlock pieces and the
stem max-pool.  Tensors cross module boundaries as NCHW-shaped views of NHWC memory.
"""
import torch
import torch.nn as nn

from diga_amd import _lib, config


def nhwc(x):
    """[N,C,H,W]-shaped tensor -> contiguous [N,H,W,C] fp32 tensor (no copy for channels_last input)."""
    t = x.permute(0, 2, 3, 1)
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def rows_ld(t):
    """Leading dimension (floats between consecutive pixels) of an [N,H,W,C] tensor whose pixels are a dense
    enumeration of rows of a wider [.., ld] matrix (a channel slice of a contiguous NHWC tensor), else None."""
    n, h, w, c = t.shape
    ld = t.stride(2)
    if t.stride(3) == 1 and ld >= c and t.stride(1) == w * ld and t.stride(0) == h * w * ld and ld % 4 == 0 \
            and t.data_ptr() % 16 == 0:
        return ld
    return None


def as_rows(t):
    """(tensor, ld) usable by the kernels; copies only when the layout is not row-sliceable."""
    ld = rows_ld(t)
    if ld is None:
        t = t.contiguous()
        ld = t.shape[3]
    return t, ld


def alias_slice(buf, lo, hi):
    """Channel slice [.., lo:hi] of a contiguous [N,H,W,C] buffer as a fresh (non-view) tensor on the same storage."""
    n, h, w, c = buf.shape
    return torch.empty(0, dtype=buf.dtype, device=buf.device).set_(
        buf.untyped_storage(), buf.storage_offset() + lo, (n, h, w, hi - lo), (h * w * c, w * c, c, 1))


def _ws(rows_per_seg, nseg, c, device):
    return _lib.workspace(_lib.lib.diga_norm_workspace_bytes(rows_per_seg, nseg, c), device, "norm")


# --------------------------------------------------------------------------------------------- BatchNorm
class _BnFn(torch.autograd.Function):
    """Train-mode BatchNorm (+ReLU, +residual).  `box` (a dict, or None) ties this BN to the convolution that consumes its
    output: that conv's backward-data kernel finishes this BN's incoming gradient in its epilogue -- adds the
    residual-branch gradient (`box['dres']`, put there by the backward of the BN that took this BN's output as residual),
    applies this BN's ReLU mask and reduces sum g / sum g*xhat -- and leaves (pointer, partials) in `box['premasked']`;
    backward() then only finalises and applies.  `res_box`: the box of the BN that produced `residual`."""

    @staticmethod
    def forward(ctx, x, residual, weight, bias, running_mean, running_var, training, relu, momentum, eps,
                partials=None, twin_out=False, dx_twin=False, box=None, res_box=None):
        _lib.require_gpu(x)
        xn = nhwc(x.detach())
        n, h, w, c = xn.shape
        m = n * h * w
        rn = None if residual is None else nhwc(residual.detach())
        y = torch.empty_like(xn)
        save_mean = torch.empty(c, dtype=torch.float32, device=xn.device)
        save_invstd = torch.empty_like(save_mean)
        # ReLU without residual: the backward re-derives the mask from x and the forward coefficients (no y read)
        save_ab = torch.empty(2 * c, dtype=torch.float32, device=xn.device) if (relu and residual is None) else None
        ws = _ws(m, 1, c, xn.device)
        # a BN with residual keeps its ReLU mask as one bit per element for the backward epilogue of the conv that reads
        # y (instead of y itself: 1/32 of the bytes); DIGA_RELU_BITS=0 reads y as before
        bits = None
        if (box is not None and relu and residual is not None and c % 32 == 0 and _relu_bits_enabled()):
            bits = torch.empty((m, c // 8), dtype=torch.uint8, device=xn.device)
        if partials is not None and training and isinstance(partials[1], tuple):
            # ... as records of unequal size (the Winograd output transform's tile groups): [R][3][C] sums, then [R] counts
            recs = int(partials[1][1])
            wsr = _lib.workspace(max(_lib.lib.diga_norm_workspace_bytes(m, 1, c), (96 * 3 * c + 2 * c + 128) * 4), xn.device, "norm_rec")
            _lib.call("diga_bn_fwd_records", _lib.ptr(xn), c, _lib.ptr(y), c, _lib.ptr(rn), c, _lib.ptr(weight),
                      _lib.ptr(bias), _lib.ptr(running_mean), _lib.ptr(running_var), _lib.ptr(save_mean),
                      _lib.ptr(save_invstd), _lib.ptr(save_ab), m, c, 1 if relu else 0, 1 if twin_out else 0,
                      _lib.ptr(bits), float(momentum), float(eps), _lib.ptr(partials[0]),
                      _lib.C.c_void_p(partials[0].data_ptr() + recs * 3 * c * 4), recs, _lib.ptr(wsr), wsr.numel(), _lib.stream())
        elif partials is not None and training:
            # the producing conv already reduced its output tile by tile: finalise + apply only
            _lib.call("diga_bn_fwd_partials", _lib.ptr(xn), c, _lib.ptr(y), c, _lib.ptr(rn), c, _lib.ptr(weight),
                      _lib.ptr(bias), _lib.ptr(running_mean), _lib.ptr(running_var), _lib.ptr(save_mean),
                      _lib.ptr(save_invstd), _lib.ptr(save_ab), m, c, 1 if relu else 0, 1 if twin_out else 0,
                      _lib.ptr(bits),
                      float(momentum), float(eps), _lib.ptr(partials[0]), int(partials[1]), _lib.ptr(ws), ws.numel(),
                      _lib.stream())
        else:
            _lib.call("diga_bn_fwd", _lib.ptr(xn), c, _lib.ptr(y), c, _lib.ptr(rn), c, _lib.ptr(weight), _lib.ptr(bias),
                      _lib.ptr(running_mean), _lib.ptr(running_var), _lib.ptr(save_mean), _lib.ptr(save_invstd),
                      _lib.ptr(save_ab), m, c, 1 if training else 0, 1 if relu else 0, 1 if twin_out else 0,
                      _lib.ptr(bits),
                      float(momentum), float(eps), _lib.ptr(ws), ws.numel(), _lib.stream())
        ctx.save_for_backward(xn, y if (relu and save_ab is None) else None, weight, save_mean, save_invstd, save_ab)
        ctx.flags = (training, residual is not None)
        ctx.dx_twin = bool(dx_twin and c % 8 == 0)
        ctx.box, ctx.res_box = box, res_box
        if box is not None:
            box.update(x=xn, mean=save_mean, invstd=save_invstd, relu_ab=save_ab, has_res=residual is not None, rows=m, C=c,
                       mask_bits=bits)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        xn, y, weight, save_mean, save_invstd, save_ab = ctx.saved_tensors
        training, has_res = ctx.flags
        n, h, w, c = xn.shape
        m = n * h * w
        g, ld_g = as_rows(gy.permute(0, 2, 3, 1))
        dx = torch.empty_like(xn)
        dgamma = dbeta = None
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
            # trainable affine pair (DigaTrainableBatchNorm2d): the two column sums of the input gradient are its gradients
            if has_res or ctx.box is not None:
                raise RuntimeError("DigaTrainableBatchNorm2d: no residual input / fused backward epilogue with a trainable affine pair")
            dgamma, dbeta = torch.empty(c, dtype=torch.float32, device=xn.device), torch.empty(c, dtype=torch.float32, device=xn.device)
            ws = _ws(m, 1, c, xn.device)
            _lib.call("diga_bn_bwd_affine", _lib.ptr(g), ld_g, _lib.ptr(xn), c, _lib.ptr(y), c, _lib.ptr(save_ab), _lib.ptr(weight),
                      _lib.ptr(save_mean), _lib.ptr(save_invstd), _lib.ptr(dx), c, _lib.ptr(dgamma), _lib.ptr(dbeta), m, c,
                      1 if training else 0, _lib.ptr(ws), ws.numel(), _lib.stream())
            return (dx.permute(0, 3, 1, 2), None, dgamma if ctx.needs_input_grad[2] else None, dbeta if ctx.needs_input_grad[3] else None,
                    None, None, None, None, None, None, None, None, None, None, None)
        pre = ctx.box.pop("premasked", None) if ctx.box is not None else None
        # (same buffer AND untouched since the conv wrote it: autograd sums a second gradient into a NEW tensor while the
        #  box holds a reference to this one; the version check also catches an in-place accumulation)
        if pre is not None and pre[0] == g.data_ptr() and pre[2]._version == pre[4] and ld_g == c and training:
            # the consumer conv's backward-data epilogue delivered g = mask * (its gradient + residual-branch gradient) and
            # its column sums: finalise + apply only; the residual gradient of this BN IS g
            ws = _lib.workspace(67 * c * 4, xn.device, "norm_kk")
            _lib.call("diga_bn_bwd_partials", _lib.ptr(g), ld_g, _lib.ptr(xn), c, _lib.ptr(weight), _lib.ptr(save_mean),
                      _lib.ptr(save_invstd), _lib.ptr(dx), c, m, c, 1 if ctx.dx_twin else 0, _lib.ptr(pre[1]), int(pre[5]),
                      _lib.ptr(ws), ws.numel(), _lib.stream())
            dres = g if has_res else None
        else:
            dres = torch.empty_like(xn) if has_res else None
            ws = _ws(m, 1, c, xn.device)
            _lib.call("diga_bn_bwd", _lib.ptr(g), ld_g, _lib.ptr(xn), c, _lib.ptr(y), c, _lib.ptr(save_ab), _lib.ptr(weight),
                      _lib.ptr(save_mean), _lib.ptr(save_invstd), _lib.ptr(dx), c, _lib.ptr(dres), c, m, c,
                      1 if training else 0, 1 if ctx.dx_twin else 0, _lib.ptr(ws), ws.numel(), _lib.stream())
        if dres is not None and ctx.res_box is not None and not ctx.res_box.get("consumed"):
            # hand the residual-branch gradient to the conv that reads the same tensor: its backward-data epilogue adds it
            # (autograd gets None here and therefore launches no add kernel)
            ctx.res_box["dres"] = dres
            dres = None
        return (dx.permute(0, 3, 1, 2), None if dres is None else dres.permute(0, 3, 1, 2),
                dgamma, dbeta, None, None, None, None, None, None, None, None, None, None, None)


class DigaBatchNorm2d(nn.BatchNorm2d):
    """BatchNorm2d whose affine parameters are frozen (the reference sets requires_grad=False on every BN,
    G5/model/seg_model_noaux.py:64-76) -- gradients flow to the input only.  forward(x, residual, relu)
    computes relu(bn(x) + residual) in one pass."""

    def forward(self, x, residual=None, relu=False, twin_out=False, dx_twin=False):
        """twin_out (ReLU, no residual, C % 8 == 0): the result is written as the split twin the staging-free conv
        kernels read (same 4 bytes per element, diga_make_twin's format) INSTEAD of fp32; the returned tensor has the
        usual shape and dtype but holds twin bytes (`_diga_is_twin`) -- only a DigaConv2d on the twin path may read it."""
        if self.weight.requires_grad or self.bias.requires_grad:
            raise RuntimeError("DigaBatchNorm2d implements the frozen-affine BN of the DiGA path; "
                               "set requires_grad=False on weight and bias")
        training = self.training or self.running_mean is None
        if getattr(self, "_nbt_external", False):
            self._nbt_external = False                # the model's forward bumped this counter for THIS call (bump_batches_tracked)
        elif self.training and self.num_batches_tracked is not None:
            self.num_batches_tracked.add_(1)
        twin_out = bool(twin_out and relu and residual is None and x.shape[1] % 8 == 0)
        box = res_box = None
        if training and relu and torch.is_grad_enabled() and x.requires_grad and fuse_backward_enabled():
            box = {}
            if residual is not None:
                rb = getattr(residual, "_diga_bn_box", None)
                if rb is not None and rb.get("consumer_ready") and "res_claimed" not in rb:
                    rb["res_claimed"] = True
                    res_box = rb
        y = _BnFn.apply(x, residual, self.weight, self.bias, self.running_mean, self.running_var, training, relu,
                        self.momentum, self.eps, getattr(x, "_diga_bn_partials", None), twin_out, bool(dx_twin), box, res_box)
        if twin_out:
            y._diga_is_twin = True
        if box is not None:
            y._diga_bn_box = box
        return y


class DigaTrainableBatchNorm2d(DigaBatchNorm2d):
    """BatchNorm2d (+ReLU) whose affine pair TRAINS: the BatchNorm of the SegFormer head's `linear_fuse` ConvModule
    (G5/model/networks/segformer_head.py:63-68, norm_cfg=dict(type='BN', requires_grad=True)).  Same forward kernels as the frozen
    BatchNorm of the DeepLab path; the backward is diga_bn_bwd_affine (the input gradient's two column sums are d gamma, d beta)."""

    def forward(self, x, relu=False):
        training = self.training or self.running_mean is None
        if getattr(self, "_nbt_external", False):
            self._nbt_external = False                # (bump_batches_tracked counted this call)
        elif self.training and self.num_batches_tracked is not None:
            self.num_batches_tracked.add_(1)
        return _BnFn.apply(x, None, self.weight, self.bias, self.running_mean, self.running_var, training, relu, self.momentum, self.eps,
                           getattr(x, "_diga_bn_partials", None), False, False, None, None)


def bump_batches_tracked(model):
    """num_batches_tracked += 1 for every DigaBatchNorm2d of `model` that is in train() mode, with ONE launch instead of one per
    layer (every BN of the DeepLab trunk runs exactly once per forward; the reference's nn.BatchNorm2d bumps its own counter per
    module and only while that module trains, seg_model_noaux.py:64-76 / torch).  The counters become views of one flat int64
    buffer (state-dict keys and values unchanged); the views are rebuilt whenever .to() / load_state_dict() replaced the
    buffers, the 0/1 increment vector whenever a module's train / eval state changed (e.g. a freeze_bn pattern)."""
    bns = [m for m in model.modules() if isinstance(m, DigaBatchNorm2d) and m.num_batches_tracked is not None]
    if not bns:
        return
    flat = getattr(model, "_nbt_flat", None)
    ok = (flat is not None and len(bns) == flat.numel() and flat.device == bns[0].num_batches_tracked.device
          and all(b.num_batches_tracked.data_ptr() == flat.data_ptr() + 8 * i for i, b in enumerate(bns)))
    if not ok:
        flat = torch.stack([b.num_batches_tracked.detach().reshape(()) for b in bns]).contiguous()
        for i, b in enumerate(bns):
            b.num_batches_tracked = flat[i]
        object.__setattr__(model, "_nbt_flat", flat)
        object.__setattr__(model, "_nbt_inc", None)
    state = tuple(b.training for b in bns)
    inc = getattr(model, "_nbt_inc", None)
    if inc is None or inc[0] != state:
        inc = (state, torch.tensor([1 if t else 0 for t in state], dtype=torch.int64).to(flat.device))
        object.__setattr__(model, "_nbt_inc", inc)
    for b in bns:
        b._nbt_external = True                 # this forward's increment is done here; the module skips its own
    if all(state):
        flat.add_(1)
    else:
        flat.add_(inc[1])


def _relu_bits_enabled():
    return config.active().relu_bits


def fuse_backward_enabled():
    """config.fuse_bwd = False switches the backward-epilogue fusion (residual add + ReLU mask + BN-backward sums inside the
    backward-data convolution) off: the A/B switch of the parity tests."""
    return config.active().fuse_bwd


# --------------------------------------------------------------------------------------------- GroupNorm
class _GnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps, relu, chan_scale, out):
        _lib.require_gpu(x)
        xn = nhwc(x.detach())
        n, h, w, c = xn.shape
        y = torch.empty_like(xn) if out is None else out
        y_, ld_y = (y, c) if out is None else (out, rows_ld(out))
        if ld_y is None:
            raise ValueError("GroupNorm output slice must be a channel slice of a contiguous NHWC tensor")
        save_mean = torch.empty(n * groups, dtype=torch.float32, device=xn.device)
        save_invstd = torch.empty_like(save_mean)
        cs = None if chan_scale is None else chan_scale.detach().float().contiguous()
        ws = _ws(h * w, n, c, xn.device)
        _lib.call("diga_gn_fwd", _lib.ptr(xn), c, _lib.ptr(y_), ld_y, _lib.ptr(weight.detach()), _lib.ptr(bias.detach()),
                  _lib.ptr(cs), _lib.ptr(save_mean), _lib.ptr(save_invstd), n, h * w, c, groups, 1 if relu else 0,
                  float(eps), _lib.ptr(ws), ws.numel(), _lib.stream())
        ctx.save_for_backward(xn, y_ if relu else None, weight, cs, save_mean, save_invstd)
        ctx.groups = groups
        return y_.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        xn, y, weight, cs, save_mean, save_invstd = ctx.saved_tensors
        n, h, w, c = xn.shape
        g, ld_g = as_rows(gy.permute(0, 2, 3, 1))
        ld_y = rows_ld(y) if y is not None else c
        dx = torch.empty_like(xn)
        dgamma = torch.empty(c, dtype=torch.float32, device=xn.device)
        dbeta = torch.empty_like(dgamma)
        ws = _ws(h * w, n, c, xn.device)
        _lib.call("diga_gn_bwd", _lib.ptr(g), ld_g, _lib.ptr(xn), c, _lib.ptr(y), ld_y, _lib.ptr(weight.detach()),
                  _lib.ptr(cs), _lib.ptr(save_mean), _lib.ptr(save_invstd), _lib.ptr(dx), c, _lib.ptr(dgamma),
                  _lib.ptr(dbeta), n, h * w, c, ctx.groups, _lib.ptr(ws), ws.numel(), _lib.stream())
        return dx.permute(0, 3, 1, 2), dgamma, dbeta, None, None, None, None, None


class DigaGroupNorm(nn.GroupNorm):
    """GroupNorm (trainable affine) with optional fused ReLU, fused per-(n,c) scale (Dropout2d) and an
    optional NHWC output slice to write into (the ASPP concat buffer)."""

    def forward(self, x, relu=False, chan_scale=None, out=None):
        return _GnFn.apply(x, self.weight, self.bias, self.num_groups, self.eps, relu, chan_scale, out)


class _AssembleFn(torch.autograd.Function):
    """The branches already wrote their channel slices into `buf`; tie them into one differentiable tensor
    without a copy.  Backward hands every branch the matching slice of the incoming gradient."""

    @staticmethod
    def forward(ctx, buf, width, *slices):
        ctx.width = width
        ctx.k = len(slices)
        return buf.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        gn = g.permute(0, 2, 3, 1)
        if not gn.is_contiguous():
            gn = gn.contiguous()
        outs = [alias_slice(gn, i * ctx.width, (i + 1) * ctx.width).permute(0, 3, 1, 2) for i in range(ctx.k)]
        return (None, None, *outs)


def assemble(buf, width, slices):
    return _AssembleFn.apply(buf, width, *slices)


# --------------------------------------------------------------------------------------------- SE block
class _AvgPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        xn = nhwc(x.detach())
        n, h, w, c = xn.shape
        out = torch.empty((n, c), dtype=torch.float32, device=xn.device)
        ws = _ws(h * w, n, c, xn.device)
        _lib.call("diga_avgpool_nhwc", _lib.ptr(xn), c, _lib.ptr(out), n, h * w, c, _lib.ptr(ws), ws.numel(), _lib.stream())
        ctx.shape = (n, h, w, c)
        return out

    @staticmethod
    def backward(ctx, g):
        n, h, w, c = ctx.shape
        return (g / float(h * w))[:, :, None, None].expand(n, c, h, w)


class _ChannelGateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gate):
        xn = nhwc(x.detach())
        n, h, w, c = xn.shape
        gt = gate.detach().float().contiguous()
        y = torch.empty_like(xn)
        _lib.call("diga_channel_affine", _lib.ptr(xn), c, _lib.ptr(y), c, _lib.ptr(gt), None, n, h * w, c, _lib.stream())
        ctx.save_for_backward(xn, gt)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        xn, gt = ctx.saved_tensors
        n, h, w, c = xn.shape
        g, ld_g = as_rows(gy.permute(0, 2, 3, 1))
        dx = torch.empty_like(xn)
        _lib.call("diga_channel_affine", _lib.ptr(g), ld_g, _lib.ptr(dx), c, _lib.ptr(gt), None, n, h * w, c, _lib.stream())
        dgate = torch.empty((n, c), dtype=torch.float32, device=xn.device)
        ws = _ws(h * w, n, c, xn.device)
        _lib.call("diga_channel_dot", _lib.ptr(g), ld_g, _lib.ptr(xn), c, _lib.ptr(dgate), n, h * w, c, _lib.ptr(ws),
                  ws.numel(), _lib.stream())
        return dx.permute(0, 3, 1, 2), dgate


class _SmallLinearFn(torch.autograd.Function):
    """y = act(x W^T + b) for the SE block's two dense layers ([N, 1280] -> 80 -> 1280) on diga_small_linear_fwd / _bwd."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        _lib.require_gpu(x, weight)
        xc, wc = x.detach().float().contiguous(), weight.detach().float().contiguous()
        bc = None if bias is None else bias.detach().float().contiguous()
        n, k = xc.shape
        o = wc.shape[0]
        y = torch.empty((n, o), dtype=torch.float32, device=xc.device)
        _lib.call("diga_small_linear_fwd", _lib.ptr(xc), _lib.ptr(wc), _lib.ptr(bc), _lib.ptr(y), n, k, o, act, _lib.stream())
        ctx.save_for_backward(xc, wc, y)
        ctx.act, ctx.has_bias = act, bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        xc, wc, y = ctx.saved_tensors
        n, k = xc.shape
        o = wc.shape[0]
        g = gy.float().contiguous()
        dz = torch.empty_like(y)
        dx = torch.empty_like(xc) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(wc) if ctx.needs_input_grad[1] else None
        db = torch.empty(o, dtype=torch.float32, device=xc.device) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        _lib.call("diga_small_linear_bwd", _lib.ptr(xc), _lib.ptr(wc), _lib.ptr(y), _lib.ptr(g), _lib.ptr(dz), _lib.ptr(dx), _lib.ptr(dw),
                  _lib.ptr(db), n, k, o, ctx.act, _lib.stream())
        return dx, dw, db, None


def small_linear(x, linear, act):
    """act(linear(x)) for an nn.Linear `linear` (a parameter container here); act: 0 none, 1 ReLU, 2 sigmoid."""
    return _SmallLinearFn.apply(x, linear.weight, linear.bias, act)


def global_avg_pool(x):
    return _AvgPoolFn.apply(x)


def channel_gate(x, gate):
    return _ChannelGateFn.apply(x, gate)


# --------------------------------------------------------------------------------------------- max-pool
class _MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        xn = nhwc(x.detach())
        n, h, w, c = xn.shape

        def osz(i):
            o = -(-(i + 2 - 3) // 2) + 1                   # ceil_mode
            return o - 1 if (o - 1) * 2 - 1 >= i else o    # last window must start inside the input

        ho, wo = osz(h), osz(w)
        y = torch.empty((n, ho, wo, c), dtype=torch.float32, device=xn.device)
        idx = torch.empty((n, ho, wo, c), dtype=torch.uint8, device=xn.device)
        _lib.call("diga_maxpool3x3s2_fwd", _lib.ptr(xn), _lib.ptr(y), _lib.ptr(idx), n, h, w, c, ho, wo, _lib.stream())
        ctx.save_for_backward(idx)
        ctx.shape = (n, h, w, c, ho, wo)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        (idx,) = ctx.saved_tensors
        n, h, w, c, ho, wo = ctx.shape
        g = gy.permute(0, 2, 3, 1)
        if not g.is_contiguous():
            g = g.contiguous()
        dx = torch.empty((n, h, w, c), dtype=torch.float32, device=g.device)
        _lib.call("diga_maxpool3x3s2_bwd", _lib.ptr(g), _lib.ptr(idx), _lib.ptr(dx), n, h, w, c, ho, wo, _lib.stream())
        return dx.permute(0, 3, 1, 2)


class DigaMaxPool3x3s2(nn.MaxPool2d):
    """nn.MaxPool2d(3, stride=2, padding=1, ceil_mode=True) of the stem."""

    def __init__(self):
        super().__init__(kernel_size=3, stride=2, padding=1, ceil_mode=True)

    def forward(self, x):
        return _MaxPoolFn.apply(x)
