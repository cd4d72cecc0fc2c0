"""Mix Transformer (MiT-B0..B5) encoder on the MI355X kernels of include/diga_mit.h -- the backbone of BASELINE.json
configs[4] ("SegFormer-B5 (MiT transformer) backbone variant, fp16").

Mirror of G5/model/networks/MixTransfomer.py (the reference's file name, typo included): same class names (Mlp,
Attention, Block, OverlapPatchEmbed, DWConv, MixVisionTransformer, mit_b0..mit_b5), same constructor arguments, same
module tree and therefore the same state-dict keys and parameter shapes (`patch_embed1.proj.weight`,
`block3.17.attn.sr.weight`, `block1.0.mlp.dwconv.dwconv.weight`, ...), same `forward(x) -> [c1, c2, c3, c4]` (NCHW fp32,
strides 4/8/16/32).  The modules are parameter containers; the arithmetic of each encoder stage runs in ONE autograd
function (`_MitStageFn`: patch embedding, the stage's blocks, the stage norm) that walks the blocks and calls the HIP kernels:

  * storage: the residual stream is fp32, every branch tensor (LayerNorm outputs, q / kv, attention output, Mix-FFN
    hidden tensors, gathered patch rows) fp16; master weights fp32 with fp16 copies made per forward;
  * arithmetic: fp16 MFMA with fp32 accumulation (Linear layers, patch-embedding / spatial-reduction convs as GEMMs over
    gathered rows, flash-style attention), fp32 LayerNorm / softmax / GELU;
  * backward: hand-written (no autograd graph inside): gradients travel in fp16 under a static loss scale
    (`loss_scale`, removed again when the fp32 parameter gradients are written), the residual-stream gradient in fp32.

Not mirrored: checkpoint loading through mmcv (`init_weights`), timm's registry decorators -- neither is arithmetic.
DropPath (timm `DropPath`, stochastic depth per sample, :156) is implemented (per-image scale in the branch GEMM's
epilogue); `drop` / `attn_drop` are 0 in every mit_b* configuration and must stay 0.
"""
import math
import os
import sys
from functools import partial

import torch
import torch.nn as nn

_pkg = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.path.dirname(_pkg) not in sys.path:
    sys.path.append(os.path.dirname(_pkg))
from diga_amd import _lib, config  # noqa: E402

P = _lib.ptr
_TAGS = ("mit_gemm", "mit_wgrad", "mit_attn_fwd", "mit_attn_bwd", "mit_norm", "mit_dwconv", "mit_misc")


def _trunc_normal_(t, std=0.02):
    return nn.init.trunc_normal_(t, std=std)


class DropPath(nn.Module):
    """Parameter-free marker module (timm.models.layers.DropPath): the rate is read by the encoder's forward."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = drop_prob


class DWConv(nn.Module):
    def __init__(self, dim=768):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, 3, 1, 1, bias=True, groups=dim)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        if drop != 0.:
            raise NotImplementedError("Mlp: dropout is 0 in every mit_b* configuration; not built")
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.dwconv = DWConv(hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0., sr_ratio=1):
        super().__init__()
        assert dim % num_heads == 0, f"dim {dim} should be divided by num_heads {num_heads}."
        if attn_drop != 0. or proj_drop != 0.:
            raise NotImplementedError("Attention: dropout is 0 in every mit_b* configuration; not built")
        self.dim = dim
        self.num_heads = num_heads
        head_dim = dim // num_heads
        if head_dim != 64:
            raise NotImplementedError(f"the attention kernel is built for head_dim 64 (every stage of mit_b1..b5); got {head_dim}")
        self.scale = qk_scale or head_dim ** -0.5
        self.q = nn.Linear(dim, dim, bias=qkv_bias)
        self.kv = nn.Linear(dim, dim * 2, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.sr_ratio = sr_ratio
        if sr_ratio > 1:
            self.sr = nn.Conv2d(dim, dim, kernel_size=sr_ratio, stride=sr_ratio)
            self.norm = nn.LayerNorm(dim)


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0., drop_path=0.,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm, sr_ratio=1):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop,
                              proj_drop=drop, sr_ratio=sr_ratio)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)


class OverlapPatchEmbed(nn.Module):
    """ Image to Patch Embedding """

    def __init__(self, img_size=224, patch_size=7, stride=4, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.H, self.W = img_size // patch_size, img_size // patch_size
        self.num_patches = self.H * self.W
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=stride, padding=(patch_size // 2, patch_size // 2))
        self.norm = nn.LayerNorm(embed_dim)


def _rup(v, q):
    return (v + q - 1) // q * q


class _Ops:
    """Thin launch helpers over include/diga_mit.h (raw pointers of torch tensors on the current stream)."""

    def __init__(self, device, side_ok=True):
        self.dev = device
        self.side_ok = side_ok       # False: every weight gradient of this pass stays on the current stream (see _MitStageFn.backward)

    def empty(self, shape, dtype=torch.float16):
        return torch.empty(shape, dtype=dtype, device=self.dev)

    def gemm(self, a, w16, bias, n, out_f32=False, residual=None, seg=None, rows_per_seg=0, out=None, accumulate=False, alpha=1.0):
        m, k = a.shape
        if out is None:
            out = self.empty((m, n), torch.float32 if out_f32 else torch.float16)
        _lib.call("diga_mit_gemm_nt", P(a), a.stride(0), P(w16), w16.stride(0), P(bias), P(out), out.stride(0), 1 if out_f32 else 0,
                  P(residual), 0 if residual is None else residual.stride(0), P(seg), int(rows_per_seg), 1 if accumulate else 0,
                  float(alpha), m, n, k, _lib.stream())
        return out

    def wgrad(self, dy, x, scale, bias=False, side_ok=True):
        """dW [n,k] = scale * dy^T x (fp32); with bias=True also the bias gradient (column sums of dy) from the same pass."""
        m, n = dy.shape
        k = x.shape[1]
        # Weight gradients are leaves of the stage's backward: under the step driver (`_lib.side_overlap`, as the DeepLab convolutions'
        # weight gradients) they run on the side stream next to the backward-data / attention chain -- in a captured HIP graph that is a
        # forked branch per weight gradient, joined before the optimizer (round 5: the MiT kernels are short and fill a fraction of the
        # chip each).  dy and x are never written again after they were produced (every in-place accumulation of the backward pass
        # goes into the residual-stream buffers, which no weight gradient reads).
        # side_ok=False: the caller post-processes the result on the CURRENT stream right away (a slice + reshape copy, the overflow check)
        side = _lib.side_stream(self.dev) if (dy.is_cuda and side_ok and self.side_ok) else None
        if side is not None:
            side.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(side):
                res = self._wgrad_launch(dy, x, scale, bias, m, n, k)
            for t in (dy, x):                   # (small tensors, hundreds of launches: the allocator's deferred reuse costs nothing here and
                t.record_stream(side)           #  _lib.release_to_side's event per launch measured +2 ms of host time on the eager step)
            return res
        return self._wgrad_launch(dy, x, scale, bias, m, n, k)

    def _wgrad_launch(self, dy, x, scale, bias, m, n, k):
        dw = self.empty((n, k), torch.float32)
        db = self.empty((n,), torch.float32) if bias else None
        nbytes = _lib.lib.diga_mit_gemm_tn_workspace_bytes(m, n, k)
        ws = _lib.workspace(nbytes, self.dev, "mit_wgrad")
        _lib.call("diga_mit_gemm_tn", P(dy), dy.stride(0), P(x), x.stride(0), P(dw), P(db), float(scale), 0, P(ws), ws.numel(), m, n, k,
                  _lib.stream())
        return (dw, db) if bias else dw

    def colsum(self, x, scale):
        m, c = x.shape
        out = self.empty((c,), torch.float32)
        ws = _lib.workspace(_lib.lib.diga_mit_colsum_workspace_bytes(m, c), self.dev, "mit_colsum")
        _lib.call("diga_mit_colsum", P(x), x.stride(0), P(out), float(scale), 0, P(ws), ws.numel(), m, c, _lib.stream())
        return out

    def ln_fwd(self, x32, gamma, beta, eps, want16=True, want32=False, save=True):
        m, c = x32.shape
        y16 = self.empty((m, c)) if want16 else None
        y32 = self.empty((m, c), torch.float32) if want32 else None
        mean = self.empty((m,), torch.float32) if save else None
        rstd = self.empty((m,), torch.float32) if save else None
        _lib.call("diga_mit_layernorm_fwd", P(x32), x32.stride(0), P(gamma), P(beta), P(y16), P(y32), c, P(mean), P(rstd), m, c,
                  float(eps), _lib.stream())
        return y16, y32, mean, rstd

    def ln_bwd(self, dy, x32, gamma, mean, rstd, dres, want32, want16, pscale, gscale=1.0):
        m, c = x32.shape
        dx32 = self.empty((m, c), torch.float32) if want32 else None
        dx16 = self.empty((m, c)) if want16 else None
        dg = self.empty((c,), torch.float32)
        db = self.empty((c,), torch.float32)
        ws = _lib.workspace(_lib.lib.diga_mit_layernorm_bwd_workspace_bytes(m, c), self.dev, "mit_ln")
        _lib.call("diga_mit_layernorm_bwd", P(dy), 1 if dy.dtype == torch.float32 else 0, dy.stride(0), float(gscale), P(x32),
                  x32.stride(0), P(gamma), P(mean), P(rstd), P(dres), 0 if dres is None else dres.stride(0), P(dx32), P(dx16), c,
                  P(dg), P(db), float(pscale), 0, P(ws), ws.numel(), m, c, _lib.stream())
        return dx32, dx16, dg, db

    def im2col(self, src, kind, b, h, w, c, k, stride, pad, kp):
        ho = (h + 2 * pad - k) // stride + 1
        wo = (w + 2 * pad - k) // stride + 1
        cols = self.empty((b * ho * wo, kp))
        _lib.call("diga_mit_im2col", P(src), kind, P(cols), b, h, w, c, k, k, stride, pad, ho, wo, kp, _lib.stream())
        return cols, ho, wo

    def col2im(self, dcols, dst, dst_f32, b, h, w, c, k, stride, pad, ho, wo, kp, gscale=1.0):
        _lib.call("diga_mit_col2im", P(dcols), P(dst), 1 if dst_f32 else 0, float(gscale), b, h, w, c, k, k, stride, pad, ho, wo, kp,
                  _lib.stream())

    def row_scale(self, x16, seg, rows_per_seg):
        if seg is None:
            return x16
        m, c = x16.shape
        y = self.empty((m, c))
        _lib.call("diga_mit_row_scale", P(x16), P(y), P(seg), int(rows_per_seg), m, c, _lib.stream())
        return y


def _lin16(w, want_t):
    """fp32 [N,K] weight -> (fp16 [N,K], fp16 [K,N] or None)."""
    n, k = w.shape
    w16 = torch.empty((n, k), dtype=torch.float16, device=w.device)
    wt = torch.empty((k, n), dtype=torch.float16, device=w.device) if want_t else None
    _lib.call("diga_mit_cast_transpose", P(w), P(w16), P(wt), n, k, _lib.stream())
    return w16, wt


def _conv16(w, want_t):
    """conv weight [Cout,Cin,R,S] -> rows [(ky*S+kx)*Cin + c] padded to a multiple of 32: (fp16 [Cout,Kp], fp16 [Kp,Cout], Kp)."""
    co, ci, r, s = w.shape
    kk = r * s * ci
    kp = _rup(kk, 32)
    flat = w.detach().permute(0, 2, 3, 1).reshape(co, kk)
    if kp != kk:
        flat = torch.nn.functional.pad(flat, (0, kp - kk))
    flat = flat.contiguous()
    w16, wt = _lin16(flat, want_t)
    return w16, wt, kp


class _MitStageFn(torch.autograd.Function):
    """ONE encoder stage (patch embedding, its blocks, the stage norm): forward(src, cfg, si, *params) -> the stage output
    c_{si+1} (NCHW-shaped view of the [B*h*w][C] fp32 matrix).  `src` is the image (stage 0) or the previous stage's output;
    `params` are this stage's parameters in `cfg['stage_names'][si]` order (so autograd hands their gradients back -- one
    stage at a time: with data parallelism the gradient buckets of the last stages leave while the first stages still run
    their backward).  Gradients cross the stage boundaries as TRUE gradients in fp32; the loss scale lives inside a stage."""

    @staticmethod
    def forward(ctx, src_t, cfg, si, *params):
        _lib.require_gpu(src_t)
        dev = src_t.device
        ops = _Ops(dev)
        names = cfg["stage_names"][si]
        par = dict(zip(names, params))
        need_grad = cfg["need_grad"]                          # (grad mode is always off inside Function.forward: decided by the caller)
        eps = cfg["eps"]                                       # per LayerNorm module (the reference mixes 1e-6 and torch's 1e-5)
        prep = cfg["prep"]
        if prep is None:
            raise RuntimeError("diga_amd ops run on the GPU only (HIP kernels, no CPU fallback)")
        B = src_t.shape[0]
        if si == 0:
            src = src_t.detach().float().contiguous()          # the image, NCHW
            src_kind, sh, sw, sc = 2, src.shape[2], src.shape[3], src.shape[1]
        else:
            src = src_t.detach().permute(0, 2, 3, 1)           # previous stage's output: [B][h][w][C] fp32
            if not (src.is_contiguous() and src.dtype == torch.float32):
                src = src.float().contiguous()
            src_kind, sh, sw, sc = 0, src.shape[1], src.shape[2], src.shape[3]
        st = cfg["stages"][si]
        if True:                                               # (one stage; the body keeps the indentation of the former stage loop)
            pre = st["embed"]
            C, heads, sr, hid = st["dim"], st["heads"], st["sr"], st["hidden"]
            w16, wt16, kp = prep[pre + ".proj.weight"]
            cols, ho, wo = ops.im2col(src, src_kind, B, sh, sw, sc, st["patch"], st["stride"], st["patch"] // 2, kp)
            M = B * ho * wo
            y32 = ops.gemm(cols, w16, par[pre + ".proj.bias"], C, out_f32=True)
            _, xcur, e_mean, e_rstd = ops.ln_fwd(y32, par[pre + ".norm.weight"], par[pre + ".norm.bias"], eps[pre + ".norm"], want16=False,
                                                 want32=True, save=need_grad)
            ssave = {"cols": cols if need_grad else None, "y32": y32 if need_grad else None, "e_mean": e_mean, "e_rstd": e_rstd,
                     "wt16": wt16, "kp": kp, "geom": (sh, sw, sc, ho, wo), "blocks": []}
            if not need_grad:
                del cols, y32
            for bi, bp in enumerate(st["blocks"]):
                seg_a = seg_m = None
                if cfg["training"] and st["drop_path"][bi] > 0.0:
                    # timm DropPath (:156,176-177): per sample keep / drop, rescaled by 1 / keep; one draw per branch -- all blocks' draws
                    # of a forward pass come from ONE rand call (forward_features: four launches per pass instead of four per block)
                    seg_a, seg_m = cfg["drop_scales"][si][bi].unbind(0)
                wq, wqt, _ = prep[bp + ".attn.q.weight"]
                wkv, wkvt, _ = prep[bp + ".attn.kv.weight"]
                wpr, wprt, _ = prep[bp + ".attn.proj.weight"]
                w1, w1t, _ = prep[bp + ".mlp.fc1.weight"]
                w2, w2t, _ = prep[bp + ".mlp.fc2.weight"]
                wdw, wdw_flip, _ = prep[bp + ".mlp.dwconv.dwconv.weight"]                                  # [9][hid] fp32, taps reversed
                a16, _, m1, r1 = ops.ln_fwd(xcur, par[bp + ".norm1.weight"], par[bp + ".norm1.bias"], eps[bp + ".norm1"], save=need_grad)
                q16 = ops.gemm(a16, wq, par.get(bp + ".attn.q.bias"), C)
                bs = {}
                if sr > 1:
                    wsr, wsrt, kps = prep[bp + ".attn.sr.weight"]
                    pc, hk, wk = ops.im2col(a16, 1, B, ho, wo, C, sr, sr, 0, kps)
                    s32 = ops.gemm(pc, wsr, par[bp + ".attn.sr.bias"], C, out_f32=True)
                    r16, _, ms, rs = ops.ln_fwd(s32, par[bp + ".attn.norm.weight"], par[bp + ".attn.norm.bias"], eps[bp + ".attn.norm"],
                                                  save=need_grad)
                    kv16 = ops.gemm(r16, wkv, par.get(bp + ".attn.kv.bias"), 2 * C)
                    nk = hk * wk
                    if need_grad:
                        bs.update(pc=pc, s32=s32, ms=ms, rs=rs, r16=r16, wsrt=wsrt, kps=kps, hk=hk, wk=wk)
                else:
                    kv16 = ops.gemm(a16, wkv, par.get(bp + ".attn.kv.bias"), 2 * C)
                    nk = ho * wo
                o16 = ops.empty((M, C))
                lse = ops.empty((B, heads, ho * wo), torch.float32) if need_grad else None
                _lib.call("diga_mit_attention_fwd", P(q16), C, P(kv16), 2 * C, P(o16), C, P(lse), B, heads, ho * wo, nk,
                          float(st["scale"]), _lib.stream())
                x1 = ops.gemm(o16, wpr, par[bp + ".attn.proj.bias"], C, out_f32=True, residual=xcur, seg=seg_a, rows_per_seg=ho * wo)
                b16, _, m2, r2 = ops.ln_fwd(x1, par[bp + ".norm2.weight"], par[bp + ".norm2.bias"], eps[bp + ".norm2"], save=need_grad)
                h1 = ops.gemm(b16, w1, par[bp + ".mlp.fc1.bias"], hid)
                u16 = ops.empty((M, hid)) if need_grad else None
                h2 = ops.empty((M, hid))
                _lib.call("diga_mit_dwconv_gelu_fwd", P(h1), P(wdw), P(par[bp + ".mlp.dwconv.dwconv.bias"]), P(u16), P(h2), B, ho, wo,
                          hid, _lib.stream())
                x2 = ops.gemm(h2, w2, par[bp + ".mlp.fc2.bias"], C, out_f32=True, residual=x1, seg=seg_m, rows_per_seg=ho * wo)
                if need_grad:
                    bs.update(x=xcur, m1=m1, r1=r1, a16=a16, q16=q16, kv16=kv16, nk=nk, o16=o16, lse=lse, x1=x1, m2=m2, r2=r2,
                              b16=b16, h1=h1, u16=u16, h2=h2, seg_a=seg_a, seg_m=seg_m, wqt=wqt, wkvt=wkvt, wprt=wprt, w1t=w1t, w2t=w2t,
                              wdw_flip=wdw_flip)
                    ssave["blocks"].append(bs)
                xcur = x2
            _, out32, n_mean, n_rstd = ops.ln_fwd(xcur, par[st["norm"] + ".weight"], par[st["norm"] + ".bias"], eps[st["norm"]], want16=False,
                                                  want32=True, save=need_grad)
            ssave.update(xlast=xcur if need_grad else None, n_mean=n_mean, n_rstd=n_rstd)
        ctx.cfg, ctx.si, ctx.saved, ctx.par = cfg, si, ssave if need_grad else None, par if need_grad else None
        ctx.B = B
        ctx.loss_scale = float(cfg["loss_scale"])
        ctx.src_needs_grad = bool(si > 0 and need_grad)
        return out32.view(B, ho, wo, C).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        cfg, si, ss, par = ctx.cfg, ctx.si, ctx.saved, ctx.par
        if ss is None:
            raise RuntimeError("MixVisionTransformer: backward without saved activations")
        ctx.saved = None                                       # activations are released as the pass walks down
        S = ctx.loss_scale
        inv = 1.0 / S
        B = ctx.B
        dev = g.device
        # A stage that entered the graph more than once (the one-backward self-training step runs the student twice) has its
        # parameters' gradients summed by autograd on the CURRENT stream: only the first contribution of a backward pass may still be
        # in flight on the side stream when it is handed over; the later ones join the side stream first and run in line
        # (same rule as DigaConv2d's `uses`, diga_amd/model/conv.py).
        uses = cfg.get("uses")
        later = False
        if uses is not None:
            later = uses[si][0] > 0
            uses[si][0] += 1
        if later:
            _lib.join_side()
        ops = _Ops(dev, side_ok=not later)
        grads = {}
        down = None                                            # true (unscaled) fp32 gradient wrt this stage's input
        if True:
            st = cfg["stages"][si]
            C, heads, sr, hid = st["dim"], st["heads"], st["sr"], st["hidden"]
            sh, sw, sc, ho, wo = ss["geom"]
            M = B * ho * wo
            g = g.permute(0, 2, 3, 1).reshape(M, C)
            g = g.float().contiguous() if not (g.dtype == torch.float32 and g.is_contiguous()) else g
            # the incoming gradient is the true one; inside the stage everything carries the loss scale S
            dx32, dx16, dg, db = ops.ln_bwd(g, ss["xlast"], par[st["norm"] + ".weight"], ss["n_mean"], ss["n_rstd"], None, True, True,
                                            inv, S)
            grads[st["norm"] + ".weight"], grads[st["norm"] + ".bias"] = dg, db
            ss["xlast"] = None
            for bi in range(len(st["blocks"]) - 1, -1, -1):
                bp, bs = st["blocks"][bi], ss["blocks"][bi]
                ss["blocks"][bi] = None
                rps = ho * wo
                # ---- Mix-FFN branch: x2 = x1 + seg * fc2(gelu(dw(fc1(norm2(x1)))))
                dxs = ops.row_scale(dx16, bs["seg_m"], rps)
                d_h2 = ops.gemm(dxs, bs["w2t"], None, hid)
                grads[bp + ".mlp.fc2.weight"], grads[bp + ".mlp.fc2.bias"] = ops.wgrad(dxs, bs["h2"], inv, bias=True)
                du = ops.empty((M, hid))
                dwd = ops.empty((hid, 9), torch.float32)
                dbd = ops.empty((hid,), torch.float32)
                ws = _lib.workspace(_lib.lib.diga_mit_dwconv_bwd_workspace_bytes(B, ho, hid), dev, "mit_dw")
                _lib.call("diga_mit_dwconv_gelu_bwd", P(d_h2), P(bs["u16"]), P(bs["h1"]), P(bs["wdw_flip"]), P(du), P(d_h2), P(dwd), P(dbd), inv, 0,
                          P(ws), ws.numel(), B, ho, wo, hid, _lib.stream())
                d_h1 = d_h2
                grads[bp + ".mlp.dwconv.dwconv.weight"] = dwd.view(hid, 1, 3, 3)
                grads[bp + ".mlp.dwconv.dwconv.bias"] = dbd
                d_b16 = ops.gemm(d_h1, bs["w1t"], None, C)
                grads[bp + ".mlp.fc1.weight"], grads[bp + ".mlp.fc1.bias"] = ops.wgrad(d_h1, bs["b16"], inv, bias=True)
                dx32, dx16, dg, db = ops.ln_bwd(d_b16, bs["x1"], par[bp + ".norm2.weight"], bs["m2"], bs["r2"], dx32, True, True, inv)
                grads[bp + ".norm2.weight"], grads[bp + ".norm2.bias"] = dg, db
                del d_h2, d_h1, du, d_b16
                # ---- attention branch: x1 = x + seg * proj(attn(norm1(x)))
                dxs = ops.row_scale(dx16, bs["seg_a"], rps)
                d_o = ops.gemm(dxs, bs["wprt"], None, C)
                grads[bp + ".attn.proj.weight"], grads[bp + ".attn.proj.bias"] = ops.wgrad(dxs, bs["o16"], inv, bias=True)
                nk = bs["nk"]
                dq = ops.empty((M, C))
                dkv = ops.empty((B * nk, 2 * C))
                ws = _lib.workspace(_lib.lib.diga_mit_attention_bwd_workspace_bytes(B, heads, ho * wo, nk), dev, "mit_attn")
                _lib.call("diga_mit_attention_bwd", P(bs["q16"]), C, P(bs["kv16"]), 2 * C, P(bs["o16"]), P(d_o), C, P(bs["lse"]), P(dq),
                          P(dkv), P(ws), ws.numel(), B, heads, ho * wo, nk, float(st["scale"]), _lib.stream())
                d_a = ops.gemm(dq, bs["wqt"], None, C)
                has_qb, has_kvb = (bp + ".attn.q.bias") in par, (bp + ".attn.kv.bias") in par
                gq = ops.wgrad(dq, bs["a16"], inv, bias=has_qb)
                grads[bp + ".attn.q.weight"], grads[bp + ".attn.q.bias"] = gq if has_qb else (gq, None)
                if sr > 1:
                    d_r = ops.gemm(dkv, bs["wkvt"], None, C)
                    gkv = ops.wgrad(dkv, bs["r16"], inv, bias=has_kvb)
                    grads[bp + ".attn.kv.weight"], grads[bp + ".attn.kv.bias"] = gkv if has_kvb else (gkv, None)
                    _, d_s, dg, db = ops.ln_bwd(d_r, bs["s32"], par[bp + ".attn.norm.weight"], bs["ms"], bs["rs"], None, False, True, inv)
                    grads[bp + ".attn.norm.weight"], grads[bp + ".attn.norm.bias"] = dg, db
                    kps = bs["kps"]
                    d_pc = ops.gemm(d_s, bs["wsrt"], None, kps)
                    dwsr, grads[bp + ".attn.sr.bias"] = ops.wgrad(d_s, bs["pc"], inv, bias=True, side_ok=False)
                    grads[bp + ".attn.sr.weight"] = dwsr[:, :sr * sr * C].reshape(C, sr, sr, C).permute(0, 3, 1, 2)
                    ops.col2im(d_pc, d_a, False, B, ho, wo, C, sr, sr, 0, bs["hk"], bs["wk"], kps)
                else:
                    ops.gemm(dkv, bs["wkvt"], None, C, out=d_a, accumulate=True)
                    gkv = ops.wgrad(dkv, bs["a16"], inv, bias=has_kvb)
                    grads[bp + ".attn.kv.weight"], grads[bp + ".attn.kv.bias"] = gkv if has_kvb else (gkv, None)
                dx32, dx16, dg, db = ops.ln_bwd(d_a, bs["x"], par[bp + ".norm1.weight"], bs["m1"], bs["r1"], dx32, True, True, inv)
                grads[bp + ".norm1.weight"], grads[bp + ".norm1.bias"] = dg, db
                del bs
            # ---- overflow check: the branch gradients of this stage travelled in fp16 under the loss scale; an inf / NaN born in
            # any of them (dq, dkv, the Mix-FFN hidden gradients) has reached the fp32 residual-stream gradient by now.  One
            # bandwidth pass over it sets the model's device flag: DigaSGD skips the step on it (no host sync), the host lowers
            # the scale at its leisure (MixVisionTransformer.adjust_loss_scale).  The reference is fp32 and cannot overflow.
            if cfg.get("overflow_flag") is not None:
                _lib.call("diga_nonfinite_flag_f32", P(dx32), dx32.numel(), P(cfg["overflow_flag"]), _lib.stream())
            # ---- patch embedding: x0 = norm(proj(cols))
            pre = st["embed"]
            _, d_y, dg, db = ops.ln_bwd(dx32, ss["y32"], par[pre + ".norm.weight"], ss["e_mean"], ss["e_rstd"], None, False, True, inv)
            grads[pre + ".norm.weight"], grads[pre + ".norm.bias"] = dg, db
            k = st["patch"]
            dwp, grads[pre + ".proj.bias"] = ops.wgrad(d_y, ss["cols"], inv, bias=True, side_ok=False)
            if cfg.get("overflow_flag") is not None:
                # (d_y is fp16 under the loss scale as well: an overflow born in the patch embedding's LayerNorm backward shows in
                #  its projection's weight gradient -- a few thousand fp32 values; covers stage 1, which no later check sees)
                _lib.call("diga_nonfinite_flag_f32", P(dwp), dwp.numel(), P(cfg["overflow_flag"]), _lib.stream())
            grads[pre + ".proj.weight"] = dwp[:, :k * k * sc].reshape(C, k, k, sc).permute(0, 3, 1, 2)
            if ctx.src_needs_grad and ctx.needs_input_grad[0]:
                d_cols = ops.gemm(d_y, ss["wt16"], None, ss["kp"])
                down = ops.empty((B, sh, sw, sc), torch.float32)
                ops.col2im(d_cols, down, True, B, sh, sw, sc, k, st["stride"], k // 2, ho, wo, ss["kp"], gscale=inv)
                down = down.permute(0, 3, 1, 2)               # NCHW-shaped view, the layout the stage output has
        out = []
        for n in cfg["stage_names"][si]:
            gr = grads.get(n)
            if gr is not None and tuple(gr.shape) != tuple(par[n].shape):
                gr = gr.reshape(par[n].shape)
            out.append(gr if par[n].requires_grad else None)
        return (down, None, None, *out)


class MixVisionTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dims=[64, 128, 256, 512],
                 num_heads=[1, 2, 4, 8], mlp_ratios=[4, 4, 4, 4], qkv_bias=False, qk_scale=None, drop_rate=0.,
                 attn_drop_rate=0., drop_path_rate=0., norm_layer=nn.LayerNorm, depths=[3, 4, 6, 3], sr_ratios=[8, 4, 2, 1]):
        super().__init__()
        self.num_classes = num_classes
        self.depths = depths
        self.embed_dims = list(embed_dims)
        self.loss_scale = float(config.active().mit_loss_scale)
        # [0]: this step's backward met an inf / NaN (set on the device, cleared by the next training forward); [1]: such steps so far
        self.register_buffer("grad_overflow", torch.zeros(2, dtype=torch.int32), persistent=False)
        self._bw_seen = [[0], [0], [0], [0]]
        self._overflow_seen = 0
        self.patch_embed1 = OverlapPatchEmbed(img_size=img_size, patch_size=7, stride=4, in_chans=in_chans, embed_dim=embed_dims[0])
        self.patch_embed2 = OverlapPatchEmbed(img_size=img_size // 4, patch_size=3, stride=2, in_chans=embed_dims[0], embed_dim=embed_dims[1])
        self.patch_embed3 = OverlapPatchEmbed(img_size=img_size // 8, patch_size=3, stride=2, in_chans=embed_dims[1], embed_dim=embed_dims[2])
        self.patch_embed4 = OverlapPatchEmbed(img_size=img_size // 16, patch_size=3, stride=2, in_chans=embed_dims[2], embed_dim=embed_dims[3])
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        cur = 0
        for s in range(4):
            blocks = nn.ModuleList([Block(dim=embed_dims[s], num_heads=num_heads[s], mlp_ratio=mlp_ratios[s], qkv_bias=qkv_bias,
                                          qk_scale=qk_scale, drop=drop_rate, attn_drop=attn_drop_rate, drop_path=dpr[cur + i],
                                          norm_layer=norm_layer, sr_ratio=sr_ratios[s]) for i in range(depths[s])])
            setattr(self, f"block{s + 1}", blocks)
            setattr(self, f"norm{s + 1}", norm_layer(embed_dims[s]))
            cur += depths[s]
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            _trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)
        elif isinstance(m, nn.Conv2d):
            fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
            fan_out //= m.groups
            m.weight.data.normal_(0, math.sqrt(2.0 / fan_out))
            if m.bias is not None:
                m.bias.data.zero_()

    def init_weights(self, pretrained=None):
        if isinstance(pretrained, str):
            sd = torch.load(pretrained, map_location="cpu")
            self.load_state_dict(sd.get("state_dict", sd), strict=False)

    def reset_drop_path(self, drop_path_rate):
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(self.depths))]
        cur = 0
        for s in range(4):
            for i, blk in enumerate(getattr(self, f"block{s + 1}")):
                if dpr[cur + i] > 0 and not isinstance(blk.drop_path, DropPath):
                    blk.drop_path = DropPath(dpr[cur + i])
                if isinstance(blk.drop_path, DropPath):
                    blk.drop_path.drop_prob = dpr[cur + i]
            cur += self.depths[s]

    def freeze_patch_emb(self):
        self.patch_embed1.requires_grad = False

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'pos_embed1', 'pos_embed2', 'pos_embed3', 'pos_embed4', 'cls_token'}

    def _prepare_weights(self):
        """fp16 GEMM operands of every Linear / conv weight ([Co][Kp] and its transpose [Kp][Co]) and the fp32 tap-major forms of
        the depthwise weights, refreshed from the fp32 master parameters by ONE launch per forward (diga_mit_weight_prep_multi;
        the parameters change in place at every SGD / EMA step, so the copies cannot be cached across steps).  The buffers are
        persistent per model: a backward pass reads the operands of ITS forward, which is the same memory -- valid as long as
        the parameters are not modified between a forward and its backward (true for every optimizer: they step after)."""
        import ctypes
        named = dict(self.named_parameters())
        sig = (next(iter(named.values())).device,) + tuple(p.data_ptr() for p in named.values())
        st = getattr(self, "_prep_state", None)
        if st is None or st["sig"] != sig:
            dev = sig[0]
            entries, out, starts, total = [], {}, [], 0
            for n, p in named.items():
                if p.dim() == 2:                                           # nn.Linear [Co, Ci]
                    co, ci, rs, mode = p.shape[0], p.shape[1], 1, 0
                elif p.dim() == 4 and p.shape[1] == 1 and n.endswith("dwconv.dwconv.weight"):
                    co, ci, rs, mode = p.shape[0], 1, p.shape[2] * p.shape[3], 1
                elif p.dim() == 4:                                         # patch-embedding / spatial-reduction conv [Co, Ci, R, S]
                    co, ci, rs, mode = p.shape[0], p.shape[1], p.shape[2] * p.shape[3], 0
                else:
                    continue
                if not (p.is_contiguous() and p.dtype == torch.float32):
                    raise RuntimeError(f"MixVisionTransformer: parameter {n} must be a contiguous fp32 tensor")
                kp = _rup(rs * ci, 32) if mode == 0 else rs
                if mode == 0:
                    a = torch.empty((co, kp), dtype=torch.float16, device=dev)
                    b = torch.empty((kp, co), dtype=torch.float16, device=dev)
                else:
                    a = torch.empty((rs, co), dtype=torch.float32, device=dev)
                    b = torch.empty((rs, co), dtype=torch.float32, device=dev)
                tiles_k = (kp + 31) // 32
                entries.append(_lib.MitWeightPrep(p.data_ptr(), a.data_ptr(), b.data_ptr(), co, ci, rs, kp, mode, tiles_k))
                starts.append(total)
                total += ((co + 31) // 32) * tiles_k
                out[n] = (a, b, kp)
            arr = (_lib.MitWeightPrep * len(entries))(*entries)
            tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
            ts = torch.tensor(starts, dtype=torch.int64).to(dev)
            st = {"sig": sig, "tab": tab, "starts": ts, "n": len(entries), "total": total, "out": out}
            object.__setattr__(self, "_prep_state", st)
        _lib.call("diga_mit_weight_prep_multi", P(st["tab"]), P(st["starts"]), st["n"], st["total"], _lib.stream())
        return st["out"]

    def _cfg(self):
        names = [n for n, _ in self.named_parameters()]
        stages = []
        eps = {n: m.eps for n, m in self.named_modules() if isinstance(m, nn.LayerNorm)}
        for s in range(4):
            pe = getattr(self, f"patch_embed{s + 1}")
            blocks = getattr(self, f"block{s + 1}")
            b0 = blocks[0]
            stages.append({"embed": f"patch_embed{s + 1}", "norm": f"norm{s + 1}", "dim": pe.proj.out_channels,
                           "patch": pe.proj.kernel_size[0], "stride": pe.proj.stride[0], "heads": b0.attn.num_heads,
                           "sr": b0.attn.sr_ratio, "scale": b0.attn.scale, "hidden": b0.mlp.fc1.out_features,
                           "blocks": [f"block{s + 1}.{i}" for i in range(len(blocks))],
                           "drop_path": [float(getattr(b.drop_path, "drop_prob", 0.0)) for b in blocks]})
        stage_names = [[n for n in names if n.startswith((f"patch_embed{s + 1}.", f"block{s + 1}.", f"norm{s + 1}."))] for s in range(4)]
        assert sum(len(v) for v in stage_names) == len(names)
        return {"names": names, "stage_names": stage_names, "stages": stages, "eps": eps, "training": self.training,
                "loss_scale": self.loss_scale}

    def adjust_loss_scale(self, factor=0.5, floor=1.0, steps=200, growth=2.0, growth_interval=1000):
        """Host side of the overflow handling, a standard dynamic loss scaler at window granularity (one device->host read: call it
        every `steps` steps, not every step).  Backward passes overflowed since the last call -> the scale is multiplied by `factor`
        (not below `floor`); `growth_interval` consecutive clean steps -> multiplied by `growth`, capped at the scale the model was
        built with (a transient spike early in an 80 000-iteration run must not leave the fp16 branch gradients at a reduced scale
        for the rest of training: they would underflow silently).  Returns the number of overflowed (= skipped) steps since the
        last call.  A trainer that replays a captured HIP graph must re-capture whenever `loss_scale` changed (the scale is a kernel
        argument): DigaTrainer compares it before and after."""
        total = int(self.grad_overflow[1].item())
        new = total - self._overflow_seen
        self._overflow_seen = total
        if not hasattr(self, "_scale_cap"):
            self._scale_cap, self._clean_steps = float(self.loss_scale), 0
        if new > 0:
            self.loss_scale = max(floor, self.loss_scale * factor)
            self._clean_steps = 0
        else:
            self._clean_steps += int(steps)
            if self._clean_steps >= growth_interval and self.loss_scale < self._scale_cap:
                self.loss_scale = min(self._scale_cap, self.loss_scale * growth)
                self._clean_steps = 0
        return new

    def _drop_path_scales(self, cfg, B, dev):
        """Per stage a [blocks][2][B] tensor of DropPath scales (0 or 1 / keep; branch 0 = attention, 1 = Mix-FFN) for this forward pass,
        or None outside training / without DropPath: one uniform draw for the whole encoder against the per-block keep probabilities
        (a cached device vector) -- the per-block form cost four tiny launches per block and network, ~400 per training step."""
        dps = [st["drop_path"] for st in cfg["stages"]]
        if not cfg["training"] or not any(d > 0.0 for dp in dps for d in dp):
            return [None] * len(dps)
        flat = [1.0 - d for dp in dps for d in dp]
        key = (str(dev), tuple(flat))
        cached = getattr(self, "_keep_cache", None)
        if cached is None or cached[0] != key:
            cached = (key, torch.tensor(flat, dtype=torch.float32).view(-1, 1, 1).to(dev))
            object.__setattr__(self, "_keep_cache", cached)
        keep = cached[1]
        scales = (torch.rand((len(flat), 2, B), device=dev) < keep).float() / keep
        out, at = [], 0
        for dp in dps:
            out.append(scales[at:at + len(dp)])
            at += len(dp)
        return out

    def forward_features(self, x):
        params = [p for _, p in self.named_parameters()]
        cfg = self._cfg()
        cfg["need_grad"] = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
        cfg["overflow_flag"] = None
        if cfg["need_grad"] and x.is_cuda:
            # The flag means "a backward pass overflowed since the optimizer last looked": it is cleared by the first training
            # forward AFTER an optimizer step has consumed it (DigaSGD.step(found_inf=flag) -> _lib.flag_consumed), never by a
            # forward that runs between a backward pass and that step -- the overlapped self-training step runs the forward of
            # student(cross_mix) next to the backward of student(cat), and a clear there would erase an overflow the step must skip on.
            if _lib.take_consumed_flag(self.grad_overflow):
                self.grad_overflow[0:1].zero_()
            cfg["overflow_flag"] = self.grad_overflow
            if any(u[0] > 0 for u in self._bw_seen):          # a backward pass has consumed the previous graph(s)
                for u in self._bw_seen:
                    u[0] = 0
            cfg["uses"] = self._bw_seen                        # per stage: [weight-gradient passes of the current backward pass]
        cfg["prep"] = self._prepare_weights() if x.is_cuda else None
        cfg["drop_scales"] = self._drop_path_scales(cfg, x.shape[0], x.device) if x.is_cuda else None
        named = dict(self.named_parameters())
        outs, src = [], x
        for si in range(4):
            src = _MitStageFn.apply(src, cfg, si, *[named[n] for n in cfg["stage_names"][si]])
            outs.append(src)
        return outs

    def forward(self, x):
        return self.forward_features(x)


def _mit(embed_dims, depths):
    class _M(MixVisionTransformer):
        def __init__(self, **kwargs):
            super().__init__(patch_size=4, embed_dims=embed_dims, num_heads=[1, 2, 5, 8], mlp_ratios=[4, 4, 4, 4], qkv_bias=True,
                             norm_layer=partial(nn.LayerNorm, eps=1e-6), depths=depths, sr_ratios=[8, 4, 2, 1], drop_rate=0.0,
                             drop_path_rate=0.1)
    return _M


class mit_b0(MixVisionTransformer):
    def __init__(self, **kwargs):
        raise NotImplementedError("mit_b0 has head_dim 32; the attention kernel is built for head_dim 64 (mit_b1..b5)")


mit_b1 = _mit([64, 128, 320, 512], [2, 2, 2, 2])
mit_b2 = _mit([64, 128, 320, 512], [3, 4, 6, 3])
mit_b3 = _mit([64, 128, 320, 512], [3, 4, 18, 3])
mit_b4 = _mit([64, 128, 320, 512], [3, 8, 27, 3])
mit_b5 = _mit([64, 128, 320, 512], [3, 6, 40, 3])
for _n, _c in (("mit_b1", mit_b1), ("mit_b2", mit_b2), ("mit_b3", mit_b3), ("mit_b4", mit_b4), ("mit_b5", mit_b5)):
    _c.__name__ = _c.__qualname__ = _n
