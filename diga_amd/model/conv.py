"""Conv2d on the MI355X fp32 matrix cores (diga_conv2d_nhwc_f32 and friends).

`DigaConv2d` is a drop-in for the nn.Conv2d layers of the reference model
(G5/model/seg_model_noaux.py:57-101,140-172): same constructor, same parameter names and [Cout,Cin,R,S]
shapes in the state_dict.  Inside, weights live in channels_last memory ([Cout][R][S][Cin]) and
activations travel NHWC; tensors handed to / returned from the module are NCHW-shaped views of that
memory (torch's channels_last format), so neighbouring torch ops see ordinary 4-D tensors.
Channel counts that are not multiples of 32 (the 3-channel image, the 19-class head) are zero-padded.
"""
import os
import sys

import torch
import torch.nn as nn

_pkg = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.path.dirname(_pkg) not in sys.path:
    sys.path.append(os.path.dirname(_pkg))
from diga_amd import _lib, config  # noqa: E402

_TAG_FWD, _TAG_BWD_DATA = _lib.PROF_TAGS.index("conv_fwd"), _lib.PROF_TAGS.index("conv_bwd_data")


def _pad_to(c, q=32):
    return (c + q - 1) // q * q


def _pad_last(t, c_to):
    """Zero-pad the last (channel) dimension of a contiguous tensor to c_to."""
    c = t.shape[-1]
    if c == c_to:
        return t
    out = t.new_zeros(t.shape[:-1] + (c_to,))
    out[..., :c] = t
    return out



def _bias_grad(gy):
    """Bias gradient of a convolution = column sums of the NHWC output gradient (torch: grad_output.sum((0, 2, 3))), on
    diga_colsum_nhwc.  gy: [N,H,W,K] (a channel slice of a contiguous NHWC tensor is fine)."""
    k = gy.shape[-1]
    ld = gy.stride(2)
    m = gy.shape[0] * gy.shape[1] * gy.shape[2]
    dense = gy.stride(3) == 1 and gy.stride(1) == gy.shape[2] * ld and gy.stride(0) == gy.shape[1] * gy.stride(1)
    if not dense or ((k % 4 != 0 or ld % 4 != 0) and k > 64):
        return gy.sum(dim=(0, 1, 2))                          # (layouts no model of this repository produces)
    out = torch.empty(k, dtype=torch.float32, device=gy.device)
    ws = _lib.workspace(_lib.lib.diga_norm_workspace_bytes(m, 1, k), gy.device, "norm")
    _lib.call("diga_colsum_nhwc", _lib.ptr(gy), ld, _lib.ptr(out), m, k, _lib.ptr(ws), ws.numel(), _lib.stream())
    return out


def _use_twin(cin, k, taps, shared):
    """The staging-free kernel (both operands pre-split, LDS-DMA) needs one extra pass over the activations to build
    their split twin (read 4 B + write 4 B per element).  That pays when the tensor is read by many tiles: convs with
    more than one tap, or several convs on one input (`shared`, the ASPP branches).  DIGA_CONV_TWIN=0 switches the
    path off, =1 forces it for every eligible conv."""
    mode = config.active().conv_twin
    if mode == "0" or cin % 32 != 0 or k <= 64:
        return False
    return mode == "1" or taps > 1 or shared


def takes_twin_only_input(conv, pointwise_ok=False):
    """True when `conv` (a DigaConv2d) reads its input exclusively through split twins -- forward on the twin kernel and
    backward-weight on the twin kernel -- so that its producer may write the twin instead of the fp32 tensor.
    pointwise_ok: also for 1x1 layers (worth it only when both of its twins are free, i.e. under autograd where the
    weight gradient gains 35-39 %; the forward kernel alone gains nothing on 8-step tiles)."""
    taps = conv.kernel_size[0] * conv.kernel_size[1]
    cfg = config.active()
    return (cfg.conv_math == 1 and cfg.conv_twin != "0" and cfg.twin_only
            and (taps > 1 or pointwise_ok) and conv.in_channels % 32 == 0 and conv.in_channels > 64 and conv.out_channels >= 256 and conv.out_channels % 8 == 0
            and tuple(conv.stride) == (1, 1) and conv.groups == 1)


INLINE_WGRAD = "inline"       # `uses` of a functional _Conv2dFn call whose weight is a non-leaf tensor (see backward)
_WINO_CACHE = {}
# bench.py sets this to a dict to learn what the convolutions of a step multiply: name -> [FLOPs of the direct
# convolution (the algorithmic work), FLOPs the matrix cores execute (16/36 of it per 2x2 tile on the Winograd path)]
flop_log = None


def _log_flops(name, direct, executed):
    if flop_log is not None:
        e = flop_log.setdefault(name, [0.0, 0.0])
        e[0] += direct
        e[1] += executed


# Path switches live in diga_amd/config.py (StepConfig: winograd, winograd_ratio, winograd_max_tile, winograd_keep_v, winograd_stats,
# keep_v_max_gb, keep_v_min_device_gb; environment variables give their DEFAULTS, read once at import); a convolution call reads the
# active configuration, makes no environment or driver query.  winograd_max_tile = 2 keeps every layer on F(2x2,3x3).
_DEVICE_TOTAL = {}
_KEEP_OFF = set()             # devices on which a keep-V allocation failed: the recompute path from then on (reset_keep_decisions())


def reset_keep_decisions():
    """Forget that a kept-transform allocation ran out of memory (e.g. after the caller freed a second model)."""
    _KEEP_OFF.clear()


def _room_for(nbytes, device):
    """Keeping a layer's transformed input V alive until its weight gradient is a memory-for-bandwidth trade (the backward-weight
    pass skips one input transform: a bandwidth pass over the input + V; 26 layers, ~4 ms of a 424 ms C2 step).  Decision:
    `winograd_keep_v` on (default), the layer's V <= `keep_v_max_gb` (8 GiB; the largest C2 layer keeps 2.2 GB) and the device
    has >= `keep_v_min_device_gb` of memory in total (160 GiB: an MI355X has 288 GB; the C2 step keeps ~21 GB of V and peaks at
    the figure bench.py reports as peak_mem_gb) -- shape and static device properties only, no free-memory query.  If the
    allocation itself fails, `_alloc_keep_v` switches the device to the recompute path."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    cfg = config.active()
    if not cfg.winograd_keep_v or idx in _KEEP_OFF or nbytes > int(cfg.keep_v_max_gb * (1 << 30)):
        return False
    total = _DEVICE_TOTAL.get(idx)
    if total is None:
        total = _DEVICE_TOTAL[idx] = torch.cuda.get_device_properties(idx).total_memory
    return total >= int(cfg.keep_v_min_device_gb * (1 << 30))


def _alloc_keep_v(nfloats, device):
    """The kept transform's buffer, or None when the policy says recompute or the allocation fails (out of memory: the device is
    switched to the recompute path instead of failing the step)."""
    if not _room_for(nfloats * 4, device):
        return None
    try:
        return torch.empty(nfloats, dtype=torch.float32, device=device)
    except torch.cuda.OutOfMemoryError:
        _KEEP_OFF.add(device.index if device.index is not None else torch.cuda.current_device())
        return None


def _wino_plan(hi, wi, d):
    """(tile, ratio): the Winograd output-tile edge for a stride-1 3x3 layer with dilation d on an hi x wi map and the share of the
    direct convolution's multiplications it executes.  The d*d sub-images {(a + d i, b + d j)} are cut into m x m tiles of
    (m + 2)^2 products each: F(2x2,3x3) 16 per 4 outputs, F(4x4,3x3) 36 per 16, F(6x6,3x3) 64 per 36; the smallest count wins, the
    smaller tile on a tie (97 x 97 map: dilation 1 / 2 / 4 / 18 -> 0.218 / 0.218 / 0.218 / 0.245 with 6x6 tiles, where 4x4 tiles
    give 0.266 / 0.266 / 0.266 / 0.551; dilation 12 / 24 -> 0.266 with 4x4 tiles, 0.435 with 6x6; direct = 1)."""
    max_tile = config.active().winograd_max_tile
    key = (hi, wi, d, max_tile)
    plan = _WINO_CACHE.get(key)
    if plan is None:
        def tiles(length, m):
            return sum((((length - a + d - 1) // d if length > a else 0) + m - 1) // m for a in range(d))
        plan = None
        for m in (2, 4, 6):
            if m > max_tile:
                continue
            ratio = float((m + 2) ** 2) * tiles(hi, m) * tiles(wi, m) / (9.0 * hi * wi)
            if plan is None or ratio < plan[1] - 1e-9:
                plan = (m, ratio)
        _WINO_CACHE[key] = plan
    return plan


_TILE_TABLES = {}


def _tile_table(n, hi, wi, d, tile, device):
    """The Winograd tile table of a geometry (a function of (N, H, W, dilation, tile) only: image and top-left output pixel of every
    tile), built once per device and reused by every call on that geometry -- forward, backward-data and backward-weight of all
    layers that share it (layer3's 23 conv2 layers: 4 calls each per step) -- instead of one wino_tiles_kernel launch per call (136
    per C2 step).  The one-time build is followed by a stream synchronisation: the table is read from several streams (student,
    teacher, weight-gradient side stream).  Not inside a stream capture (no sync there): the call then builds its own."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, n, hi, wi, d, tile)
    tab = _TILE_TABLES.get(key)
    if tab is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        tab = torch.empty(_lib.lib.diga_conv2d_winograd_tile_table_bytes(n, hi, wi, d, tile), dtype=torch.uint8, device=device)
        _lib.call("diga_conv2d_winograd_tile_table", _lib.ptr(tab), n, hi, wi, d, tile, _lib.stream())
        torch.cuda.current_stream(device).synchronize()
        # never evicted: a table is a few KB, its device pointer may be baked into captured HIP graphs and kernels on the side /
        # teacher streams may still be reading it (clearing the cache at 256 geometries freed tables in use)
        _TILE_TABLES[key] = tab
    return tab


def winograd_stats_plan(n, hi, wi, cin_padded, k, r, s, stride, padding, dilation, ho, wo):
    """(floats, records) of the statistics buffer a forward DigaConv2d on the fp32 Winograd path with 4x4 / 6x6 tiles fills for the
    BatchNorm behind it (diga_conv2d_winograd_stats_floats / _records), or None when the layer is not on that path."""
    if _lib.get_conv_math() != 0 or not _winograd_ok(n, hi, wi, cin_padded, k, r, s, stride, (-padding[0], -padding[1]), tuple(dilation), ho, wo):
        return None
    tile = _wino_plan(hi, wi, dilation[0])[0]
    if tile < 4 or k % 4 != 0:
        return None
    return (_lib.lib.diga_conv2d_winograd_stats_floats(n, hi, wi, k, dilation[0], tile),
            _lib.lib.diga_conv2d_winograd_stats_records(n, hi, wi, k, dilation[0], tile))


def _wino_ratio(hi, wi, d):
    return _wino_plan(hi, wi, d)[1]


def _winograd_ok(n, hi, wi, cin, k, r, s, stride, off0, doff, ho, wo):
    """Exact-fp32 mode: stride-1 'same' 3x3 layers (dilation d = padding, forward or backward-data geometry) go through
    Winograd (csrc/winograd.hip; tile from _wino_plan) when cutting the d*d sub-images into tiles leaves few enough
    multiplications: share of the direct convolution <= `winograd_ratio` (default 0.62; dilation 24 on a 97x97 map has 5x5
    sub-images -> 0.98 and stays direct, where the kernel also skips the dead taps).  `winograd = False` switches the path off."""
    cfg = config.active()
    if not cfg.winograd:
        return False
    d = abs(doff[0])
    if not (r == 3 and s == 3 and tuple(stride) == (1, 1) and doff[0] == doff[1] and d >= 1 and off0[0] == -doff[0]
            and off0[1] == -doff[1] and hi == ho and wi == wo and cin % 32 == 0 and cin >= 128 and k % 4 == 0 and k >= 128):
        return False
    ratio = _wino_ratio(hi, wi, d)
    # (the batched GEMM indexes its products * tiles rows as a [rows / 256][256] image with 15-bit row coordinates)
    return ratio <= cfg.winograd_ratio and n * hi * wi * ratio * 9.0 / 256.0 < 32000


def _conv_launch(x, w_krsc, bias, out, stride, off0, doff, tag, stats=None, twin_box=None, must_twin=False, epi=None,
                 opts=None, keep_v=None, wino_stats=False):
    """x [N,Hi,Wi,Cin] (contiguous or a channel slice of a contiguous tensor), w_krsc [K,R,S,Cin],
    out [N,Ho,Wo,K] (same rule).  twin_box: a one-element list shared by the convs that read the very same x.
    epi: a _lib.BwdEpilogue (backward-data only, bias-free): the `_epi` entry points finish the gradient in the epilogue.
    opts: (reflect_pad, upsample_shift, activation) = a diga_conv_options_t handed to the `_opts` entry points -- x is then the SOURCE tensor of the
    (virtually) upsampled / mirrored input."""
    import ctypes

    copt = None
    if opts is not None and any(opts):
        if stats is not None or epi is not None:
            raise RuntimeError("DigaConv2d: folded padding / upsampling / activation cannot be combined with BN statistics or a backward epilogue")
        copt = _lib.ConvOptions(int(opts[0]), int(opts[1]), int(opts[2]))
    n, hi, wi, cin = x.shape
    _, ho, wo, k = out.shape
    _, r, s, _ = w_krsc.shape
    if (_lib.get_conv_math() == 1 and _use_twin(cin, k, r * s, twin_box is not None)
            and n * hi * wi * cin * 4 < (1 << 40)):
        # split-bf16 arithmetic without register staging: both operands pre-split, copied global -> LDS by LDS-DMA
        twin = twin_box[0] if twin_box is not None else None
        if twin is None:
            xc = x if x.is_contiguous() else x.contiguous()
            twin = torch.empty(n * hi * wi * cin * 4, dtype=torch.uint8, device=x.device)
            _lib.call("diga_make_twin", _lib.ptr(xc), cin, _lib.ptr(twin), n * hi * wi, cin, _lib.stream())
            if twin_box is not None:
                twin_box[0] = twin
        img = torch.empty(_lib.lib.diga_split_bf16_image_bytes(k, r * s, cin), dtype=torch.uint8, device=x.device)
        _lib.call("diga_split_bf16_image", _lib.ptr(w_krsc), _lib.ptr(img), k, r * s, cin, _lib.stream())
        if epi is not None:
            _lib.call("diga_conv2d_nhwc_twin_epi", _lib.ptr(twin), _lib.ptr(img), _lib.ptr(out), n, hi, wi, cin, ho, wo, k,
                      out.stride(2), r, s, stride[0], stride[1], off0[0], off0[1], doff[0], doff[1], ctypes.byref(epi), tag,
                      _lib.stream())
            return twin
        if copt is not None:
            _lib.call("diga_conv2d_nhwc_twin_opts", _lib.ptr(twin), _lib.ptr(img), _lib.ptr(bias), _lib.ptr(out), n, hi, wi, cin, ho, wo, k,
                      out.stride(2), r, s, stride[0], stride[1], off0[0], off0[1], doff[0], doff[1], ctypes.byref(copt), tag, _lib.stream())
            return twin
        _lib.call("diga_conv2d_nhwc_twin", _lib.ptr(twin), _lib.ptr(img), _lib.ptr(bias), _lib.ptr(out), n, hi, wi, cin, ho, wo, k,
                  out.stride(2), r, s, stride[0], stride[1], off0[0], off0[1], doff[0], doff[1], _lib.ptr(stats), tag,
                  _lib.stream())
        return twin
    if must_twin:
        raise RuntimeError("DigaConv2d: the input holds split-twin bytes but the twin kernel is not selected "
                           "(conv math or config.conv_twin changed since the producer ran)")
    if _lib.get_conv_math() == 1:
        # split-bf16 arithmetic: the weights are split once here (two bf16 arrays), the activations inside the kernel
        nel = w_krsc.numel()
        w_hi = torch.empty(nel, dtype=torch.int16, device=w_krsc.device)
        w_lo = torch.empty(nel, dtype=torch.int16, device=w_krsc.device)
        _lib.call("diga_split_bf16", _lib.ptr(w_krsc), _lib.ptr(w_hi), _lib.ptr(w_lo), nel, _lib.stream())
        if epi is not None:
            _lib.call("diga_conv2d_nhwc_bf16x3_epi", _lib.ptr(x), _lib.ptr(w_hi), _lib.ptr(w_lo), _lib.ptr(out),
                      n, hi, wi, cin, x.stride(2), ho, wo, k, out.stride(2), r, s, stride[0], stride[1], off0[0], off0[1],
                      doff[0], doff[1], ctypes.byref(epi), tag, _lib.stream())
            return None
        if copt is not None:
            _lib.call("diga_conv2d_nhwc_bf16x3_opts", _lib.ptr(x), _lib.ptr(w_hi), _lib.ptr(w_lo), _lib.ptr(bias), _lib.ptr(out),
                      n, hi, wi, cin, x.stride(2), ho, wo, k, out.stride(2), r, s, stride[0], stride[1], off0[0], off0[1],
                      doff[0], doff[1], ctypes.byref(copt), tag, _lib.stream())
            return None
        _lib.call("diga_conv2d_nhwc_bf16x3", _lib.ptr(x), _lib.ptr(w_hi), _lib.ptr(w_lo), _lib.ptr(bias), _lib.ptr(out),
                  n, hi, wi, cin, x.stride(2), ho, wo, k, out.stride(2), r, s, stride[0], stride[1], off0[0], off0[1],
                  doff[0], doff[1], _lib.ptr(stats), tag, _lib.stream())
        return None
    direct = 2.0 * n * ho * wo * k * r * s * cin
    name = "conv_bwd_data" if tag == _TAG_BWD_DATA else "conv_fwd"
    if (_lib.get_conv_math() == 0 and copt is not None and copt.upsample_shift == 0 and copt.activation == 0 and copt.reflect_pad
            and doff[0] > 0 and doff[0] < min(hi, wi) and _winograd_ok(n, hi, wi, cin, k, r, s, stride, off0, doff, ho, wo)
            and _wino_plan(hi, wi, doff[0])[0] >= 4):
        # reflection padding folded into the Winograd input transform (the translator's 3x3 ResBlock convs; round 5)
        d = doff[0]
        tile, ratio = _wino_plan(hi, wi, d)
        _log_flops(name, direct, direct * ratio)
        ws = _lib.workspace(_lib.lib.diga_conv2d_winograd_workspace_bytes(n, hi, wi, cin, k, d, tile), x.device, "winograd")
        _lib.call("diga_conv2d_winograd_f32_opts", _lib.ptr(x), _lib.ptr(w_krsc), _lib.ptr(bias), _lib.ptr(out), _lib.ptr(ws), ws.numel(),
                  n, hi, wi, cin, x.stride(2), k, out.stride(2), d, tile, ctypes.byref(copt), _lib.ptr(_tile_table(n, hi, wi, d, tile, x.device)),
                  tag, _lib.stream())
        return None
    if (_lib.get_conv_math() == 0 and copt is None and (stats is None or wino_stats)
            and _winograd_ok(n, hi, wi, cin, k, r, s, stride, off0, doff, ho, wo)):
        d = abs(doff[0])
        tile, ratio = _wino_plan(hi, wi, d)
        if stats is not None and (tile < 4 or epi is not None or doff[0] < 0):
            raise RuntimeError("DigaConv2d: Winograd statistics come with the forward output transform of 4x4 / 6x6 tiles")
        _log_flops(name, direct, direct * ratio)
        nbytes = _lib.lib.diga_conv2d_winograd_workspace_bytes(n, hi, wi, cin, k, d, tile)
        ws = _lib.workspace(nbytes, x.device, "winograd")
        tab = _tile_table(n, hi, wi, d, tile, x.device)
        if epi is not None:
            _lib.call("diga_conv2d_winograd_f32_epi", _lib.ptr(x), _lib.ptr(w_krsc), _lib.ptr(out), _lib.ptr(ws), ws.numel(),
                      n, hi, wi, cin, x.stride(2), k, out.stride(2), d, tile, 1 if doff[0] < 0 else 0, ctypes.byref(epi), _lib.ptr(tab), tag,
                      _lib.stream())
            return None
        if keep_v is not None and doff[0] > 0:
            # keep_v: a one-element list -- the transformed input stays alive for this layer's weight gradient
            keep_v[0] = _alloc_keep_v(_lib.lib.diga_conv2d_winograd_v_floats(n, hi, wi, cin, d, tile), x.device)
        if keep_v is not None and doff[0] > 0 and keep_v[0] is not None:
            _lib.call("diga_conv2d_winograd_f32_keep", _lib.ptr(x), _lib.ptr(w_krsc), _lib.ptr(bias), _lib.ptr(out), _lib.ptr(keep_v[0]),
                      _lib.ptr(ws), ws.numel(), n, hi, wi, cin, x.stride(2), k, out.stride(2), d, tile, _lib.ptr(stats), _lib.ptr(tab), tag,
                      _lib.stream())
            return None
        _lib.call("diga_conv2d_winograd_f32", _lib.ptr(x), _lib.ptr(w_krsc), _lib.ptr(bias), _lib.ptr(out), _lib.ptr(ws), ws.numel(),
                  n, hi, wi, cin, x.stride(2), k, out.stride(2), d, tile, 1 if doff[0] < 0 else 0, _lib.ptr(stats), _lib.ptr(tab), tag,
                  _lib.stream())
        return None
    _log_flops(name, direct, direct)
    if epi is not None:
        _lib.call("diga_conv2d_nhwc_f32_epi", _lib.ptr(x), _lib.ptr(w_krsc), _lib.ptr(out), n, hi, wi, cin,
                  x.stride(2), ho, wo, k, out.stride(2), r, s, stride[0], stride[1], off0[0], off0[1], doff[0], doff[1],
                  ctypes.byref(epi), tag, _lib.stream())
        return None
    if copt is not None:
        _lib.call("diga_conv2d_nhwc_f32_opts", _lib.ptr(x), _lib.ptr(w_krsc), _lib.ptr(bias), _lib.ptr(out), n, hi, wi, cin,
                  x.stride(2), ho, wo, k, out.stride(2), r, s, stride[0], stride[1], off0[0], off0[1], doff[0], doff[1],
                  ctypes.byref(copt), tag, _lib.stream())
        return None
    _lib.call("diga_conv2d_nhwc_f32", _lib.ptr(x), _lib.ptr(w_krsc), _lib.ptr(bias), _lib.ptr(out), n, hi, wi, cin,
              x.stride(2), ho, wo, k, out.stride(2), r, s, stride[0], stride[1], off0[0], off0[1], doff[0], doff[1],
              _lib.ptr(stats), tag, _lib.stream())


class _StemConvFn(torch.autograd.Function):
    """Few-channel conv (the 7x7/2 stem on the 3-channel image): im2col into rows of R*S*C (padded to 32)
    floats, then a 1x1 conv on the GEMM kernels.  No gradient wrt the input (it is the image)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, dilation, stats=None, uses=None, twin_box=None):
        _lib.require_gpu(x, weight)
        xc = x.detach().float().contiguous()                      # NCHW
        n, c, h, w_ = xc.shape
        k, _, r, s = weight.shape
        ho = (h + 2 * padding[0] - (r - 1) - 1) // stride[0] + 1
        wo = (w_ + 2 * padding[1] - (s - 1) - 1) // stride[1] + 1
        kk = r * s * c
        kp = _pad_to(kk)
        # student and teacher read the very same image batch (train_step.py: both get `cat`): the gathered rows depend on
        # the images and the geometry only, so the second stem to arrive reuses them (the entry rides on the input tensor
        # and dies with it; an event orders the two streams)
        key = (r, s, stride[0], padding[0], kp, x._version, x.data_ptr())
        hit = getattr(x, "_diga_xcol", None)
        cur = torch.cuda.current_stream(x.device)
        if hit is not None and hit[0] == key:
            xcol = hit[1]
            cur.wait_event(hit[2])
            xcol.record_stream(cur)
        else:
            xcol = torch.empty((n, ho, wo, kp), dtype=torch.float32, device=x.device)
            _lib.call("diga_im2col_nchw", _lib.ptr(xc), _lib.ptr(xcol), n, c, h, w_, r, s, stride[0], padding[0], ho, wo, kp,
                      _lib.stream())
            ev = torch.cuda.Event()
            ev.record(cur)
            x._diga_xcol = (key, xcol, ev)
        w2 = _pad_last(weight.detach().permute(0, 2, 3, 1).reshape(k, 1, 1, kk).contiguous(), kp)
        out = torch.empty((n, ho, wo, k), dtype=torch.float32, device=x.device)
        b = None if bias is None else bias.detach().float().contiguous()
        _conv_launch(xcol, w2, b, out, (1, 1), (0, 0), (1, 1), _TAG_FWD, stats)
        ctx.save_for_backward(xcol)
        ctx.geom = (k, c, r, s, kk, kp, bias is not None, weight.stride())
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, grad_out):
        (xcol,) = ctx.saved_tensors
        k, c, r, s, kk, kp, has_bias, w_strides = ctx.geom
        if ctx.needs_input_grad[0]:
            raise NotImplementedError("the stem conv path does not produce a gradient wrt its (image) input")
        n, ho, wo, _ = xcol.shape
        gy = grad_out.permute(0, 2, 3, 1)
        if not gy.is_contiguous():
            gy = gy.contiguous()
        kq = _pad_to(k)
        gyp = _pad_last(gy, kq)
        dw = db = None
        if ctx.needs_input_grad[1]:
            dwp = torch.empty((kq, 1, 1, kp), dtype=torch.float32, device=xcol.device)
            nbytes = _lib.lib.diga_conv2d_wgrad_workspace_bytes(n, ho, wo, kq, kp, 1, 1)
            ws = _lib.workspace(nbytes, xcol.device, "wgrad")
            _lib.call("diga_conv2d_wgrad_nhwc_f32", _lib.ptr(gyp), _lib.ptr(xcol), _lib.ptr(dwp), _lib.ptr(ws), ws.numel(),
                      n, ho, wo, kp, kp, ho, wo, kq, kq, 1, 1, 1, 1, 0, 0, 1, 1, _lib.get_conv_math(), _lib.stream())
            dw = torch.empty_strided((k, c, r, s), w_strides, dtype=torch.float32, device=xcol.device)
            dw.copy_(dwp[:k, 0, 0, :kk].reshape(k, r, s, c).permute(0, 3, 1, 2))
        if has_bias and ctx.needs_input_grad[2]:
            db = _bias_grad(gy)
        return None, dw, db, None, None, None, None, None, None


def _set_mask(epi, box, xn, cp):
    """ReLU mask of a BatchNorm with residual for the backward epilogue: its bit mask when the forward wrote one
    (norm._BnFn, `mask_bits`), else its output (= this conv's input xn)."""
    if not box["has_res"]:
        epi.mask_y, epi.mask_ld = None, 0
    elif box.get("mask_bits") is not None:
        epi.mask_y, epi.mask_ld = None, 0
        epi.mask_bits, epi.mask_bits_ld = _lib.ptr(box["mask_bits"]), cp // 8
    else:
        epi.mask_y, epi.mask_ld = _lib.ptr(xn), cp


class _Conv2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, dilation, stats=None, uses=None, twin_box=None, x_is_twin=False,
                dy_is_twin=False, bn_box=None, opts=None, chain=None):
        # x: NCHW-shaped; weight: [K,C,R,S] (any dense layout); returns an NCHW-shaped channels_last tensor
        _lib.require_gpu(x, weight)
        if x_is_twin:          # the producer wrote the split twin instead of fp32 (same bytes per element): hand it on
            xt = x.detach().permute(0, 2, 3, 1)
            if not (xt.is_contiguous() and xt.dtype == torch.float32):
                raise RuntimeError("DigaConv2d: a twin-only input must be a dense NHWC float32-shaped buffer")
            twin_box = [xt.reshape(-1).view(torch.uint8)]
        xn = x.detach().permute(0, 2, 3, 1)
        if not xn.is_contiguous() or xn.dtype != torch.float32:
            xn = xn.contiguous().float()
        k, c, r, s = weight.shape
        cp = _pad_to(c)
        xn = _pad_last(xn, cp)
        w = weight.detach().permute(0, 2, 3, 1)
        if not w.is_contiguous():
            w = w.contiguous()
        w = _pad_last(w, cp)
        n, hi, wi, _ = xn.shape
        up = int(opts[1]) if opts is not None else 0
        if opts is not None and any(opts) and (torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad)):
            raise RuntimeError("DigaConv2d: folded reflect padding / upsampling / tanh are inference-only (no backward)")
        ho = ((hi << up) + 2 * padding[0] - dilation[0] * (r - 1) - 1) // stride[0] + 1
        wo = ((wi << up) + 2 * padding[1] - dilation[1] * (s - 1) - 1) // stride[1] + 1
        out = torch.empty((n, ho, wo, k), dtype=torch.float32, device=x.device)
        b = None if bias is None else bias.detach().float().contiguous()
        # exact-fp32 Winograd layers whose weight gradient is wanted keep their transformed input (4x the input's bytes, HBM
        # is 288 GB): the backward-weight pass then skips a bandwidth pass of 5x the input (config.winograd_keep_v = False: recompute)
        keep_v = None
        wino_stats = isinstance(stats, tuple)         # (buffer, "records"): the Winograd output transform fills it (DigaConv2d.forward)
        if wino_stats:
            stats = stats[0]
        if (ctx.needs_input_grad[1] and _lib.get_conv_math() == 0 and k % 256 == 0 and cp % 128 == 0 and (stats is None or wino_stats)
                and (opts is None or not any(opts)) and config.active().winograd_keep_v
                and _winograd_ok(n, hi, wi, cp, k, r, s, stride, (-padding[0], -padding[1]), dilation, ho, wo)):
            keep_v = [None]
        x_twin = _conv_launch(xn, w, b, out, stride, (-padding[0], -padding[1]), dilation, _TAG_FWD, stats, twin_box,
                              must_twin=bool(x_is_twin), opts=opts, keep_v=keep_v, wino_stats=wino_stats)
        ctx.wino_v = keep_v[0] if keep_v is not None else None
        ctx.max_tile = config.active().winograd_max_tile      # (the kept transform's layout is the forward's tile: checked in backward)
        ctx.save_for_backward(xn, w)
        # the split twin of the input serves the weight gradient too (multi-tap / shared-input layers, Cout >= 256)
        ctx.x_twin = x_twin if (ctx.needs_input_grad[1] and k >= 256 and k % 8 == 0 and cp == c) else None
        ctx.dy_is_twin = bool(dy_is_twin)
        if dy_is_twin and ctx.x_twin is None and ctx.needs_input_grad[1]:
            raise RuntimeError("DigaConv2d: twin_grad=True on a layer whose weight gradient is not on the twin kernel")
        if x_is_twin and (x_twin is None or (ctx.needs_input_grad[1] and ctx.x_twin is None)):
            raise RuntimeError("DigaConv2d: got a twin-only input but this layer is not on the twin kernels "
                               "(check takes_twin_only_input before asking the producer for a twin)")
        ctx.geom = (stride, padding, dilation, c, bias is not None, weight.stride())
        ctx.uses = uses
        # data-parallel runs: the parameter offers its slice of the all-reduce bucket as the gradient's home (ddp.GradReducer.grad_view)
        ctx.grad_view = getattr(weight, "_diga_grad_view", None) if isinstance(weight, nn.Parameter) else None
        ctx.weight_param = weight if ctx.grad_view is not None else None
        # the arithmetic / twin decisions of this forward bind its backward: saved tensors may hold twin bytes
        ctx.math = _lib.get_conv_math()
        ctx.x_is_twin = bool(x_is_twin)
        # the BatchNorm that produced x lets this conv's backward-data epilogue finish its incoming gradient (norm._BnFn)
        ctx.bn_box = None
        if (bn_box is not None and ctx.needs_input_grad[0] and stride == (1, 1) and cp == c and c % 4 == 0
                and "claimed" not in bn_box and bn_box.get("rows") == n * hi * wi and bn_box.get("C") == c
                and (not bn_box["has_res"] or not x_is_twin)):
            bn_box["claimed"] = True
            bn_box["consumer_ready"] = True
            ctx.bn_box = bn_box
        # several convs on one tensor (the ASPP branches): their input gradients are summed by chaining the running sum
        # through the backward-data epilogues (`addend`) instead of autograd adds; the last one to run also finishes the
        # gradient of the BatchNorm that produced the tensor.  All members must be eligible, else the chain is off.
        ctx.chain = chain
        if chain is not None:
            if not (ctx.needs_input_grad[0] and stride == (1, 1) and cp == c and c % 4 == 0 and not x_is_twin):
                chain["disabled"] = True
            box = chain.get("box")
            if box is not None and ("claimed" in box or box.get("rows") != n * hi * wi or box.get("C") != c):
                chain["box"] = None
            elif box is not None:
                chain["box_ok"] = True
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, grad_out):
        xn, w = ctx.saved_tensors
        stride, padding, dilation, c_true, has_bias, w_strides = ctx.geom
        if (ctx.x_is_twin or ctx.dy_is_twin or ctx.x_twin is not None) and _lib.get_conv_math() != ctx.math:
            raise RuntimeError("DigaConv2d: the conv arithmetic (_lib.set_conv_math) changed between forward and backward of a graph that "
                               "holds split-twin tensors (their bytes are only readable by the twin kernels)")
        if ctx.x_is_twin and ctx.needs_input_grad[1] and ctx.x_twin is None:
            raise RuntimeError("DigaConv2d: twin-only input saved for backward but the weight gradient is off the twin kernel")
        n, hi, wi, cp = xn.shape
        k, r, s, _ = w.shape
        gy = grad_out.permute(0, 2, 3, 1)
        if not gy.is_contiguous():
            gy = gy.contiguous()
        _, ho, wo, _ = gy.shape
        kp = _pad_to(k)
        gyp = _pad_last(gy, kp)
        dx = dw = db = None
        st = _lib.stream()
        x_twin = getattr(ctx, "x_twin", None)
        use_tw = x_twin is not None and ctx.needs_input_grad[1] and _lib.get_conv_math() == 1 and kp == k
        dy_box = [None] if use_tw else None            # the twin of dy: built once, read by backward-data and -weight
        if ctx.dy_is_twin:          # the BatchNorm after this conv wrote its dx as a twin (same bytes per element)
            if not (kp == k and _lib.get_conv_math() == 1 and cp > 64 and (use_tw or not ctx.needs_input_grad[1])):
                raise RuntimeError("DigaConv2d: twin gradient on a layer that is not on the twin kernels")
            dy_box = [gyp.reshape(-1).view(torch.uint8)]
        if ctx.needs_input_grad[0]:
            # backward-data = stride-1 correlation of dy with the [C][R][S][K] transpose, tap offsets negated
            if kp == k:
                wt = torch.empty((cp, r, s, kp), dtype=torch.float32, device=w.device)
                _lib.call("diga_weight_transpose", _lib.ptr(w), _lib.ptr(wt), k, r * s, cp, st)
            else:                                   # 19-class head only: tiny, padded with torch ops
                wt = torch.zeros((cp, r, s, kp), dtype=torch.float32, device=w.device)
                wt[..., :k] = w.permute(3, 1, 2, 0)
            if stride == (1, 1):
                dxn = torch.empty((n, hi, wi, cp), dtype=torch.float32, device=w.device)
                epi, box = None, ctx.bn_box
                # rows per partial-sum record of the backward-data epilogue this call will run (the Winograd form writes one
                # record per tile group: any divisor of the row count the finaliser is told works, 128 as before)
                epi_chunk = _lib.lib.diga_conv2d_epi_chunk_rows(n, ho, wo, kp, hi, wi, cp, r, s, 1, 1, padding[0], padding[1],
                                                                  _lib.get_conv_math())
                chain = ctx.chain if (ctx.chain is not None and not ctx.chain.get("disabled")) else None
                last_of_chain = False
                if chain is not None:
                    # `remaining` counts the members whose backward has not run yet IN THIS backward pass.  The running sum
                    # only reaches autograd through the member that brings it to 0, so a pass in which a member never runs
                    # (its output unused / detached) or a second pass over a retained graph would hand the trunk a missing
                    # or stale sum: both are refused loudly instead.
                    if chain.get("done"):
                        raise RuntimeError("DigaConv2d: a second backward pass through convolutions that chain their input "
                                           "gradients (retain_graph) is not supported; run the forward again")
                    chain["remaining"] -= 1
                    if chain["remaining"] < 0:
                        raise RuntimeError("DigaConv2d: gradient chain ran more members than it has")
                    last_of_chain = chain["remaining"] == 0
                    if last_of_chain:
                        chain["done"] = True
                    prev = chain.get("acc")
                    cbox = chain.get("box") if (last_of_chain and chain.get("box_ok")) else None
                    if prev is not None or cbox is not None:
                        epi = _lib.BwdEpilogue()
                        epi.addend, epi.addend_ld = _lib.ptr(prev), cp
                        if cbox is not None:
                            m_rows = n * hi * wi
                            part = torch.empty(((m_rows + 63) // 64) * 2 * cp, dtype=torch.float32, device=w.device)      # (room for 64-row chunks)
                            _set_mask(epi, cbox, xn, cp)
                            epi.x, epi.x_ld = _lib.ptr(cbox["x"]), cp
                            epi.relu_ab = _lib.ptr(cbox["relu_ab"]) if not cbox["has_res"] else None
                            epi.mean, epi.invstd, epi.partials = _lib.ptr(cbox["mean"]), _lib.ptr(cbox["invstd"]), _lib.ptr(part)
                            cbox["claimed"] = True
                            cbox["premasked"] = (dxn.data_ptr(), part, dxn, prev, dxn._version, epi_chunk)
                    chain["acc"] = dxn
                elif box is not None:
                    # finish the gradient of the BatchNorm in front of this conv in the epilogue: + residual-branch
                    # gradient, ReLU mask, sum g / sum g*xhat per 128-row chunk (include/diga_hip.h, diga_bwd_epilogue_t)
                    m_rows = n * hi * wi
                    part = torch.empty(((m_rows + 63) // 64) * 2 * cp, dtype=torch.float32, device=w.device)      # (room for 64-row chunks)
                    # the residual-branch gradient of the tensor this conv reads, left by the BatchNorm that took it as
                    # residual -- whether or not the producing BN itself had a residual (the epilogue combines `addend`
                    # with either mask form)
                    add = box.pop("dres", None)
                    box["consumed"] = True          # a residual gradient arriving from now on must go through autograd
                    epi = _lib.BwdEpilogue()
                    epi.addend, epi.addend_ld = _lib.ptr(add), cp
                    _set_mask(epi, box, xn, cp)
                    epi.x, epi.x_ld = _lib.ptr(box["x"]), cp
                    epi.relu_ab = _lib.ptr(box["relu_ab"]) if not box["has_res"] else None
                    epi.mean, epi.invstd, epi.partials = _lib.ptr(box["mean"]), _lib.ptr(box["invstd"]), _lib.ptr(part)
                    box["premasked"] = (dxn.data_ptr(), part, dxn, add, dxn._version, epi_chunk)
                _conv_launch(gyp, wt, None, dxn, (1, 1), (padding[0], padding[1]), (-dilation[0], -dilation[1]),
                             _TAG_BWD_DATA, None, dy_box if ((use_tw or ctx.dy_is_twin) and cp > 64) else None,
                             must_twin=ctx.dy_is_twin, epi=epi)
            else:
                if (r, s) != (1, 1) or padding != (0, 0):
                    raise NotImplementedError("backward-data of strided convs is only needed (and built) for 1x1")
                dense = torch.empty((n, ho, wo, cp), dtype=torch.float32, device=w.device)
                _conv_launch(gyp, wt, None, dense, (1, 1), (0, 0), (1, 1), _TAG_BWD_DATA)
                dxn = torch.zeros((n, hi, wi, cp), dtype=torch.float32, device=w.device)
                dxn[:, ::stride[0], ::stride[1]] = dense
            dx = dxn[..., :c_true].permute(0, 3, 1, 2)
            if ctx.chain is not None and not ctx.chain.get("disabled") and ctx.chain["remaining"] > 0:
                dx = None                           # the running sum travels on through the chain; the last member returns it
        if ctx.needs_input_grad[1]:
            # A weight that entered the graph more than once (the self-training step runs the student twice) gets its
            # contributions summed by autograd on the main stream: only the first one of a backward pass may still be
            # in flight on the side stream when it is handed over, the later ones run in line after a join.
            # A weight that is NOT a leaf (uses == INLINE_WGRAD: the folded matrices of the SegFormer head, whose gradient the next
            # autograd node reads at once) is computed in line on the main stream.
            later = False
            inline = isinstance(ctx.uses, str)
            if ctx.uses is not None and not inline:
                later = ctx.uses[0] > 0
                ctx.uses[0] += 1
            # data-parallel runs: the gradient is written straight into the parameter's slice of its all-reduce bucket
            # (ddp.GradReducer.grad_view: a fresh tensor of the weight's shape and strides over that slice) -- nothing packs it
            gv = getattr(ctx, "grad_view", None)
            dw = gv() if (gv is not None and not later and not inline and kp == k and cp == c_true) else None
            if dw is not None:
                # ... unless the slice already IS the parameter's gradient: a second graph differentiated before reduce() (the
                # overlapped self-training step's student(cross_mix), gradient accumulation under GradReducer.hold()) would overwrite
                # the first graph's result and AccumulateGrad would then add the slice to itself (this check replaced a
                # module-global switch the step driver had to flip around the second forward)
                held = ctx.weight_param.grad
                if held is not None and held.data_ptr() == dw.data_ptr():
                    dw = None
            if dw is None or tuple(dw.stride()) != tuple(w_strides) or dw.device != w.device:
                dw = torch.empty_strided((k, c_true, r, s), w_strides, dtype=torch.float32, device=w.device)
            # the kernels write [K][R][S][C]: for an unpadded channels_last weight that IS dw's memory (no copy afterwards)
            alias = kp == k and cp == c_true and dw.permute(0, 2, 3, 1).is_contiguous()
            dwp = dw.permute(0, 2, 3, 1) if alias else torch.empty((kp, r, s, cp), dtype=torch.float32, device=w.device)

            dy_twin = None
            if use_tw:
                dy_twin = dy_box[0]
                if dy_twin is None:                 # backward-data did not build it (no input gradient, strided, narrow)
                    dy_twin = torch.empty(n * ho * wo * kp * 4, dtype=torch.uint8, device=w.device)
                    _lib.call("diga_make_twin", _lib.ptr(gyp), kp, _lib.ptr(dy_twin), n * ho * wo, kp, st)

            wino_v = getattr(ctx, "wino_v", None)
            ctx.wino_v = None
            if wino_v is not None and config.active().winograd_max_tile != ctx.max_tile:
                raise RuntimeError("DigaConv2d: the Winograd tile cap changed between this layer's forward and its backward "
                                   f"({ctx.max_tile} -> {config.active().winograd_max_tile}): the kept input transform has the forward's "
                                   "layout (set_conv_math(..., exact=) belongs between steps, not inside one)")

            def run_twin():
                nbytes = _lib.lib.diga_conv2d_wgrad_twin_workspace_bytes(n, ho, wo, kp, cp, r, s)
                ws = _lib.workspace(nbytes, w.device, "wgrad")
                _lib.call("diga_conv2d_wgrad_twin", _lib.ptr(dy_twin), _lib.ptr(x_twin), _lib.ptr(dwp), _lib.ptr(ws), ws.numel(),
                          n, hi, wi, cp, ho, wo, kp, r, s, stride[0], stride[1], -padding[0], -padding[1], dilation[0],
                          dilation[1], _lib.stream())
                if not alias:
                    dw.copy_(dwp[:k, :, :, :c_true].permute(0, 3, 1, 2))

            def run():
                if use_tw:
                    return run_twin()
                if (_lib.get_conv_math() == 0 and kp % 256 == 0 and cp % 128 == 0 and gyp.stride(2) % 4 == 0
                        and _winograd_ok(n, hi, wi, cp, kp, r, s, stride, (-padding[0], -padding[1]), dilation, ho, wo)):
                    tile, ratio = _wino_plan(hi, wi, dilation[0])
                    _log_flops("conv_bwd_weight", 2.0 * n * ho * wo * kp * 9 * cp, 2.0 * n * ho * wo * kp * 9 * cp * ratio)
                    nb = _lib.lib.diga_conv2d_wgrad_winograd_workspace_bytes(n, hi, wi, cp, kp, dilation[0], tile, 1 if wino_v is not None else 0)
                    wsw = _lib.workspace(nb, w.device, "winograd_wgrad")
                    tab = _tile_table(n, hi, wi, dilation[0], tile, w.device)
                    _lib.call("diga_conv2d_wgrad_winograd_f32", _lib.ptr(gyp), _lib.ptr(xn), _lib.ptr(wino_v), _lib.ptr(dwp), _lib.ptr(wsw),
                              wsw.numel(), n, hi, wi, cp, xn.stride(2), kp, gyp.stride(2), dilation[0], tile, _lib.ptr(tab), _lib.stream())
                    if not alias:
                        dw.copy_(dwp[:k, :, :, :c_true].permute(0, 3, 1, 2))
                    return
                _log_flops("conv_bwd_weight", 2.0 * n * ho * wo * kp * r * s * cp, 2.0 * n * ho * wo * kp * r * s * cp)
                nbytes = _lib.lib.diga_conv2d_wgrad_workspace_bytes(n, ho, wo, kp, cp, r, s)
                ws = _lib.workspace(nbytes, w.device, "wgrad")
                _lib.call("diga_conv2d_wgrad_nhwc_f32", _lib.ptr(gyp), _lib.ptr(xn), _lib.ptr(dwp), _lib.ptr(ws), ws.numel(),
                          n, hi, wi, cp, xn.stride(2), ho, wo, kp, gyp.stride(2), r, s, stride[0], stride[1],
                          -padding[0], -padding[1], dilation[0], dilation[1], _lib.get_conv_math(), _lib.stream())
                if not alias:
                    dw.copy_(dwp[:k, :, :, :c_true].permute(0, 3, 1, 2))

            # The weight gradient is a leaf of the backward graph: under the step driver it runs on a second stream
            # next to the backward-data / BatchNorm chain (the driver joins the streams before the optimizer step);
            # `later` / `inline` (above) keep the exceptions in line.
            if later:
                _lib.join_side()
            side = None if (later or inline) else _lib.side_stream(w.device)
            if side is None:
                run()
            else:
                side.wait_stream(torch.cuda.current_stream(w.device))
                with torch.cuda.stream(side):
                    run()
                dw.record_stream(side)          # (a returned gradient: never held, see _lib.release_to_side)
                _lib.release_to_side(side, (gyp, xn) + (() if alias else (dwp,)) + ((dy_twin, x_twin) if use_tw else ())
                                     + ((wino_v,) if wino_v is not None else ()))
        if has_bias and ctx.needs_input_grad[2]:
            db = _bias_grad(gy)
        return dx, dw, db, None, None, None, None, None, None, None, None, None, None, None


class DigaConv2d(nn.Conv2d):
    """nn.Conv2d whose compute is the HIP implicit-GEMM kernel; weights kept in channels_last memory."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if self.groups != 1 or self.padding_mode != "zeros":
            raise NotImplementedError("DigaConv2d: groups=1 and zero padding only")
        self.emit_bn_stats = False      # set by the model on convs that feed a train-mode BatchNorm
        self._bw_seen = [0]
        self.share_twin = False         # set by the model on convs that share their input with other convs (ASPP branches)
        self.weight.data = self.weight.data.contiguous(memory_format=torch.channels_last)

    def _apply(self, fn, *a, **k):
        super()._apply(fn, *a, **k)
        if self.weight.dim() == 4 and not self.weight.data.is_contiguous(memory_format=torch.channels_last):
            self.weight.data = self.weight.data.contiguous(memory_format=torch.channels_last)
        return self

    def forward(self, x, twin_grad=False, opts=None, chain=None):
        """twin_grad: the gradient of this conv's output will arrive as a split twin (the BatchNorm that consumes the
        output was called with dx_twin=True).  opts = (reflect_pad, upsample_shift, activation): inference-only input
        map / output activation folded into the kernel (diga_conv_options_t, the `_opts` entry points)."""
        fn = _Conv2dFn
        if (self.in_channels < 8 and not x.requires_grad and tuple(self.dilation) == (1, 1)
                and self.stride[0] == self.stride[1] and self.padding[0] == self.padding[1]):
            fn = _StemConvFn           # image-like input: gather the few channels of all taps into the K dimension
        stats = None
        if self.emit_bn_stats and self.training and self.out_channels % 4 == 0:
            n, _, h, w = x.shape
            ho = (h + 2 * self.padding[0] - self.dilation[0] * (self.kernel_size[0] - 1) - 1) // self.stride[0] + 1
            wo = (w + 2 * self.padding[1] - self.dilation[1] * (self.kernel_size[1] - 1) - 1) // self.stride[1] + 1
        wino_records = None
        if self.emit_bn_stats and self.training and self.out_channels % 4 == 0:
            on_wino = (_lib.get_conv_math() == 0 and fn is _Conv2dFn and (opts is None or not any(opts))
                       and _winograd_ok(n, h, w, _pad_to(self.in_channels), self.out_channels, self.kernel_size[0], self.kernel_size[1],
                                        self.stride, (-self.padding[0], -self.padding[1]), tuple(self.dilation), ho, wo))
            if not on_wino:
                stats = torch.empty(_lib.lib.diga_conv2d_stats_floats(n, ho, wo, self.out_channels), dtype=torch.float32,
                                    device=x.device)
            elif config.active().winograd_stats:
                # round 5: the Winograd output transform (4x4 / 6x6 tiles) leaves the statistics as records of unequal size
                # (diga_bn_fwd_records); F(2x2)-only runs keep the BatchNorm's own statistics pass
                plan = winograd_stats_plan(n, h, w, _pad_to(self.in_channels), self.out_channels, self.kernel_size[0], self.kernel_size[1],
                                           self.stride, self.padding, self.dilation, ho, wo)
                if plan is not None:
                    wino_records = int(plan[1])
                    stats = (torch.empty(plan[0], dtype=torch.float32, device=x.device), "records")
        uses = None
        if torch.is_grad_enabled() and self.weight.requires_grad:
            if self._bw_seen[0] > 0:          # a backward pass has consumed the previous graph(s)
                self._bw_seen[0] = 0
            uses = self._bw_seen              # [weight-gradient calls of the current backward pass]
        twin_box = None
        if self.share_twin:                   # several convs read this very tensor: the first one builds its split twin
            twin_box = getattr(x, "_diga_twin_box", None)
            if twin_box is None:
                twin_box = x._diga_twin_box = [None]
        x_is_twin = bool(getattr(x, "_diga_is_twin", False))
        if x_is_twin and fn is not _Conv2dFn:
            raise RuntimeError("DigaConv2d: twin-only input on the stem path")
        bn_box = getattr(x, "_diga_bn_box", None)
        if bn_box is not None and (self.share_twin or chain is not None or fn is not _Conv2dFn or not torch.is_grad_enabled()):
            bn_box = None                     # several convs read this tensor (the chain carries the box) / no backward
        if opts is not None and any(opts):
            if fn is not _Conv2dFn or stats is not None:
                raise RuntimeError("DigaConv2d: folded padding / upsampling / activation need the implicit-GEMM path without BN statistics")
            y = fn.apply(x, self.weight, self.bias, tuple(self.stride), tuple(self.padding), tuple(self.dilation), None, uses,
                         twin_box, False, False, None, tuple(int(v) for v in opts))
        elif x_is_twin or twin_grad or bn_box is not None or chain is not None:
            if fn is not _Conv2dFn or (self.bias is not None and twin_grad):
                if chain is not None:
                    chain["disabled"] = True
                else:
                    raise RuntimeError("DigaConv2d: twin gradient needs a bias-free conv on the implicit-GEMM path")
            y = fn.apply(x, self.weight, self.bias, tuple(self.stride), tuple(self.padding), tuple(self.dilation), stats, uses,
                         twin_box, x_is_twin, bool(twin_grad), bn_box, None, chain) if fn is _Conv2dFn else \
                fn.apply(x, self.weight, self.bias, tuple(self.stride), tuple(self.padding), tuple(self.dilation), stats, uses, twin_box)
        else:
            y = fn.apply(x, self.weight, self.bias, tuple(self.stride), tuple(self.padding), tuple(self.dilation), stats, uses,
                         twin_box)
        if wino_records is not None:
            y._diga_bn_partials = (stats[0], ("records", wino_records))
        elif stats is not None:
            chunk = _lib.lib.diga_conv2d_stats_chunk_rows(n, h, w, _pad_to(self.in_channels), ho, wo, self.out_channels,
                                                          self.kernel_size[0], self.kernel_size[1], self.stride[0], self.stride[1],
                                                          -self.padding[0], -self.padding[1], _lib.get_conv_math())
            y._diga_bn_partials = (stats, chunk)     # picked up by the DigaBatchNorm2d that consumes y
        return y
