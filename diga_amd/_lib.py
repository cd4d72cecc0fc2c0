"""ctypes binding of libdiga_hip.so (include/diga_hip.h).

There is no CPU fallback: importing this module without the built library, or calling an op
with non-GPU tensors, raises.  Build with `python -m diga_amd.build`.
"""
import collections
import ctypes as C
import os

import torch

from diga_amd import config

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DIGA_LIB") or os.path.join(_HERE, "libdiga_hip.so")      # DIGA_LIB: A/B builds of the library

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: the DiGA hot path has no CPU fallback. "
        "Build the HIP extension first: python -m diga_amd.build")

lib = C.CDLL(LIB_PATH)

P = C.c_void_p
I64 = C.c_int64
F32 = C.c_float
INT = C.c_int
SZ = C.c_size_t

# name -> (restype, argtypes); mirrors include/diga_hip.h and include/diga_mit.h one to one
SIGNATURES = {
    "diga_version": (INT, []),
    "diga_last_error_string": (C.c_char_p, []),
    "diga_loss_workspace_bytes": (SZ, [I64]),
    "diga_ce2d_fwd_bwd": (INT, [P, P, P, P, P, SZ, I64, I64, I64, I64, F32, P]),
    "diga_ohem_ce_workspace_bytes": (SZ, [I64, I64, I64]),
    "diga_ohem_ce_fwd_bwd": (INT, [P, P, P, P, P, SZ, I64, I64, I64, I64, I64, F32, I64, F32, P]),
    "diga_distill_fwd_bwd": (INT, [P, P, P, P, P, SZ, I64, I64, I64, I64, F32, F32, P]),
    "diga_scale_inplace": (INT, [P, P, I64, P]),
    "diga_upsample_loss_workspace_bytes": (SZ, [I64, I64, I64, I64]),
    "diga_upsample_ce_distill_fwd_bwd": (INT, [P, P, P, P, P, P, SZ, I64, I64, I64, I64, I64, I64, F32, F32, F32, P]),
    "diga_upsample_ce_fwd_bwd": (INT, [P, P, P, P, P, SZ, I64, I64, I64, I64, I64, I64, F32, P]),
    "diga_upsample_bilinear_ac": (INT, [P, P, I64, I64, I64, I64, I64, P]),
    "diga_ema_update_flat": (INT, [P, P, I64, F32, F32, P]),
    "diga_ema_update_multi": (INT, [P, P, P, P, P, I64, I64, F32, F32, P]),
    "diga_sgd_momentum_multi": (INT, [P, P, P, P, P, P, P, P, I64, I64, F32, F32, INT, F32, P, P]),
    "diga_nonfinite_flag_f32": (INT, [P, I64, P, P]),
    "diga_label_hist256": (INT, [P, P, I64, I64, P]),
    "diga_classmix_paste": (INT, [P, P, P, P, P, P, P, I64, I64, I64, P]),
    "diga_centroid_softmax_weights": (INT, [P, P, P, P, I64, I64, I64, I64, P]),
    "diga_upsample_argmax_consensus": (INT, [P, P, P, P, I64, I64, I64, I64, I64, I64, P]),
    "diga_class_mean_workspace_bytes": (SZ, [I64, I64]),
    "diga_class_mean_vectors": (INT, [P, P, P, P, P, P, P, SZ, I64, I64, I64, I64, I64, I64, I64, P]),
    "diga_centroid_ema_apply": (INT, [P, P, P, P, I64, I64, I64, I64, F32, INT, INT, P]),
    "diga_confusion_matrix": (INT, [P, P, P, I64, I64, P]),
    "diga_two_scale_confusion": (INT, [P, I64, I64, P, I64, I64, P, P, P, I64, I64, I64, I64, P]),
    "diga_conv2d_nhwc_f32": (INT, [P, P, P, P] + [I64] * 17 + [P, INT, P]),
    "diga_conv2d_stats_floats": (SZ, [I64, I64, I64, I64]),
    "diga_conv2d_stats_chunk_rows": (INT, [I64] * 13 + [INT]),
    "diga_conv2d_epi_chunk_rows": (INT, [I64] * 13 + [INT]),
    "diga_split_bf16": (INT, [P, P, P, I64, P]),
    "diga_conv2d_nhwc_bf16x3": (INT, [P, P, P, P, P] + [I64] * 17 + [P, INT, P]),
    "diga_make_twin": (INT, [P, I64, P, I64, I64, P]),
    "diga_split_bf16_image_bytes": (SZ, [I64, I64, I64]),
    "diga_split_bf16_image": (INT, [P, P, I64, I64, I64, P]),
    "diga_conv2d_nhwc_twin": (INT, [P, P, P, P] + [I64] * 16 + [P, INT, P]),
    "diga_conv2d_nhwc_f32_opts": (INT, [P, P, P, P] + [I64] * 17 + [P, INT, P]),
    "diga_conv2d_nhwc_bf16x3_opts": (INT, [P, P, P, P, P] + [I64] * 17 + [P, INT, P]),
    "diga_conv2d_nhwc_twin_opts": (INT, [P, P, P, P] + [I64] * 16 + [P, INT, P]),
    "diga_conv2d_wgrad_workspace_bytes": (SZ, [I64] * 7),
    "diga_conv2d_wgrad_nhwc_f32": (INT, [P, P, P, P, SZ] + [I64] * 17 + [INT, P]),
    "diga_conv2d_wgrad_twin_workspace_bytes": (SZ, [I64] * 7),
    "diga_conv2d_wgrad_twin": (INT, [P, P, P, P, SZ] + [I64] * 15 + [P]),
    "diga_weight_transpose": (INT, [P, P, I64, I64, I64, P]),
    "diga_im2col_nchw": (INT, [P, P] + [I64] * 11 + [P]),
    "diga_norm_workspace_bytes": (SZ, [I64, I64, I64]),
    "diga_bn_fwd": (INT, [P, I64, P, I64, P, I64, P, P, P, P, P, P, P, I64, I64, INT, INT, INT, P, F32, F32, P, SZ, P]),
    "diga_bn_fwd_partials": (INT, [P, I64, P, I64, P, I64, P, P, P, P, P, P, P, I64, I64, INT, INT, P, F32, F32, P, I64, P, SZ, P]),
    "diga_bn_fwd_records": (INT, [P, I64, P, I64, P, I64, P, P, P, P, P, P, P, I64, I64, INT, INT, P, F32, F32, P, P, I64, P, SZ, P]),
    "diga_bn_bwd": (INT, [P, I64, P, I64, P, I64, P, P, P, P, P, I64, P, I64, I64, I64, INT, INT, P, SZ, P]),
    "diga_bn_bwd_affine": (INT, [P, I64, P, I64, P, I64, P, P, P, P, P, I64, P, P, I64, I64, INT, P, SZ, P]),
    "diga_pyramid_sum_fwd": (INT, [P, I64, I64, P, P, I64, I64, P, I64, I64, P, I64, I64, I64, I64, P]),
    "diga_pyramid_sum_bwd": (INT, [P, I64, I64, P, I64, I64, I64, I64, P]),
    "diga_pyramid_sum_fwd3": (INT, [P, I64, I64, P, P, P, P, P, I64, I64, P]),
    "diga_pyramid_sum_bwd3": (INT, [P, I64, I64, P, P, P, I64, I64, P]),
    "diga_bn_bwd_partials": (INT, [P, I64, P, I64, P, P, P, P, I64, I64, I64, INT, P, I64, P, SZ, P]),
    "diga_conv2d_nhwc_f32_epi": (INT, [P, P, P] + [I64] * 17 + [P, INT, P]),
    "diga_conv2d_winograd_workspace_bytes": (SZ, [I64] * 7),
    "diga_conv2d_winograd_tile_table_bytes": (SZ, [I64] * 5),
    "diga_conv2d_winograd_tile_table": (INT, [P] + [I64] * 5 + [P]),
    "diga_conv2d_winograd_stats_records": (SZ, [I64] * 6),
    "diga_conv2d_winograd_stats_floats": (SZ, [I64] * 6),
    "diga_conv2d_winograd_f32": (INT, [P, P, P, P, P, SZ] + [I64] * 9 + [INT, P, P, INT, P]),
    "diga_conv2d_winograd_f32_opts": (INT, [P, P, P, P, P, SZ] + [I64] * 9 + [P, P, INT, P]),
    "diga_conv2d_winograd_f32_epi": (INT, [P, P, P, P, SZ] + [I64] * 9 + [INT, P, P, INT, P]),
    "diga_conv2d_wgrad_winograd_workspace_bytes": (SZ, [I64] * 7 + [INT]),
    "diga_conv2d_wgrad_winograd_f32": (INT, [P, P, P, P, P, SZ] + [I64] * 9 + [P, P]),
    "diga_conv2d_winograd_v_floats": (SZ, [I64] * 6),
    "diga_conv2d_winograd_f32_keep": (INT, [P, P, P, P, P, P, SZ] + [I64] * 9 + [P, P, INT, P]),
    "diga_conv2d_nhwc_bf16x3_epi": (INT, [P, P, P, P] + [I64] * 17 + [P, INT, P]),
    "diga_conv2d_nhwc_twin_epi": (INT, [P, P, P] + [I64] * 16 + [P, INT, P]),
    "diga_gn_fwd": (INT, [P, I64, P, I64, P, P, P, P, P, I64, I64, I64, I64, INT, F32, P, SZ, P]),
    "diga_gn_bwd": (INT, [P, I64, P, I64, P, I64, P, P, P, P, P, I64, P, P, I64, I64, I64, I64, P, SZ, P]),
    "diga_avgpool_nhwc": (INT, [P, I64, P, I64, I64, I64, P, SZ, P]),
    "diga_colsum_nhwc": (INT, [P, I64, P, I64, I64, P, SZ, P]),
    "diga_small_linear_fwd": (INT, [P, P, P, P, I64, I64, I64, INT, P]),
    "diga_small_linear_bwd": (INT, [P, P, P, P, P, P, P, P, I64, I64, I64, INT, P]),
    "diga_channel_affine": (INT, [P, I64, P, I64, P, P, I64, I64, I64, P]),
    "diga_channel_dot": (INT, [P, I64, P, I64, P, I64, I64, I64, P, SZ, P]),
    "diga_maxpool3x3s2_fwd": (INT, [P, P, P, I64, I64, I64, I64, I64, I64, P]),
    "diga_maxpool3x3s2_bwd": (INT, [P, P, P, I64, I64, I64, I64, I64, I64, P]),
    "diga_color_aug_view": (INT, [P, P, P, P, I64, I64, I64, F32, P, P, P]),
    # ---- include/diga_mit.h
    "diga_mit_gemm_nt": (INT, [P, I64, P, I64, P, P, I64, INT, P, I64, P, I64, INT, F32, I64, I64, I64, P]),
    "diga_mit_gemm_tn_workspace_bytes": (SZ, [I64, I64, I64]),
    "diga_mit_gemm_tn": (INT, [P, I64, P, I64, P, P, F32, INT, P, SZ, I64, I64, I64, P]),
    "diga_mit_colsum_workspace_bytes": (SZ, [I64, I64]),
    "diga_mit_colsum": (INT, [P, I64, P, F32, INT, P, SZ, I64, I64, P]),
    "diga_mit_cast_transpose": (INT, [P, P, P, I64, I64, P]),
    "diga_mit_weight_prep_multi": (INT, [P, P, I64, I64, P]),
    "diga_mit_cast_scale": (INT, [P, P, I64, F32, P]),
    "diga_mit_row_scale": (INT, [P, P, P, I64, I64, I64, P]),
    "diga_mit_layernorm_fwd": (INT, [P, I64, P, P, P, P, I64, P, P, I64, I64, F32, P]),
    "diga_mit_layernorm_bwd_workspace_bytes": (SZ, [I64, I64]),
    "diga_mit_layernorm_bwd": (INT, [P, INT, I64, F32, P, I64, P, P, P, P, I64, P, P, I64, P, P, F32, INT, P, SZ, I64, I64, P]),
    "diga_mit_dwconv_gelu_fwd": (INT, [P, P, P, P, P, I64, I64, I64, I64, P]),
    "diga_mit_dwconv_bwd_workspace_bytes": (SZ, [I64, I64, I64]),
    "diga_mit_dwconv_gelu_bwd": (INT, [P, P, P, P, P, P, P, P, F32, INT, P, SZ, I64, I64, I64, I64, P]),
    "diga_mit_im2col": (INT, [P, INT, P] + [I64] * 11 + [P]),
    "diga_mit_col2im": (INT, [P, P, INT, F32] + [I64] * 11 + [P]),
    "diga_mit_attention_fwd": (INT, [P, I64, P, I64, P, I64, P, I64, I64, I64, I64, F32, P]),
    "diga_mit_attention_bwd_workspace_bytes": (SZ, [I64, I64, I64, I64]),
    "diga_mit_attention_bwd": (INT, [P, I64, P, I64, P, P, I64, P, P, P, P, SZ, I64, I64, I64, I64, F32, P]),
    "diga_prof_enable": (INT, [INT]),
    "diga_prof_reset": (INT, []),
    "diga_prof_query": (INT, [INT, P, P]),
    "diga_prof_query_work": (INT, [INT, P]),
}

class ConvOptions(C.Structure):
    """diga_conv_options_t of include/diga_hip.h."""
    _fields_ = [("reflect_pad", INT), ("upsample_shift", INT), ("activation", INT)]


class MitWeightPrep(C.Structure):
    """diga_mit_weight_prep_t of include/diga_mit.h."""
    _fields_ = [("src", P), ("out_a", P), ("out_b", P), ("Co", INT), ("Ci", INT), ("RS", INT), ("Kp", INT), ("mode", INT),
                ("tiles_k", INT)]


class BwdEpilogue(C.Structure):
    """diga_bwd_epilogue_t of include/diga_hip.h."""
    _fields_ = [("addend", P), ("addend_ld", I64), ("mask_y", P), ("mask_ld", I64), ("x", P), ("x_ld", I64),
                ("relu_ab", P), ("mean", P), ("invstd", P), ("partials", P), ("mask_bits", P), ("mask_bits_ld", I64)]


# enum order of include/diga_hip.h
PROF_TAGS = ["ce2d", "distill", "upsample_loss", "ema", "sgd", "classmix_hist", "classmix_paste",
             "centroid_weights", "consensus", "class_means", "centroid_apply", "conv_fwd", "conv_bwd_data",
             "conv_bwd_weight", "norm", "elementwise", "mit_gemm", "mit_wgrad", "mit_attn_fwd", "mit_attn_bwd", "mit_norm",
             "mit_dwconv", "mit_misc"]


def prof_query(tag):
    """(launch count, total ms) of one kernel family since the last prof_reset."""
    n, ms = C.c_int64(0), C.c_double(0.0)
    call("diga_prof_query", PROF_TAGS.index(tag), C.byref(n), C.byref(ms))
    return n.value, ms.value

def prof_work(tag):
    """Algorithmic work (bytes or FLOPs) the calls of one kernel family declared since the last prof_reset."""
    w = C.c_double(0.0)
    call("diga_prof_query_work", PROF_TAGS.index(tag), C.byref(w))
    return w.value


for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)          # AttributeError here = header and library out of sync
    _fn.restype = _res
    _fn.argtypes = _args


# ---- conv arithmetic: host-side policy (the library keeps no process-wide mode; include/diga_hip.h, DIGA_CONV_MATH_*)
CONV_MATH_F32, CONV_MATH_BF16X3 = 0, 1
# (the mode is a field of the active configuration, diga_amd/config.py; DIGA_CONV_MATH gives its default)


def set_conv_math(mode, exact=None):
    """0 / "f32": fp32 operands and fp32 accumulation on the fp32 matrix cores -- the 1x1 / strided / stem layers as a k-ordered
    fmaf chain, the stride-1 3x3 layers through fp32 Winograd with output tiles up to 6x6 (per-layer error vs float64 <= 2.7e-5 of
    scale instead of the direct chain's 3e-7); 1 / "bf16x3": fp32 operands split into bf16 hi+lo, three bf16
    MFMAs per product.  `exact=True` (fp32 only) caps every Winograd layer at F(2x2,3x3), whose transforms hold 0, +-1, +-1/2 only:
    the direct kernels' error level (7e-7), at ~25 % more step time; `exact=False` restores the default tiles; None leaves the
    tile cap alone.  Selects which entry points DigaConv2d calls from now on; graphs already built keep the arithmetic of their
    forward pass only where they hold split-twin tensors (DigaConv2d checks and raises otherwise)."""
    mode = {"f32": 0, "bf16x3": 1}.get(mode, mode)
    if mode not in (CONV_MATH_F32, CONV_MATH_BF16X3):
        raise ValueError(f"conv math must be 0 / 'f32' or 1 / 'bf16x3', not {mode!r}")
    if exact is not None:
        if exact and mode != CONV_MATH_F32:
            raise ValueError("exact=True belongs to the fp32 arithmetic")
        config.active().winograd_max_tile = 2 if exact else config.active().winograd_default_max_tile
    config.active().conv_math = int(mode)


def get_conv_math():
    return config.active().conv_math


def last_error():
    return lib.diga_last_error_string().decode("utf-8", "replace")


def call(name, *args):
    """Call an int-returning entry point; raise on any non-zero status."""
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed (code {rc}): {last_error()}")


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "diga_amd ops run on the GPU only (HIP kernels, no CPU fallback); got a "
                f"{t.device} tensor")


def contiguous(t, dtype=None):
    if t is None:
        return None
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t if t.is_contiguous() else t.contiguous()


_workspaces = {}

# ---- overflow flags (the fp16 MiT backward's device flag) that an optimizer step has read since their model last cleared them
_consumed_flags = set()


def flag_consumed(flag):
    """Called by DigaSGD.step(found_inf=flag): the step has been enqueued behind everything that could set the flag."""
    _consumed_flags.add((flag.device.index, flag.data_ptr()))


def take_consumed_flag(flag):
    """True once per optimizer step that read `flag` (the model clears the flag then); False for a forward that runs between a
    backward pass and the optimizer step that has yet to read its verdict."""
    key = (flag.device.index, flag.data_ptr())
    if key in _consumed_flags:
        _consumed_flags.discard(key)
        return True
    return False


# ---- optional second stream for work that is off the critical path (weight gradients during backward)
_side_streams = {}
_side_dirty = set()
_side_holds = {}                    # owning stream -> deque of (event on the side stream, owning stream, tensors): release_to_side
side_overlap = False          # switched on by the step driver around backward(); plain autograd users stay in line


def side_stream(device):
    """The per-device side stream when overlap is switched on, else None."""
    if not side_overlap or not config.active().wgrad_stream:
        return None
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _side_streams.get(idx)
    if st is None:
        st = _side_streams[idx] = torch.cuda.Stream(device=idx)
    _side_dirty.add(idx)
    return st


def active_side_stream(device):
    """The side stream of `device` if work has been put on it since the last join, else None."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    return _side_streams.get(idx) if idx in _side_dirty else None


def release_to_side(side, tensors):
    """Hand `tensors` (allocated on the CURRENT stream, just read or written by work enqueued on `side`) over for the side stream's use.

    `Tensor.record_stream` would do: the caching allocator then returns a freed block to its pool only once an event on the side stream
    has completed -- polled from the HOST at the next allocation.  The host runs a backward pass far ahead of the device, so those
    events are never complete when it asks, every such block (an activation, an output gradient, a kept transform) stays out of the
    pool until well into the next step, and the pool opens new segments instead: 126 GB reserved on the main stream for tensors that
    peak at 104 GB (profiles/r06_memory_by_stream.txt).  Instead the tensors are kept alive HERE for `wgrad_hold` (+ up to `wgrad_hold_batch`) more layers, and the
    stream that owns them is made to wait (on the device) for the side stream's event of that time before the reference is dropped:
    the block is then free in stream order, reusable by the very next allocation, and the wait is on work the side stream finished
    layers ago.  wgrad_hold = 0, and stream capture, keep record_stream.

    NEVER pass a tensor that is returned from backward as a gradient: AccumulateGrad takes a gradient over without a copy only while
    nobody else holds it; with a second reference it CLONES it on the main stream -- while the side stream is still writing it (found
    the hard way: bit-identity of the self-training step's stream forms).  Gradients keep record_stream (they live until the next
    zero_grad anyway)."""
    hold = config.active().wgrad_hold
    if hold <= 0 or torch.cuda.is_current_stream_capturing():
        for t in tensors:
            t.record_stream(side)
        return
    ev = torch.cuda.Event()
    ev.record(side)
    owner = torch.cuda.current_stream(side.device)
    key = (owner.device.index, owner.cuda_stream)        # one queue per owning stream: the self-training step feeds the side stream from two
    queue = _side_holds.get(key)
    if queue is None:
        queue = _side_holds[key] = collections.deque()
    queue.append((ev, owner, tensors))
    # released in batches: every device-side wait is a barrier packet in the owner's queue (one per layer measured +6 ms on the
    # self-training step's 2 x 105 layers); events of one stream complete in order, so the batch's LAST event covers the batch
    batch = max(1, config.active().wgrad_hold_batch)
    if len(queue) >= hold + batch:
        ev0, owner0, _ = queue[batch - 1]
        owner0.wait_event(ev0)
        for _ in range(batch):
            queue.popleft()


def held_for_side():
    """Number of entries release_to_side is holding (tests)."""
    return sum(len(q) for q in _side_holds.values())


def join_side():
    """Make the current stream of every device that used its side stream wait for it."""
    for idx in list(_side_dirty):
        torch.cuda.current_stream(idx).wait_stream(_side_streams[idx])
    _side_dirty.clear()
    for queue in _side_holds.values():
        if queue:
            ev0, owner, _ = queue[-1]
            owner.wait_event(ev0)
            queue.clear()


def workspace(nbytes, device, tag="default"):
    """Grow-only scratch buffer owned by the PyTorch caching allocator (one per device and tag)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), tag,
           torch.cuda.current_stream().cuda_stream)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 16), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf
