"""Training utilities of the DiGA hot path on MI355X -- same names and call signatures as the
reference's `util/utils.py`:
  poly_lr_scheduler / adjust_learning_rate   G5/util/utils.py:32-41
  save_models / load_models                  G5/util/utils.py:83-91
  create_teacher_params / update_teacher_params   G5/util/utils.py:93-116
  UnNormalize / Normalize                    G5/util/utils.py:126-156
  process_label                              G5/util/utils.py:158-163
plus what the reference inlines in its scripts and the build provides as functions:
  classmix / classmix_select / classmix_paste   warm_up.py:240-259, self_training.py:306-325
  DigaSGD (fused, duplicate-aware momentum SGD)  warm_up.py:156,301-305; SURVEY App. A-9
All device work is done by libdiga_hip.so.  No CPU fallback.
"""
import os
import random
import sys

import numpy as np
import torch

_pkg = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.path.dirname(_pkg) not in sys.path:
    sys.path.append(os.path.dirname(_pkg))
from diga_amd import _lib  # noqa: E402

CHUNK_ELEMS = 16384          # elements of one tensor handled by one 256-thread block


# --------------------------------------------------------------------------- LR schedule
def poly_lr_scheduler_warm(base_lr, iter, warmup=1000, max_iter=80000, power=1.0):
    if iter <= warmup:
        return base_lr * (iter / warmup)
    return base_lr * ((1 - float(iter - warmup) / max_iter) ** power)


def poly_lr_scheduler(base_lr, iter, max_iter=30000, power=0.9):
    return base_lr * ((1 - float(iter) / max_iter) ** power)


def _set_lr(opts, lr):
    for opt in opts:
        opt.param_groups[0]["lr"] = lr
        if len(opt.param_groups) > 1:
            opt.param_groups[1]["lr"] = lr * 10


def adjust_learning_rate(opts, base_lr, i_iter, max_iter, power):
    _set_lr(opts, poly_lr_scheduler(base_lr, i_iter, max_iter, power))


def adjust_learning_rate_warm(opts, base_lr, i_iter, max_iter, power):
    _set_lr(opts, poly_lr_scheduler_warm(base_lr, i_iter, max_iter, power))


# --------------------------------------------------------------------------- checkpoints
def save_models(model_dict, prefix="./"):
    """One `<key>.pth` state_dict per entry (same on-disk format as the reference)."""
    os.makedirs(prefix, exist_ok=True)
    for key, value in model_dict.items():
        torch.save(value.state_dict(), os.path.join(prefix, key + ".pth"))


def load_models(model_dict, prefix="./"):
    for key, value in model_dict.items():
        value.load_state_dict(torch.load(os.path.join(prefix, key + ".pth"), map_location="cpu"))


# --------------------------------------------------------------------------- multi-tensor tables
class TensorTables:
    """Device-side pointer/size/chunk tables for one list of equally shaped tensor lists."""

    def __init__(self, sizes, device):
        self.device = device
        self.n = len(sizes)
        self.sizes_host = list(sizes)
        ct, cs = [], []
        for i, n in enumerate(sizes):
            for start in range(0, n, CHUNK_ELEMS):
                ct.append(i)
                cs.append(start)
        self.n_chunks = len(ct)
        self.sizes = torch.tensor(sizes, dtype=torch.int64, device=device)
        self.chunk_tensor = torch.tensor(ct, dtype=torch.int32, device=device)
        self.chunk_start = torch.tensor(cs, dtype=torch.int64, device=device)
        self._ptrs = {}

    def pointers(self, slot, tensors):
        """Device int64 table of data pointers; re-uploaded only when a pointer changed."""
        host = [t.data_ptr() for t in tensors]
        cached = self._ptrs.get(slot)
        if cached is None or cached[0] != host:
            dev = torch.tensor(host, dtype=torch.int64, device=self.device)
            self._ptrs[slot] = (host, dev)
            return dev
        return cached[1]


def same_dense_layout(a, b):
    """True when a and b are dense and store element (i,j,..) at the same offset, so that an
    elementwise kernel may walk their raw memory (contiguous, or both channels_last)."""
    if a.shape != b.shape:
        return False
    if a.is_contiguous() and b.is_contiguous():
        return True
    return (a.dim() == 4 and a.is_contiguous(memory_format=torch.channels_last)
            and b.is_contiguous(memory_format=torch.channels_last))


def _check_param_lists(a, b, what):
    if len(a) != len(b):
        raise ValueError(f"{what}: {len(a)} vs {len(b)} parameters")
    for x, y in zip(a, b):
        if x.dtype != torch.float32 or y.dtype != torch.float32:
            raise TypeError(f"{what}: fp32 parameters expected")
        if not same_dense_layout(x, y):
            raise ValueError(f"{what}: parameters differ in shape or memory layout "
                             f"({tuple(x.shape)}/{x.stride()} vs {tuple(y.shape)}/{y.stride()})")
    _lib.require_gpu(*a, *b)


_ema_tables = {}


def _ema_multi(teacher_params, student_params, alpha):
    t = [p.data for p in teacher_params]
    s = [p.data for p in student_params]
    _check_param_lists(t, s, "update_teacher_params")
    key = tuple(x.data_ptr() for x in t) + tuple(x.data_ptr() for x in s)
    tab = _ema_tables.get(key)
    if tab is None:
        _ema_tables.clear()              # one (teacher, student) pair is live at a time
        tab = _ema_tables[key] = TensorTables([x.numel() for x in t], t[0].device)
    tp, sp = tab.pointers("t", t), tab.pointers("s", s)
    one_minus = float(1 - alpha)         # rounded in double first, as `(1 - alpha_teacher)` is
    _lib.call("diga_ema_update_multi", _lib.ptr(tp), _lib.ptr(sp), _lib.ptr(tab.sizes), _lib.ptr(tab.chunk_tensor),
              _lib.ptr(tab.chunk_start), tab.n_chunks, CHUNK_ELEMS, float(alpha), one_minus, _lib.stream())


def ema_alpha(iteration, stage0=True, mean=False, replace=False):
    if stage0:
        return min(1 - 1 / (iteration + 1), 0.999)
    if mean:
        return 0.9
    if replace:
        return 0.0
    return 0.999


def create_teacher_params(teacher, student):
    """Teacher := copy of the student's PARAMETERS (buffers keep their own values)."""
    for param in teacher.parameters():
        param.detach_()
    with torch.no_grad():
        for tp, sp in zip(teacher.parameters(), student.parameters()):
            tp.data.copy_(sp.data)
    return teacher.cuda()


def update_teacher_params(teacher, student, iteration, stage0=True, mean=False, replace=False):
    """theta_t <- a*theta_t + (1-a)*theta_s over parameters() (incl. frozen BN affine, not buffers),
    a = min(1 - 1/(it+1), 0.999): one multi-tensor launch for the whole model."""
    alpha = ema_alpha(iteration, stage0, mean, replace)
    tflat, sflat = getattr(teacher, "flat_params", None), getattr(student, "flat_params", None)
    if tflat is not None and sflat is not None and tflat.numel() == sflat.numel():
        _lib.require_gpu(tflat, sflat)
        _lib.call("diga_ema_update_flat", _lib.ptr(tflat), _lib.ptr(sflat), tflat.numel(), float(alpha),
                  float(1 - alpha), _lib.stream())
    else:
        _ema_multi(list(teacher.parameters()), list(student.parameters()), alpha)
    # the reference returns teacher.cuda(); everything above already required device tensors, and Module.cuda() walks every
    # module and buffer (~1 ms of host time per step on ResNet-101) only to find nothing to move
    return teacher


# --------------------------------------------------------------------------- normalisation helpers
class UnNormalize(object):
    def __init__(self, mean, std):
        self.mean, self.std = mean, std

    def __call__(self, tensor):
        m = torch.as_tensor(self.mean, dtype=tensor.dtype, device=tensor.device).view(1, -1, 1, 1)
        s = torch.as_tensor(self.std, dtype=tensor.dtype, device=tensor.device).view(1, -1, 1, 1)
        return tensor * s + m


class Normalize(object):
    def __init__(self, mean, std):
        self.mean, self.std = mean, std

    def __call__(self, tensor):
        m = torch.as_tensor(self.mean, dtype=tensor.dtype, device=tensor.device).view(1, -1, 1, 1)
        s = torch.as_tensor(self.std, dtype=tensor.dtype, device=tensor.device).view(1, -1, 1, 1)
        return (tensor - m) / s


def process_label(label, class_numbers=19):
    """One-hot with a (class_numbers+1)-th bucket for everything >= class_numbers."""
    batch, channel, w, h = label.size()
    pred1 = torch.zeros(batch, class_numbers + 1, w, h, device=label.device)
    idx = torch.where(label < class_numbers, label, torch.full_like(label, class_numbers))
    return pred1.scatter_(1, idx.long(), 1)


# --------------------------------------------------------------------------- ClassMix
def classmix_present(labels):
    """Per image, the ascending list of label values present (== torch.unique(img).tolist()),
    from ONE device histogram and ONE D->H copy for the whole batch."""
    _lib.require_gpu(labels)
    lab = _lib.contiguous(labels, torch.int64)
    b = lab.shape[0]
    hist = torch.zeros((b, 256), dtype=torch.int32, device=lab.device)
    _lib.call("diga_label_hist256", _lib.ptr(lab), _lib.ptr(hist), b, lab[0].numel(), _lib.stream())
    present = (hist != 0).cpu().numpy()
    return [np.nonzero(present[i])[0].tolist() for i in range(b)]


class PinnedRing(object):
    """A few page-locked host buffers reused round-robin for the step's small asynchronous copies.  Allocating page-locked
    memory inside the step is not free: every allocation maps pages into the GPU's address space, and with two processes on
    one GPU (the 2-rank test configuration) a fresh pinned allocation per step slowed every kernel of BOTH processes down
    25-75x (tools/diag/two_rank_legs.sh).  A slot is handed out again only after the event of its previous use completed."""

    def __init__(self, shape, dtype, slots=4):
        self.bufs = [torch.empty(shape, dtype=dtype, pin_memory=True) for _ in range(slots)]
        self.events = [None] * slots
        self.i = 0

    def take(self):
        i = self.i
        self.i = (i + 1) % len(self.bufs)
        if self.events[i] is not None:
            self.events[i].synchronize()
        return i, self.bufs[i]

    def mark(self, i, event):
        self.events[i] = event


_PRESENT_RINGS = {}


def classmix_present_async(labels, stream):
    """classmix_present started on `stream` (a side stream): returns (pinned host tensor [B,256] of 0/1, event).  The caller
    guarantees that `labels` is complete (its producer has finished); nothing here waits for the current stream, so the D->H copy
    does not queue behind the training step in flight.  `present_lists(host)` turns the result into the per-image lists (read it
    before four more calls with the same batch size reuse the buffer)."""
    _lib.require_gpu(labels)
    lab = _lib.contiguous(labels, torch.int64)
    b = lab.shape[0]
    ring = _PRESENT_RINGS.get(b)
    if ring is None:
        ring = _PRESENT_RINGS[b] = PinnedRing((b, 256), torch.bool, slots=8)
    with torch.cuda.stream(stream):
        hist = torch.zeros((b, 256), dtype=torch.int32, device=lab.device)
        _lib.call("diga_label_hist256", _lib.ptr(lab), _lib.ptr(hist), b, lab[0].numel(), _lib.stream())
        slot, host = ring.take()
        host.copy_(hist != 0, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(stream)
        ring.mark(slot, ev)
    lab.record_stream(stream)
    return host, ev


def present_lists(host):
    present = host.numpy()
    return [np.nonzero(present[i])[0].tolist() for i in range(present.shape[0])]


def classmix_select(present, rng=random):
    """Reference class choice: random.sample(half of the classes present) from Python's global RNG,
    one call per image, then 255 is force-added (warm_up.py:247-250)."""
    sels = []
    for lst in present:
        pick = rng.sample(lst, len(lst) // 2)
        if 255 not in pick:
            pick.append(255)
        sels.append(pick)
    return sels


def classmix_paste(background, foreground, labels, selections, bg_labels=None):
    """out = foreground where labels in selections[b] else background (mask broadcast over channels);
    with bg_labels also the pasted label map (self_training.py:318-319)."""
    _lib.require_gpu(background, foreground, labels)
    bg = _lib.contiguous(background, torch.float32)
    fg = _lib.contiguous(foreground, torch.float32)
    lab = _lib.contiguous(labels, torch.int64)
    b, ch = bg.shape[0], bg.shape[1]
    hw = lab[0].numel()
    if fg.shape != bg.shape or lab.shape[0] != b or bg[0, 0].numel() != hw:
        raise ValueError("classmix_paste: shape mismatch")
    lut_host = np.zeros((b, 256), dtype=np.uint8)
    for i, sel in enumerate(selections):
        lut_host[i, np.asarray(sel, dtype=np.int64)] = 1
    lut = torch.from_numpy(lut_host).to(bg.device, non_blocking=True)
    out = torch.empty_like(bg)
    bgl = lab_out = None
    if bg_labels is not None:
        bgl = _lib.contiguous(bg_labels, torch.int64)
        lab_out = torch.empty_like(lab)
    _lib.call("diga_classmix_paste", _lib.ptr(bg), _lib.ptr(fg), _lib.ptr(lab), _lib.ptr(lut), _lib.ptr(out),
              _lib.ptr(bgl), _lib.ptr(lab_out), b, ch, hw, _lib.stream())
    return out if bg_labels is None else (out, lab_out)


def classmix(background, foreground, labels, rng=random, bg_labels=None, present=None):
    """The whole cross-domain mixture block.  Returns (mixed[, mixed_labels], selections).  present: the per-image class lists
    when the caller already has them (DigaTrainer.prefetch_classmix), else one histogram launch + one D->H copy here."""
    sels = classmix_select(classmix_present(labels) if present is None else present, rng)
    res = classmix_paste(background, foreground, labels, sels, bg_labels)
    return (res, sels) if bg_labels is None else (res[0], res[1], sels)


# --------------------------------------------------------------------------- fused SGD
class DigaSGD(torch.optim.Optimizer):
    """torch.optim.SGD(momentum, weight_decay) semantics (single-tensor path) for the param groups the
    reference builds, where a tensor may occur k times in a group (k sequential micro-steps sharing one
    momentum buffer; on the very first step every occurrence starts from a fresh buffer, as torch 2.x does).
    One launch per step for the whole model; `param_groups[i]['lr']` stays the knob that
    adjust_learning_rate() turns.  Duplicate entries follow torch's single-tensor path (foreach=False; torch 2.x's
    foreach path on GPU would treat them differently: shared buffer multiplied k times, all d = g + wd*p formed before
    any update) -- the reference pins torch 1.7.1, which has only the single-tensor path.  momentum and weight_decay
    are one scalar each for the launch: every group must carry the same values (the reference's two groups do).
    Momentum buffers live in `self.state[p]['momentum_buffer']`, so state_dict()/load_state_dict() round-trip them; the
    'first step' condition is not stored: it is inferred from the absence of momentum buffers (as torch does)."""

    def __init__(self, params, lr=2.5e-4, momentum=0.9, weight_decay=5e-4, grad_scale=1.0):
        groups = []
        for g in (params if isinstance(params, (list, tuple)) else [{"params": params}]):
            g = dict(g) if isinstance(g, dict) else {"params": g}
            plist = list(g["params"])
            uniq, mult = [], {}
            for p in plist:
                if id(p) not in mult:
                    uniq.append(p)
                    mult[id(p)] = 0
                mult[id(p)] += 1
            g["params"] = uniq
            g["mult"] = [mult[id(p)] for p in uniq]
            groups.append(g)
        super().__init__(groups, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        self.grad_scale = float(grad_scale)
        self._tab = None
        self._first = True

    def _build(self):
        self._params, self._mult, self._group_of = [], [], []
        for gi, g in enumerate(self.param_groups):
            for p, k in zip(g["params"], g["mult"]):
                if p.requires_grad:
                    self._params.append(p)
                    self._mult.append(k)
                    self._group_of.append(gi)
        dev = self._params[0].device
        _lib.require_gpu(*self._params)
        for key in ("momentum", "weight_decay"):
            vals = {float(g[key]) for g in self.param_groups}
            if len(vals) != 1:
                raise ValueError(f"DigaSGD: all param groups must share one {key} (got {sorted(vals)}); "
                                 "per-group values are not supported by the fused kernel")
        self._bufs = []
        for p in self._params:
            st = self.state[p]
            buf = st.get("momentum_buffer")
            if buf is None or buf.shape != p.shape or buf.device != p.device or not same_dense_layout(buf, p.data):
                loaded = buf
                buf = torch.zeros_like(p.data)
                if loaded is not None:                 # restored by load_state_dict: bring to the parameter's layout
                    buf.copy_(loaded.to(p.device))
                    self._first = False
                st["momentum_buffer"] = buf
            else:
                self._first = False
            self._bufs.append(buf)
        self._tab = TensorTables([p.numel() for p in self._params], dev)
        self._mult_dev = torch.tensor(self._mult, dtype=torch.int32, device=dev)
        self._lr_host = None
        self._lr_ring = None
        self._lr_dev = torch.empty(len(self._params), dtype=torch.float32, device=dev)

    @torch.no_grad()
    def step(self, closure=None, found_inf=None):
        """found_inf: a device int32 tensor; when its first element is non-zero at execution time the step changes nothing (the
        decision is taken inside the kernel: no host sync).  DigaTrainer passes the student's `grad_overflow` flag."""
        if self._tab is None:
            self._build()
        g0 = self.param_groups[0]
        grads = []
        for p in self._params:
            if p.grad is None:
                raise RuntimeError("DigaSGD: every trainable parameter must have a gradient")
            g = p.grad
            if not same_dense_layout(g, p.data):          # walk raw memory: layouts must agree
                g = torch.empty_like(p.data).copy_(g)
            grads.append(g)
        lrs = [float(self.param_groups[gi]["lr"]) for gi in self._group_of]
        if lrs != self._lr_host:
            # (the poly schedule changes the rates every step.)  Asynchronous copy from a page-locked ring slot: a blocking
            # copy from pageable memory here drained the stream once per step -- the host could never run ahead of the GPU
            if self._lr_ring is None:
                self._lr_ring = PinnedRing((len(lrs),), torch.float32, slots=4)
            slot, pin = self._lr_ring.take()
            pin.copy_(torch.tensor(lrs, dtype=torch.float32))
            self._lr_dev.copy_(pin, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._lr_ring.mark(slot, ev)
            self._lr_host = lrs
        tab = self._tab
        pp = tab.pointers("p", [p.data for p in self._params])
        gp = tab.pointers("g", grads)
        bp = tab.pointers("b", self._bufs)
        _lib.call("diga_sgd_momentum_multi", _lib.ptr(pp), _lib.ptr(gp), _lib.ptr(bp), _lib.ptr(tab.sizes),
                  _lib.ptr(self._mult_dev), _lib.ptr(self._lr_dev), _lib.ptr(tab.chunk_tensor),
                  _lib.ptr(tab.chunk_start), tab.n_chunks, CHUNK_ELEMS, float(g0["momentum"]),
                  float(g0["weight_decay"]), 1 if self._first else 0, self.grad_scale,
                  _lib.ptr(found_inf) if found_inf is not None else None, _lib.stream())
        if found_inf is not None:
            _lib.flag_consumed(found_inf)          # the owning model clears it at its next training forward, not before
        self._first = False
        return None

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._tab = None                   # rebuild the pointer tables (and adopt the restored buffers) on the next step

    def momentum_buffers(self):
        return {id(p): b for p, b in zip(self._params, self._bufs)} if self._tab else {}
