"""The build's own driver of one DiGA training step (what the `for i_iter` bodies of the reference's
scripts do, minus data loading, visualisation and logging):
  warm-up        G5/train_DiGA_gta2city_warm_up.py:197-305
  self-training  G5/train_DiGA_gta2city_self_training.py:214-387
Inputs the reference produces outside the scoped path (kornia colour augmentation `x_aug`, frozen
translator output `rec_s2t`) are arguments.

Differences in mechanics, not in results: the loss block runs at the low-res boundary with the
upsampling fused (no [2B,19,H,W] tensors), ClassMix costs one D->H copy per call instead of one
per image, EMA/SGD are single launches, no per-step .cpu()/.item() syncs; the unused
`student(tdatav)` visualisation forward (warm_up.py:265-266) is not executed.
"""
import random

import torch

from diga_amd import _lib, ddp
from diga_amd import config as _config
from diga_amd.util import loss as L
from diga_amd.util import utils as U


_STREAMS = {}
_MISMATCH_WARNING_OFF = False
# Stream policy, the self-training step's overlap form (c4_overlap) and graph capture are fields of diga_amd.config.StepConfig:
# DigaTrainer(config=...) owns one and runs every step under it; nothing here reads or writes the environment.


def _accumulate_on_other_streams_is_intended():
    """The overlapped self-training step runs the backward passes of two student graphs on two streams, so a parameter's
    AccumulateGrad node (created by whichever forward touched the parameter first) may sit on another stream than the node that
    hands it a gradient.  The autograd engine orders the two with events and joins every leaf stream with the caller's at the end
    of backward(); torch only warns that this costs a synchronisation.  Say that it is intended (once per process)."""
    global _MISMATCH_WARNING_OFF
    if not _MISMATCH_WARNING_OFF:
        fn = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
        if fn is not None:
            fn(False)
        _MISMATCH_WARNING_OFF = True


def _shared_stream(device, role):
    """The step driver's side streams (teacher forward, ClassMix prefetch) exist once per device and process, not once per
    DigaTrainer: every new HIP stream may become another hardware queue, and a process that had built a second trainer (the
    bench's second leg) next to another process on the same GPU (the 2-rank gloo test configuration) ran every kernel 25-75x
    slower once the queues were oversubscribed (tools/diag/two_rank_legs.sh)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _STREAMS.get((idx, role))
    if st is None:
        st = _STREAMS[(idx, role)] = torch.cuda.Stream(device=device)
    return st


def _prefetch_stream(device):
    return _shared_stream(device, "prefetch")


class DigaTrainer:
    def __init__(self, student, teacher, base_lr=2.5e-4, max_iter=80000, power=0.9, momentum=0.9,
                 weight_decay=5e-4, rng=random, distill_scale=0.5, centroid_exchange=None, graph=None, config=None):
        """config: a diga_amd.config.StepConfig (teacher / weight-gradient streams, c4 overlap form, graph capture, conv arithmetic,
        Winograd tile cap, keep-V policy, data-parallel buckets ...).  None = follow whatever configuration is active when a step is
        called (the process defaults unless the caller runs the step inside `config.use(...)` / `config.override(...)`).
        `centroid_exchange` / `graph`, when given, override the corresponding fields."""
        self.cfg = config
        cfg = self._cfg()
        self.student, self.teacher = student, teacher
        self.base_lr, self.max_iter, self.power = base_lr, max_iter, power
        self.rng = rng
        self.distill_scale = distill_scale
        self.world = ddp.world_size()
        # self-training, N > 1: "allgather" = exact (every rank applies all ranks' class means in global order, bit-identical
        # to one process on the concatenated batch); "allreduce" = BASELINE configs[3]'s cheaper, approximate exchange
        # (diga_amd/ddp.py::allreduce_class_means).  One collective per centroid pass either way.
        self.centroid_exchange = centroid_exchange or cfg.centroid_exchange
        if self.centroid_exchange not in ("allgather", "allreduce"):
            raise ValueError(f"centroid_exchange must be 'allgather' or 'allreduce', not {self.centroid_exchange!r}")
        self.opt = U.DigaSGD(student.optim_parameters(base_lr), lr=base_lr, momentum=momentum,
                             weight_decay=weight_decay, grad_scale=1.0 / self.world)
        # graph=True (or DIGA_STEP_GRAPH=1): the static part of the warm-up step -- both forward passes, the loss block and the
        # whole backward pass, ~1000-2500 launches -- is captured ONCE into a HIP graph (torch.cuda.CUDAGraph: stream capture of
        # the library's launches on torch's capture stream) and replayed every step; what depends on the iteration or on host
        # decisions stays outside (learning rate, EMA coefficient, ClassMix class choice, all-reduce, SGD).  For the launch-bound
        # configurations (small backbone, MiT encoder: hundreds of 5-30 us kernels); the big ResNet-101 step is GPU-bound either way.
        self.use_graph = cfg.step_graph if graph is None else bool(graph)
        self._g = None
        with _config.use(cfg):
            self.reducer = ddp.GradReducer([p for p in student.parameters() if p.requires_grad], overlap=not self.use_graph)
        self._side = None
        U.create_teacher_params(teacher, student)
        for p in teacher.parameters():
            p.requires_grad_(False)

    def _cfg(self):
        return self.cfg if self.cfg is not None else _config.active()

    # ------------------------------------------------------------------ ClassMix class lists ahead of time
    def prefetch_classmix(self, labels):
        """Software pipelining of the step's one host round trip: the class lists ClassMix needs (label histogram on the
        device, D->H copy) for an UPCOMING step's `labels`, started now on a side stream.  Call it as soon as the labels of
        the next batch are resident (a data loader's prefetch hook; `labels` must be complete -- nothing here waits for the
        training stream); the step that receives the same tensor then finds the lists ready instead of draining the GPU to
        learn which classes are present.  Same kernel, same copy, same RNG draws -- only earlier."""
        if not labels.is_cuda:
            return
        if getattr(self, "_pf_stream", None) is None:
            self._pf_stream = _prefetch_stream(labels.device)          # one per device and process, not per trainer
            self._pf = {}
        host, ev = U.classmix_present_async(labels, self._pf_stream)
        # keyed by the tensor OBJECT (held here, so its address cannot be reused by another tensor while the entry lives) and
        # its version counter: an in-place change of the labels invalidates the entry
        self._pf[id(labels)] = (labels, labels._version, host, ev)
        if len(self._pf) > 4:
            self._pf.pop(next(iter(self._pf)))

    def _present(self, labels):
        pf = getattr(self, "_pf", None)
        hit = pf.pop(id(labels), None) if pf else None
        if hit is None or hit[0] is not labels or hit[1] != labels._version:
            return None
        hit[3].synchronize()
        return U.present_lists(hit[2])

    # ------------------------------------------------------------------ common pieces
    def _begin(self, it):
        self.student.train()
        U.adjust_learning_rate([self.opt], self.base_lr, it, self.max_iter, self.power)
        with torch.no_grad():
            U.update_teacher_params(self.teacher, self.student, it)

    def _teacher_async(self, *inputs):
        """Teacher forward(s) (no grad) on a second HIP stream, concurrent with the student's forward on the current
        one: the passes are independent, and kernels of one fill the CUs that the tile tails of the other leave
        idle (config.teacher_stream = False runs them in line).  Returns one (logits, feat) pair per input; call
        `_teacher_join` on the result before using it."""
        dev = inputs[0].device
        if not (inputs[0].is_cuda and _config.active().teacher_stream):
            with torch.no_grad():
                return [self.teacher(x)[2:4] for x in inputs]
        if self._side is None:
            self._side = _shared_stream(dev, "teacher")
        main = torch.cuda.current_stream(dev)
        self._side.wait_stream(main)                 # EMA update of the teacher and the inputs are ready
        self._teacher_events = []
        with torch.cuda.stream(self._side), torch.no_grad():
            outs = []
            for x in inputs:
                outs.append(self.teacher(x)[2:4])
                ev = torch.cuda.Event()
                ev.record(self._side)                # (the overlapped self-training step waits for the passes one by one)
                self._teacher_events.append(ev)
        for x in inputs:
            x.record_stream(self._side)
        self._pending_join = True
        return outs

    def _student_and_teacher(self, student_in, *teacher_in):
        """Student forward on the current stream, teacher forward(s) on the side stream -- the teacher enqueued from a hook
        behind the student's first stage (config.teacher_offset, default `layer1`; `0` = both at once, as before).  Why the
        offset: student and teacher run the SAME kernel sequence, and two matrix-core kernels launched together share the
        CUs half and half, finish together and leave the BatchNorm / transform passes of both streams to run together with no
        matrix-core kernel in flight (tools/diag/overlap_timeline.py: 52 ms of the forward).  Started one stage apart, a
        persistent GEMM of one stream holds every CU while the other stream's bandwidth passes run under it, and the two
        alternate from then on.  Returns (student outputs, pending teacher outputs for `_teacher_join`)."""
        where = _config.active().teacher_offset
        stage = None
        if where not in ("", "0"):
            stage = self.student
            for part in where.split("."):                       # a stage ("layer1") or a block inside one ("layer1.1", "layer2.0")
                stage = getattr(stage, part, None) if not part.isdigit() else (stage[int(part)] if stage is not None and int(part) < len(stage) else None)
                if stage is None:
                    break
        if not isinstance(stage, torch.nn.Module) or not _config.active().teacher_stream or not student_in.is_cuda:
            pending = self._teacher_async(*teacher_in)
            return self.student(student_in), pending
        box = {}

        def launch(mod, inp, out):
            if "p" not in box:
                box["p"] = self._teacher_async(*teacher_in)

        handle = stage.register_forward_hook(launch)
        try:
            s_out = self.student(student_in)
        finally:
            handle.remove()
        if "p" not in box:                                     # (the stage did not run in this forward)
            box["p"] = self._teacher_async(*teacher_in)
        return s_out, box["p"]

    def _teacher_join(self, outs):
        if getattr(self, "_pending_join", False):
            main = torch.cuda.current_stream(outs[0][0].device)
            main.wait_stream(self._side)
            for pair in outs:
                for t in pair:
                    t.record_stream(main)
            self._pending_join = False
        return outs

    def _finish(self, total):
        self.opt.zero_grad(set_to_none=True)
        _lib.side_overlap = True          # weight gradients on the side stream (diga_amd/model/conv.py)
        try:
            total.backward()
        finally:
            _lib.side_overlap = False
            _lib.join_side()
        self.reducer.reduce()
        self._opt_step()

    def _opt_step(self):
        """SGD with the student's overflow flag (loss-scaled fp16 backward of the MiT student; None for the fp32 ResNet): the skip
        is decided inside the kernel.  N > 1: the flag is max-reduced first so that every rank skips the same steps."""
        flag = getattr(self.student, "grad_overflow", None)
        if flag is not None and self.world > 1:
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
        self.opt.step(found_inf=flag)
        self._steps_done = getattr(self, "_steps_done", 0) + 1
        if flag is not None and self._steps_done % 200 == 0:
            before = self.student.loss_scale
            self.student.adjust_loss_scale(steps=200)
            if self.student.loss_scale != before:
                self._g = None        # the scale is baked into a captured graph: capture again

    # ------------------------------------------------------------------ warm-up step, static part as a HIP graph
    def _warmup_step_graphed(self, it, x, x_aug, rec_s2t, labels, lambda_seg, lambda_distil):
        """Step 0 runs eagerly (lazy initialisations, allocator warm-up); step 1 captures; every later step replays.
        BatchNorm running statistics, DropPath / Dropout2d draws (torch's graph-safe Philox offsets) and the gradients evolve
        exactly as in the eager step: capture records, only replays execute."""
        g = self._g
        key = (tuple(x.shape), tuple(labels.shape), float(lambda_seg), float(lambda_distil))
        if g is None:
            self._g = g = {"calls": 0}
        if g["calls"] == 0 or g.get("key", key) != key:
            if g.get("key", key) != key:
                raise RuntimeError("DigaTrainer(graph=True): the captured step is bound to its input shapes and loss weights")
            g["calls"] = 1
            g["key"] = key
            return self._warmup_step_eager(it, x, x_aug, rec_s2t, labels, lambda_seg, lambda_distil)
        self._begin(it)
        with torch.no_grad():
            mix, _ = U.classmix(rec_s2t, x_aug, labels, self.rng, present=self._present(labels))
        if "graph" not in g:
            B = x.shape[0]
            g["cat"] = torch.empty((2 * B,) + tuple(x.shape[1:]), dtype=torch.float32, device=x.device)
            g["labels"] = torch.empty_like(labels)
            g["cat"][:B].copy_(x)
            g["cat"][B:].copy_(mix)
            g["labels"].copy_(labels)
            self.opt.zero_grad(set_to_none=True)          # the captured backward ASSIGNS the gradients (static buffers of the graph's pool)
            cfg = _config.active()
            flag = getattr(self.student, "grad_overflow", None)
            if flag is not None:
                _lib.flag_consumed(flag)                  # (the captured forward must contain the flag's clear)
            # one capture stream: the teacher / weight-gradient side streams are off while capturing, except as FORKED branches
            with _config.use(cfg.replace(teacher_stream=False, wgrad_stream=False)):
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    if cfg.graph_fork_teacher:
                        # round 5: the teacher's forward on a stream FORKED from the capture stream and joined before the loss -- in the
                        # replayed graph the two networks' forwards are independent branches and run concurrently (the MiT student's
                        # kernels are short and fill a fraction of the chip each)
                        cap = torch.cuda.current_stream(x.device)
                        fork = _shared_stream(x.device, "capture_fork")
                        fork.wait_stream(cap)
                        with torch.cuda.stream(fork), torch.no_grad():
                            t_lr = self.teacher(g["cat"])[2]
                        s_lr = self.student(g["cat"])[2]
                        cap.wait_stream(fork)
                    else:
                        with torch.no_grad():
                            t_lr = self.teacher(g["cat"])[2]
                        s_lr = self.student(g["cat"])[2]
                    total, ce, di = L.upsample_ce_distill(s_lr, t_lr, g["labels"], lambda_seg, lambda_distil, self.distill_scale)
                    if cfg.graph_fork_wgrad and cfg.wgrad_stream:
                        # weight gradients as forked branches of the captured graph
                        with _config.use(cfg.replace(teacher_stream=False, wgrad_stream=True)):
                            _lib.side_overlap = True
                            try:
                                total.backward()
                            finally:
                                _lib.side_overlap = False
                                _lib.join_side()
                    else:
                        total.backward()
            # the captured backward writes into THESE gradient tensors (memory of the graph's pool) on every replay
            g.update(graph=graph, out=(total.detach(), ce, di),
                     grads=[(p, p.grad) for p in self.student.parameters() if p.requires_grad])
        else:
            B = x.shape[0]
            g["cat"][:B].copy_(x)
            g["cat"][B:].copy_(mix)
            g["labels"].copy_(labels)
        for p, captured in g["grads"]:
            # an eager step on the same trainer (selftrain_step) or an external zero_grad() re-binds p.grad; the replay would then
            # fill orphaned buffers while the optimizer read stale ones -- point the parameters back at the captured tensors
            if p.grad is not captured:
                p.grad = captured
        g["graph"].replay()
        self.reducer.reduce()
        self._opt_step()
        total, ce, di = g["out"]
        # (clones: the graph's output tensors are overwritten by the next replay)
        return {"total": total.clone(), "ce": ce.clone(), "distil": di.clone()}

    # ------------------------------------------------------------------ warm-up step
    def warmup_step(self, it, x, x_aug, rec_s2t, labels, lambda_seg=1.0, lambda_distil=0.5):
        """x, x_aug, rec_s2t [B,3,H,W]; labels [B,H,W] int64.  Returns device scalars (no sync)."""
        with _config.use(self._cfg()):
            if self.use_graph and x.is_cuda:
                return self._warmup_step_graphed(it, x, x_aug, rec_s2t, labels, lambda_seg, lambda_distil)
            return self._warmup_step_eager(it, x, x_aug, rec_s2t, labels, lambda_seg, lambda_distil)

    def _warmup_step_eager(self, it, x, x_aug, rec_s2t, labels, lambda_seg=1.0, lambda_distil=0.5):
        self._begin(it)
        with torch.no_grad():
            mix, _ = U.classmix(rec_s2t, x_aug, labels, self.rng, present=self._present(labels))
            cat = torch.cat([x, mix])
        (_, _, s_lr, _), pending = self._student_and_teacher(cat, cat)
        (t_lr, _), = self._teacher_join(pending)
        total, ce, di = L.upsample_ce_distill(s_lr, t_lr, labels, lambda_seg, lambda_distil, self.distill_scale)
        self._finish(total)
        return {"total": total.detach(), "ce": ce, "distil": di}

    # ------------------------------------------------------------------ self-training step
    def selftrain_step(self, it, x, x_aug, rec_s2t, labels, t_img, t_aug, pseudo_prob, class_features,
                       lambda_seg=1.0, lambda_distil=0.25):
        """Adds target images `t_img`, their augmented view and offline pseudo-labels; `class_features`
        is a diga_amd.calc_centroids.Class_Features."""
        with _config.use(self._cfg()):
            return self._selftrain_step(it, x, x_aug, rec_s2t, labels, t_img, t_aug, pseudo_prob, class_features, lambda_seg, lambda_distil)

    def _selftrain_step(self, it, x, x_aug, rec_s2t, labels, t_img, t_aug, pseudo_prob, class_features, lambda_seg, lambda_distil):
        self._begin(it)
        B = x.shape[0]
        with torch.no_grad():
            present = self._present(labels)
            if present is None:
                present = U.classmix_present(labels)          # both ClassMix blocks of the step draw from the same label lists
            mix, _ = U.classmix(rec_s2t, x_aug, labels, self.rng, present=present)
            cat = torch.cat([x, mix])
        (_, _, s_lr, _), pending = self._student_and_teacher(cat, cat, t_img)
        # (gloo is the smoke-test backend: its host-synchronous GPU collectives next to extra streams degenerate to seconds per step --
        #  ddp.init_from_env switches the side streams off for it, and the overlapped forms stay off too unless a test asks for them)
        cfg = _config.active()
        overlap = cfg.c4_overlap
        if self.world > 1 and torch.distributed.get_backend() == "gloo" and not cfg.c4_overlap_gloo:
            overlap = 0
        if (x.is_cuda and overlap >= 2 and getattr(self, "_pending_join", False)
                and len(getattr(self, "_teacher_events", ())) == 2):
            return self._selftrain_rest_overlapped(s_lr, pending, x, labels, t_aug, pseudo_prob, class_features, present, B,
                                                   lambda_seg, lambda_distil)
        (t_lr, t_feat), (tt_lr, tt_feat) = self._teacher_join(pending)
        with torch.no_grad():
            # bilateral consensus: keep the offline pseudo-label where the centroid label agrees
            pseudo = class_features.consensus_pseudo_labels(tt_feat, pseudo_prob)
            cross_mix, cross_lab, _ = U.classmix(t_aug, x, labels, self.rng, bg_labels=pseudo, present=present)
            # centroid EMA: target (filtered pseudo-labels) first, then source (teacher feats of the mixed view)
            for feat, out, lab in ((tt_feat, tt_lr, pseudo), (t_feat[B:], t_lr[B:], labels)):
                sums, counts = class_features._class_sums(feat, out, labels_full=lab)[:2]
                if self.centroid_exchange == "allreduce":
                    ms, nv = ddp.allreduce_class_means(sums, counts, class_features.min_pixels)
                    cents, nums = class_features._state_on(sums.device)
                    ddp.apply_mean_of_vectors(cents, nums, ms, nv, class_features.centroid_momentum)
                    continue
                sums, counts = ddp.gather_class_sums(sums, counts)
                class_features._apply(sums, counts, feat.shape[-2] * feat.shape[-1], class_features.min_pixels, 0)
        if x.is_cuda and overlap >= 1:
            return self._selftrain_tail_overlapped(s_lr, t_lr, labels, cross_mix, cross_lab, lambda_seg, lambda_distil)
        _, _, c_lr, _ = self.student(cross_mix)
        total_s, ce, di = L.upsample_ce_distill(s_lr, t_lr, labels, lambda_seg, lambda_distil, self.distill_scale)
        ce_mix = L.upsample_ce(c_lr, cross_lab, lambda_seg)
        total = total_s + ce_mix
        self._finish(total)
        return {"total": total.detach(), "ce": ce, "distil": di, "ce_mix": ce_mix.detach()}

    def _selftrain_rest_overlapped(self, s_lr, pending, x, labels, t_aug, pseudo_prob, class_features, present, B, lambda_seg, lambda_distil):
        """config.c4_overlap = 2 (default): everything behind the student(cat) / teacher(cat) forwards as TWO concurrent branches.
        Main stream: wait for the teacher's pass over `cat` only (an event between its two passes), form CE + distill and start the
        backward pass of the student(cat) graph (weight gradients on their side stream).  Third stream: wait for the teacher's pass
        over the target images, then the consensus filter, ClassMix #2, the two centroid updates, the forward of student(cross_mix),
        CE_mix and its backward (torch.autograd.grad) -- which used to run with the GPU to themselves, one stream, between the
        forwards and the one backward pass.  Host order of every ClassMix draw, BatchNorm running-statistics update
        (student(cat) before student(cross_mix): an event) and centroid update is the reference's; the gradient of a shared weight is
        g_cat + g_cross either way: bit-identical to the one-backward form (tests/test_selftrain.py)."""
        _accumulate_on_other_streams_is_intended()
        dev = s_lr.device
        main = torch.cuda.current_stream(dev)
        sb = _shared_stream(dev, "cross")
        (t_lr, t_feat), (tt_lr, tt_feat) = pending
        ev_cat, ev_tgt = self._teacher_events
        main.wait_event(ev_cat)
        for t in (t_lr, t_feat):
            t.record_stream(main)
        total_s, ce, di = L.upsample_ce_distill(s_lr, t_lr, labels, lambda_seg, lambda_distil, self.distill_scale)
        ev_main = torch.cuda.Event()
        ev_main.record(main)                                    # student(cat)'s forward (its BatchNorm statistics updates) is enqueued
        params = [p for p in self.student.parameters() if p.requires_grad]
        self.opt.zero_grad(set_to_none=True)
        _lib.side_overlap = True
        try:
            with self.reducer.hold():                           # (N > 1: these are partial sums -- the buckets leave after the join)
                total_s.backward()                              # main stream; weight gradients on the side stream
            sb.wait_event(ev_tgt)
            sb.wait_event(ev_main)
            for t in (t_lr, t_feat, tt_lr, tt_feat):
                t.record_stream(sb)
            with torch.cuda.stream(sb):
                with torch.no_grad():
                    pseudo = class_features.consensus_pseudo_labels(tt_feat, pseudo_prob)
                    cross_mix, cross_lab, _ = U.classmix(t_aug, x, labels, self.rng, bg_labels=pseudo, present=present)
                    for feat, out, lab in ((tt_feat, tt_lr, pseudo), (t_feat[B:], t_lr[B:], labels)):
                        sums, counts = class_features._class_sums(feat, out, labels_full=lab)[:2]
                        if self.centroid_exchange == "allreduce":
                            ms, nv = ddp.allreduce_class_means(sums, counts, class_features.min_pixels)
                            cents, nums = class_features._state_on(sums.device)
                            ddp.apply_mean_of_vectors(cents, nums, ms, nv, class_features.centroid_momentum)
                            continue
                        sums, counts = ddp.gather_class_sums(sums, counts)
                        class_features._apply(sums, counts, feat.shape[-2] * feat.shape[-1], class_features.min_pixels, 0)
                # (N > 1: this graph is built AFTER the first one's backward ran; its convolutions see that their bucket slices already
                #  ARE their parameters' gradients and write into tensors of their own -- model/conv.py, `held`)
                _, _, c_lr, _ = self.student(cross_mix)
                ce_mix = L.upsample_ce(c_lr, cross_lab, lambda_seg)
                g2 = torch.autograd.grad(ce_mix, params, allow_unused=True)
        finally:
            _lib.side_overlap = False
            _lib.join_side()
        self._pending_join = False
        main.wait_stream(sb)
        main.wait_stream(self._side)
        return self._join_gradients(params, g2, total_s, ce, di, ce_mix, main)

    def _join_gradients(self, params, g2, total_s, ce, di, ce_mix, main):
        second = {}
        for p, g in zip(params, g2):
            if g is not None:
                g.record_stream(main)
                second[id(p)] = g

        def add(group):
            pg, gg = [], []
            for p in group:
                g = second.get(id(p))
                if g is None:
                    continue
                if p.grad is None:
                    p.grad = g
                else:
                    pg.append(p.grad)
                    gg.append(g)
            if pg:
                torch._foreach_add_(pg, gg)

        if self.reducer.active:
            # N > 1: bucket by bucket, in the order the buckets leave -- a bucket's all-reduce starts as soon as ITS sum exists and
            # travels under the adds of the buckets behind it (round 5 added everything, then sent all 260 MB at once)
            in_buckets = set()
            for i, bucket in enumerate(self.reducer.buckets):
                add(bucket)
                in_buckets.update(id(p) for p in bucket)
                self.reducer.launch_bucket(i)
            add([p for p in params if id(p) not in in_buckets])
        else:
            add(params)
        ce_mix_d = ce_mix.detach()
        ce_mix_d.record_stream(main)
        total = total_s.detach() + ce_mix_d
        self.reducer.reduce()
        self._opt_step()
        return {"total": total, "ce": ce, "distil": di, "ce_mix": ce_mix_d}

    def _selftrain_tail_overlapped(self, s_lr, t_lr, labels, cross_mix, cross_lab, lambda_seg, lambda_distil):
        """The last third of the self-training step with its two student graphs on two streams (round 5, one process per GPU only):
        the reference forms total = CE + distill (graph of student(cat)) + CE_mix (graph of student(cross_mix)) and calls one
        backward; the cross-mixed forward can only start after the teacher's target pass, the consensus filter and ClassMix #2, so in
        the one-backward form it runs ALONE on the GPU (single stream: every bandwidth pass exposed) before the backward pass starts.
        Here the backward of the first graph starts as soon as its loss exists, on the main stream (+ the weight-gradient side
        stream), while the cross-mixed forward and then ITS backward (torch.autograd.grad: gradients in tensors of their own) run on a
        third stream; one multi-tensor add joins the two gradient sets.  Same terms: grad = g_cat + g_cross, a two-term sum either way
        (fp addition commutes), every kernel and its arithmetic unchanged -- bit-identical to the one-backward form
        (tests/test_selftrain.py::test_selftrain_overlapped_tail_is_bit_identical).  config.c4_overlap = 0 switches it off."""
        _accumulate_on_other_streams_is_intended()
        dev = s_lr.device
        main = torch.cuda.current_stream(dev)
        sb = _shared_stream(dev, "cross")
        total_s, ce, di = L.upsample_ce_distill(s_lr, t_lr, labels, lambda_seg, lambda_distil, self.distill_scale)
        sb.wait_stream(main)                                    # cross_mix / cross_lab (ClassMix #2) and the EMA'd state are ready
        with torch.cuda.stream(sb):
            _, _, c_lr, _ = self.student(cross_mix)
            ce_mix = L.upsample_ce(c_lr, cross_lab, lambda_seg)
        for t in (cross_mix, cross_lab):
            t.record_stream(sb)
        params = [p for p in self.student.parameters() if p.requires_grad]
        self.opt.zero_grad(set_to_none=True)
        _lib.side_overlap = True
        try:
            with self.reducer.hold():                           # (N > 1: partial sums -- the buckets leave after the join)
                total_s.backward()                              # main stream; weight gradients on the side stream
            with torch.cuda.stream(sb):
                g2 = torch.autograd.grad(ce_mix, params, allow_unused=True)
        finally:
            _lib.side_overlap = False
            _lib.join_side()
        main.wait_stream(sb)
        return self._join_gradients(params, g2, total_s, ce, di, ce_mix, main)
