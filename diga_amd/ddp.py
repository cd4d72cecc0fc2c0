"""Data-parallel exchange steps of the DiGA hot path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference is single-process, so this layer is new functionality:
  * GradReducer  -- sum of the student's gradients over ranks in a few large flat buckets (xGMI is
    point-to-point: few big ring transfers beat many small ones); the 1/world averaging is folded into
    the fused SGD kernel's grad_scale, so no extra pass touches the gradients.
  * gather_class_sums -- all-gather of the per-image class sums/counts of the centroid update so that
    every rank applies the order-dependent EMA in the global (rank-major, image-major, class-minor)
    order: bit-identical to one process that saw the concatenated batch.
BatchNorm statistics stay local and the EMA teacher is recomputed on every rank from the (identical)
all-reduced student, exactly as a per-rank run of the reference would.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from RANK/WORLD_SIZE/MASTER_* (torchrun contract).
    Returns (rank, world, local_rank).  World size 1 needs no process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("DIGA_DDP_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            # one process per GPU; with fewer devices than ranks (single-GPU smoke test over gloo) ranks share a device
            local = local % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
        if backend == "gloo" and torch.cuda.is_available():
            # gloo reduces GPU tensors through host-synchronous copies on its own streams; next to the step driver's
            # side streams (teacher forward, weight gradients) that degenerates to seconds per step on this stack
            # (measured: 2 ranks on one GPU 15 s vs 0.13 s).  gloo is the smoke-test backend only -- run it on one
            # stream.  The RCCL path is stream-ordered and keeps the side streams (tools/nccl_1rank_proxy.py).
            os.environ["DIGA_TEACHER_STREAM"] = "0"
            os.environ["DIGA_WGRAD_STREAM"] = "0"
    return rank, world, local


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


class GradReducer:
    """Bucketed all-reduce(sum) of the gradients of `params` (unique tensors, reverse order so the
    buckets of the last layers -- whose gradients exist first -- go out first).

    With more than one rank every parameter carries a post-accumulate-grad hook: the moment the last
    gradient of a bucket exists (inside backward) the bucket is packed by one multi-tensor copy and its
    all-reduce is started asynchronously on RCCL's stream, so all but the last bucket travel under the
    rest of the backward pass; `reduce()` starts whatever is still pending (gradients that were set by
    hand), waits, and unpacks with one multi-tensor copy per bucket."""

    def __init__(self, params, bucket_bytes=128 << 20, group=None, overlap=True):
        self.group = group
        seen, uniq = set(), []
        for p in params:
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                uniq.append(p)
        self.params = list(reversed(uniq))
        self.buckets, cur, cur_bytes = [], [], 0
        for p in self.params:
            nb = p.numel() * p.element_size()
            if cur and cur_bytes + nb > bucket_bytes:
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nb
        if cur:
            self.buckets.append(cur)
        self._flat = [None] * len(self.buckets)
        self._work = [None] * len(self.buckets)
        self._ready = [0] * len(self.buckets)
        self._hooks = []
        if overlap and world_size() > 1 and os.environ.get("DIGA_DDP_OVERLAP", "1") != "0":
            for i, bucket in enumerate(self.buckets):
                for p in bucket:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))

    def _make_hook(self, i):
        def hook(_param):
            self._ready[i] += 1
            if self._ready[i] > len(self.buckets[i]):
                # a second backward() accumulated into p.grad after the bucket was packed and sent: its contribution
                # would be overwritten by reduce().  Gradient accumulation needs the hooks off (overlap=False).
                raise RuntimeError("GradReducer: a parameter's gradient was accumulated again after its bucket had been "
                                   "launched (two backward passes before reduce()); build the reducer with "
                                   "overlap=False for gradient accumulation")
            if self._ready[i] == len(self.buckets[i]) and self._work[i] is None:
                self._launch(i)
        return hook

    def _views(self, i):
        flat, views, off = self._flat[i], [], 0
        for p in self.buckets[i]:
            views.append(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        return views

    def _launch(self, i):
        grads = [p.grad for p in self.buckets[i]]
        if any(g is None for g in grads):
            raise RuntimeError("GradReducer: a parameter has no gradient")
        n = sum(g.numel() for g in grads)
        flat = self._flat[i]
        if flat is None or flat.numel() != n or flat.device != grads[0].device:
            self._flat[i] = torch.empty(n, dtype=grads[0].dtype, device=grads[0].device)
        side = None
        if grads[0].is_cuda:
            from diga_amd import _lib
            side = _lib.active_side_stream(grads[0].device)
        if side is not None:
            # Weight gradients of this bucket may still be running on the side stream.  Pack and launch FROM the side
            # stream (it first waits for the main stream's work so far: bias / GroupNorm gradients live there): RCCL's
            # stream then orders itself behind the side stream only, and the main stream -- the backward-data /
            # BatchNorm chain -- is never made to wait for a weight gradient inside backward.
            side.wait_stream(torch.cuda.current_stream(grads[0].device))
            with torch.cuda.stream(side):
                torch._foreach_copy_(self._views(i), [g.detach() for g in grads])
                self._work[i] = dist.all_reduce(self._flat[i], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            for g in grads:
                g.record_stream(side)
        else:
            torch._foreach_copy_(self._views(i), [g.detach() for g in grads])
            self._work[i] = dist.all_reduce(self._flat[i], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def reduce(self):
        """Sum gradients over all ranks in place.  No-op for world size 1."""
        if world_size() == 1:
            return
        for i in range(len(self.buckets)):
            if self._work[i] is None:
                self._launch(i)
        for i, bucket in enumerate(self.buckets):
            self._work[i].wait()
            torch._foreach_copy_([p.grad for p in bucket], self._views(i))
            self._work[i] = None
            self._ready[i] = 0


def gather_class_sums(sums, counts, group=None):
    """[N,K,D] sums and [N,K] counts of every rank, concatenated rank-major along the image axis."""
    w = world_size()
    if w == 1:
        return sums, counts
    s_all = [torch.empty_like(sums) for _ in range(w)]
    c_all = [torch.empty_like(counts) for _ in range(w)]
    dist.all_gather(s_all, sums.contiguous(), group=group)
    dist.all_gather(c_all, counts.contiguous(), group=group)
    return torch.cat(s_all, 0), torch.cat(c_all, 0)


def broadcast_module(module, src=0, group=None):
    """Make parameters and buffers of every rank equal to rank `src`'s (start-of-training sync)."""
    if world_size() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)
