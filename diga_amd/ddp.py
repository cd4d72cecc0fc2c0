"""Data-parallel exchange steps of the DiGA hot path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference is single-process, so this layer is new functionality:
  * GradReducer  -- sum of the student's gradients over ranks in a few large flat buckets (xGMI is
    point-to-point: few big ring transfers beat many small ones); the 1/world averaging is folded into
    the fused SGD kernel's grad_scale, so no extra pass touches the gradients.
  * gather_class_sums -- all-gather of the per-image class sums/counts of the centroid update so that
    every rank applies the order-dependent EMA in the global (rank-major, image-major, class-minor)
    order: bit-identical to one process that saw the concatenated batch.
  * allreduce_class_means / apply_mean_of_vectors -- the approximate "centroid all-reduce" of BASELINE
    configs[3]: [19,256] sums of class means + [19] vector counts in one all-reduce.
BatchNorm statistics stay local and the EMA teacher is recomputed on every rank from the (identical)
all-reduced student, exactly as a per-rank run of the reference would.
"""
import os

import torch
import torch.distributed as dist

from diga_amd import config


def init_from_env(backend=None, cfg=None, single_rank_group=False):
    """Initialise the default process group from RANK/WORLD_SIZE/MASTER_* (torchrun contract).
    Returns (rank, world, local_rank).  World size 1 needs no process group (single_rank_group=True creates one anyway: the
    1-rank RCCL test).  Under the gloo smoke-test backend on a GPU the side streams are switched off IN `cfg` (default: the
    process defaults, diga_amd.config.DEFAULTS) -- a field write, not an environment variable."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or single_rank_group) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("DIGA_DDP_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            ndev = max(torch.cuda.device_count(), 1)
            # LOCAL_WORLD_SIZE is set by torchrun; srun / mpirun / custom spawners of a multi-node job often set only RANK,
            # LOCAL_RANK and WORLD_SIZE -- then LOCAL_RANK is trusted (world 16 on 2 x 8 GPUs is fine) and only validated
            local_world = os.environ.get("LOCAL_WORLD_SIZE")
            if local_world is not None and int(local_world) > ndev:
                if backend == "nccl":
                    raise RuntimeError(f"{local_world} ranks on {ndev} GPU(s): RCCL needs one device per rank "
                                       "(only the gloo smoke configuration may share a device)")
                local = local % ndev          # single-GPU smoke test over gloo: ranks share the device
            elif local_world is None and backend != "nccl" and local >= ndev:
                local = local % ndev          # (the same smoke configuration started without LOCAL_WORLD_SIZE)
            elif not 0 <= local < ndev:
                # the production layout: ONE process per GPU, device index = LOCAL_RANK, nothing shared
                raise RuntimeError(f"LOCAL_RANK={local} but this node has {ndev} GPU(s)")
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
        if dist.get_world_size() != world or dist.get_rank() != rank:
            raise RuntimeError(f"process group came up as rank {dist.get_rank()}/{dist.get_world_size()}, "
                               f"environment says {rank}/{world}")
        if backend == "gloo" and torch.cuda.is_available():
            # gloo reduces GPU tensors through host-synchronous copies on its own streams; next to the step driver's
            # side streams (teacher forward, weight gradients) that degenerates to seconds per step on this stack
            # (measured: 2 ranks on one GPU 15 s vs 0.13 s).  gloo is the smoke-test backend only -- run it on one
            # stream.  The RCCL path is stream-ordered and keeps the side streams (tools/nccl_1rank_proxy.py).
            cfg = cfg if cfg is not None else config.DEFAULTS
            cfg.teacher_stream = False
            cfg.wgrad_stream = False
    return rank, world, local


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def reducer_active():
    """Do gradients travel?  More than one rank -- or ONE rank of an initialised group with config.ddp_single_rank (the 1-rank
    RCCL test: hooks, bucket views and collectives run exactly as with N ranks, the sum over one rank is the identity)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or config.active().ddp_single_rank


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


class GradReducer:
    """Bucketed all-reduce(sum) of the gradients of `params` (unique tensors, reverse order so the
    buckets of the last layers -- whose gradients exist first -- go out first).

    Every bucket is ONE flat buffer, allocated once (first use) and kept for the life of the reducer, and -- with more than one
    rank and `as_views` -- the gradients LIVE in it: `grad_view(p)` hands out a fresh tensor of p's shape and strides over p's
    slice of its bucket.  A producer that writes its gradient there (DigaConv2d's weight-gradient kernels do: the parameter
    carries the callable as `p._diga_grad_view`) has packed it; a gradient that arrives in a tensor of its own (biases, GroupNorm
    affines, a weight used twice in one graph) is copied into its slice when the bucket is launched and `p.grad` is re-pointed at
    the slice, so nothing is unpacked after the all-reduce: RCCL reduces the bucket in place and the optimizer reads p.grad =
    the bucket.  Round 4 packed AND unpacked every bucket with multi-tensor copies (2 x 260 MB read + written per step).

    Every parameter carries a post-accumulate-grad hook: the moment the last gradient of a bucket exists (inside backward) its
    all-reduce is started asynchronously on RCCL's stream, so all but the last bucket travel under the rest of the backward
    pass; `reduce()` starts whatever is still pending (gradients that were set by hand) and waits.
    `as_views=False` (the HIP-graph step: the captured backward writes into ITS static gradient tensors, which must stay
    p.grad): pack, all-reduce, copy back."""

    def __init__(self, params, bucket_bytes=None, group=None, overlap=True, as_views=None):
        self.group = group
        if bucket_bytes is None:
            # ~25 MB buckets: 260 MB of ResNet-101 gradients leave in ~11 ring all-reduces while backward is
            # still running and only the last one (stem .. first layer-1 blocks, produced at the very end) is exposed; with
            # the 128 MB buckets of round 2 the exposed tail was a third of the payload.  Per-link xGMI time of a 25 MB
            # ring all-reduce on 8 GPUs: 2*(7/8)*25 MB / ~50 GB/s effective ~ 0.9 ms -- far above launch latency.
            bucket_bytes = int(config.active().ddp_bucket_mb) << 20
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
        self._copy_back = [None] * len(self.buckets)
        self._hooks = []
        self._held = 0
        self._where = {}                      # id(param) -> (bucket, element offset)
        for i, bucket in enumerate(self.buckets):
            off = 0
            for p in bucket:
                self._where[id(p)] = (i, off)
                off += p.numel()
        active = self.active = reducer_active()
        self.as_views = (overlap if as_views is None else bool(as_views)) and active and config.active().ddp_grad_views
        if overlap and active and config.active().ddp_overlap:
            for i, bucket in enumerate(self.buckets):
                for p in bucket:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
        if self.as_views:
            for p in self.params:
                if p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last)):
                    p._diga_grad_view = self._make_view_fn(p)

    def close(self):
        """Detach from the parameters (hooks, view callables): a second reducer / trainer on the same parameters starts clean."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for p in self.params:
            if getattr(p, "_diga_grad_view", None) is not None:
                del p._diga_grad_view

    def hold(self):
        """Context: gradients produced inside are NOT counted or sent by the hooks (the self-training step's two-graph form: the
        first backward() leaves partial sums in p.grad, the second graph's gradients are added by hand, then `reduce()` sends every
        bucket at once)."""
        reducer = self

        class _Hold:
            def __enter__(self):
                reducer._held += 1

            def __exit__(self, *exc):
                reducer._held -= 1
                return False
        return _Hold()

    def _make_hook(self, i):
        def hook(_param):
            if self._held:
                return
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

    def _bucket(self, i, like):
        flat = self._flat[i]
        if flat is None or flat.device != like.device or flat.dtype != like.dtype:
            n = sum(p.numel() for p in self.buckets[i])
            flat = self._flat[i] = torch.empty(n, dtype=like.dtype, device=like.device)
        return flat

    def _slice(self, p):
        i, off = self._where[id(p)]
        return self._bucket(i, p)[off:off + p.numel()]

    def grad_view(self, p):
        """A FRESH tensor of p's shape and strides over p's slice of its bucket (fresh: autograd's AccumulateGrad takes a
        gradient over without a copy only when nothing else holds the tensor object)."""
        return self._slice(p).as_strided(p.shape, p.stride())

    def _make_view_fn(self, p):
        def view():
            return self.grad_view(p)
        return view

    def _launch(self, i):
        bucket = self.buckets[i]
        grads = [p.grad for p in bucket]
        if any(g is None for g in grads):
            raise RuntimeError("GradReducer: a parameter has no gradient")
        flat = self._bucket(i, grads[0])
        esz = flat.element_size()
        dst, src, back = [], [], []
        for p, g in zip(bucket, grads):
            _, off = self._where[id(p)]
            if g.data_ptr() == flat.data_ptr() + off * esz and g.stride() == p.stride():
                continue                                    # produced in place (or still there from the re-pointing below)
            if self.as_views and g.stride() == p.stride():
                v = self.grad_view(p)
                dst.append(v)
                src.append(g.detach())
                p.grad = v                                  # no copy back: the optimizer reads the bucket
            else:
                v = flat[off:off + p.numel()].view(g.shape) if g.is_contiguous() else None
                if v is None:
                    raise RuntimeError("GradReducer: a gradient that is neither dense nor laid out like its parameter")
                dst.append(v)
                src.append(g.detach())
                back.append((g, v))
        self._copy_back[i] = back
        side = None
        if flat.is_cuda:
            from diga_amd import _lib
            side = _lib.active_side_stream(flat.device)
        if side is not None:
            # Weight gradients of this bucket may still be running on the side stream.  Pack and launch FROM the side
            # stream (it first waits for the main stream's work so far: bias / GroupNorm gradients live there): RCCL's
            # stream then orders itself behind the side stream only, and the main stream -- the backward-data /
            # BatchNorm chain -- is never made to wait for a weight gradient inside backward.
            side.wait_stream(torch.cuda.current_stream(flat.device))
            with torch.cuda.stream(side):
                if dst:
                    torch._foreach_copy_(dst, src)
                self._work[i] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            for g in src:
                g.record_stream(side)
        else:
            if dst:
                torch._foreach_copy_(dst, src)
            self._work[i] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def launch_bucket(self, i):
        """Start bucket i's all-reduce now (its gradients are final): the overlapped self-training forms add the second graph's
        gradients bucket by bucket, in the order the buckets were formed, and send each bucket as soon as its sum exists instead of
        all of them after the last add."""
        if self.active and self._work[i] is None:
            self._launch(i)

    def reduce(self):
        """Sum gradients over all ranks in place.  No-op for world size 1 (unless config.ddp_single_rank built the reducer active)."""
        if not self.active:
            return
        for i in range(len(self.buckets)):
            if self._work[i] is None:
                self._launch(i)
        for i in range(len(self.buckets)):
            self._work[i].wait()
            back = self._copy_back[i]
            if back:
                torch._foreach_copy_([g for g, _ in back], [v for _, v in back])
            self._copy_back[i] = None
            self._work[i] = None
            self._ready[i] = 0


def gather_class_sums(sums, counts, group=None):
    """[N,K,D] sums and [N,K] counts of every rank, concatenated rank-major along the image axis: the exact exchange
    (every rank then applies the order-dependent EMA in global order, bit-identical to one process on the concatenated
    batch).  ONE collective per call: sums and counts travel packed as [N,K,D+1] fp32 (a count is at most h*w < 2^24,
    exact in fp32)."""
    w = world_size()
    if w == 1 and not reducer_active():
        return sums, counts
    n, k, d = sums.shape
    packed = torch.empty((n, k, d + 1), dtype=torch.float32, device=sums.device)
    packed[..., :d] = sums
    packed[..., d] = counts.to(torch.float32)
    out = torch.empty((w * n, k, d + 1), dtype=torch.float32, device=sums.device)
    dist.all_gather_into_tensor(out, packed, group=group)
    return out[..., :d].contiguous(), out[..., d].round().to(counts.dtype)


def allreduce_class_means(sums, counts, min_pixels, group=None):
    """The cheaper exchange BASELINE configs[3] names ("centroid all-reduce"): every rank folds its images'
    valid class means (count >= min_pixels and a non-zero vector, the reference's skip rules, calc_centroids.py:131,148)
    into [K,D] sums of means + [K] numbers of vectors, ONE all-reduce of the packed [K,D+1] buffer, and the centroid
    bank takes the n_k vectors of a class in one closed-form step (`apply_mean_of_vectors`).  Approximate: the reference
    applies the vectors one by one (calc_centroids.py:147-164); replacing them by their mean changes the result by
    O(momentum^2 * n_k) -- measured in tests/test_host_cpu.py.  Returns (mean_sums [K,D], n_vectors [K]) of ALL ranks."""
    n, k, d = sums.shape
    cnt = counts.to(torch.float32)
    means = sums / cnt.clamp_min(1.0)[..., None]
    valid = (counts >= max(int(min_pixels), 1)) & (means.sum(-1) != 0)
    packed = torch.empty((k, d + 1), dtype=torch.float32, device=sums.device)
    packed[:, :d] = (means * valid[..., None]).sum(0)
    packed[:, d] = valid.sum(0).to(torch.float32)
    if reducer_active():
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
    return packed[:, :d].contiguous(), packed[:, d].contiguous()


def apply_mean_of_vectors(cents, nums, mean_sums, n_vectors, momentum, cap=3000.0):
    """n_k EMA steps of class k towards the MEAN of its n_k vectors, in closed form and in place:
    c_k <- (1-m)^n_k c_k + (1 - (1-m)^n_k) * mean_k ; num_k <- min(num_k + n_k, cap)  (calc_centroids.py:147-164 with
    every vector replaced by the class mean).  A few [19,256] torch ops on the device, no host sync."""
    keep = torch.pow(torch.full_like(n_vectors, 1.0 - float(momentum)), n_vectors)
    mean = mean_sums / n_vectors.clamp_min(1.0)[:, None]
    has = (n_vectors > 0)[:, None]
    cents.copy_(torch.where(has, keep[:, None] * cents + (1.0 - keep)[:, None] * mean, cents))
    nums.copy_(torch.clamp(nums + n_vectors.to(nums.dtype), max=cap))


def broadcast_module(module, src=0, group=None):
    """Make parameters and buffers of every rank equal to rank `src`'s (start-of-training sync)."""
    if world_size() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)
