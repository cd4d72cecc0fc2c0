"""Explicit configuration of the step driver and of the layers' path choices (round 6).

Until round 5 the stream policy and a dozen path switches were steered through `os.environ` AT RUN TIME (the trainer set and
popped DIGA_TEACHER_STREAM / DIGA_WGRAD_STREAM around a graph capture, ddp.init_from_env wrote them for gloo, bench.py flipped
them around timed code): process-global, not thread-safe, invisible in a trainer's signature.  Now:

  * `StepConfig` holds every switch as a typed field;
  * environment variables are read ONCE, at import, into `DEFAULTS` (so `DIGA_*=... python train.py` keeps working as a way to
    set defaults) -- nothing in the package reads or writes `os.environ` after that (ddp.init_from_env's MASTER_* defaults aside);
  * `DigaTrainer(config=...)` owns a config and runs every step under `use(cfg)`; the layers (DigaConv2d, the norm layers, the
    bottleneck wiring, GradReducer) read `active()` at call time;
  * tests and A/B runs change fields (`with override(c4_overlap=0): ...`, `dataclasses.replace(DEFAULTS, ...)`) instead of the
    environment.

A config is plain data: copying one (`dataclasses.replace`) never touches a trainer that holds the original.
"""
import contextlib
import dataclasses
import os
from dataclasses import dataclass


def _flag(name, default):
    v = os.environ.get(name)
    if v is None:
        return default
    return v not in ("0", "", "false", "False")


@dataclass
class StepConfig:
    # ---- streams of the step driver (diga_amd/train_step.py)
    teacher_stream: bool = True          # teacher forward(s) on a second HIP stream next to the student's forward
    teacher_offset: str = "layer1"       # the student stage behind which the teacher is enqueued ("" / "0": both at once)
    wgrad_stream: bool = True            # weight gradients on a side stream next to the backward-data / BatchNorm chain
    wgrad_hold: int = 4                  # ... whose inputs are held for this many layers and released behind an event (0: record_stream)
    wgrad_hold_batch: int = 4            # ... that many at a time (one device-side wait per batch)
    c4_overlap: int = 2                  # self-training step: 0 one backward pass; 1 cross-mixed fwd/bwd on a third stream; 2 the whole target branch
    c4_overlap_gloo: bool = False        # keep the overlapped forms under the gloo smoke-test backend (slow there, not wrong)
    step_graph: bool = False             # replay the static part of the warm-up step from a HIP graph (launch-bound configurations)
    graph_fork_teacher: bool = True      # ... with the teacher's forward as a forked branch of the captured graph
    graph_fork_wgrad: bool = True        # ... and the weight gradients as forked branches
    centroid_exchange: str = "allgather"  # N > 1: "allgather" (exact) or "allreduce" (BASELINE configs[3]'s approximate exchange)
    # ---- convolution arithmetic and path choices (diga_amd/model/conv.py)
    conv_math: int = 0                   # 0 exact fp32 (matrix cores, Winograd for stride-1 3x3), 1 split bf16 (three bf16 MFMAs per product)
    winograd: bool = True
    winograd_ratio: float = 0.62
    winograd_default_max_tile: int = 6
    winograd_max_tile: int = 6           # 2 keeps every layer on F(2x2,3x3) (set_conv_math(0, exact=True))
    winograd_keep_v: bool = True         # keep the forward's input transform for the weight gradient
    winograd_stats: bool = True          # BatchNorm statistics from the Winograd output transform
    keep_v_max_gb: float = 8.0           # per layer
    keep_v_min_device_gb: float = 160.0
    conv_twin: str = "3"                 # split-bf16 twin kernels: "0" off, "1" every eligible conv, "3" multi-tap / shared inputs
    twin_only: bool = True               # producers write the split twin INSTEAD of the fp32 tensor where every reader takes twins
    twin_conv3: str = "1"
    # ---- norm layers / bottleneck wiring (diga_amd/model/norm.py, seg_model_noaux.py)
    relu_bits: bool = True               # ReLU masks as bit planes instead of reading the activated tensor in backward
    fuse_bwd: bool = True                # residual-junction add + ReLU mask + BN-backward reduce in the backward-data epilogue
    junction_chain: bool = True
    # ---- data parallelism (diga_amd/ddp.py)
    ddp_bucket_mb: int = 25
    ddp_grad_views: bool = True          # gradients live in the all-reduce buckets
    ddp_overlap: bool = True             # hook-driven bucket launches inside backward
    ddp_single_rank: bool = False        # run the reducer (hooks, views, collectives) even at world size 1 (the 1-rank RCCL test)
    # ---- MiT / SegFormer student
    mit_loss_scale: float = 1024.0

    @classmethod
    def from_env(cls):
        e = os.environ.get
        c = cls()
        c.teacher_stream = _flag("DIGA_TEACHER_STREAM", c.teacher_stream)
        c.teacher_offset = e("DIGA_TEACHER_OFFSET", c.teacher_offset)
        c.wgrad_stream = _flag("DIGA_WGRAD_STREAM", c.wgrad_stream)
        c.wgrad_hold = int(e("DIGA_WGRAD_HOLD", c.wgrad_hold))
        c.wgrad_hold_batch = int(e("DIGA_WGRAD_HOLD_BATCH", c.wgrad_hold_batch))
        c.c4_overlap = int(e("DIGA_C4_OVERLAP", c.c4_overlap))
        c.c4_overlap_gloo = _flag("DIGA_C4_OVERLAP_GLOO", c.c4_overlap_gloo)
        c.step_graph = _flag("DIGA_STEP_GRAPH", c.step_graph)
        c.graph_fork_teacher = _flag("DIGA_GRAPH_FORK_TEACHER", c.graph_fork_teacher)
        c.graph_fork_wgrad = _flag("DIGA_GRAPH_FORK_WGRAD", c.graph_fork_wgrad)
        c.centroid_exchange = e("DIGA_CENTROID_EXCHANGE", c.centroid_exchange)
        c.conv_math = 1 if e("DIGA_CONV_MATH", "") in ("bf16x3", "1") else 0
        c.winograd = _flag("DIGA_CONV_WINOGRAD", c.winograd)
        c.winograd_ratio = float(e("DIGA_CONV_WINOGRAD_RATIO", c.winograd_ratio))
        c.winograd_default_max_tile = c.winograd_max_tile = int(e("DIGA_CONV_WINOGRAD_TILE", c.winograd_default_max_tile))
        c.winograd_keep_v = _flag("DIGA_WINOGRAD_KEEP_V", c.winograd_keep_v)
        c.winograd_stats = _flag("DIGA_WINOGRAD_STATS", c.winograd_stats)
        c.keep_v_max_gb = float(e("DIGA_WINOGRAD_KEEP_V_MAX_GB", c.keep_v_max_gb))
        c.keep_v_min_device_gb = float(e("DIGA_WINOGRAD_KEEP_V_MIN_DEVICE_GB", c.keep_v_min_device_gb))
        c.conv_twin = e("DIGA_CONV_TWIN", c.conv_twin)
        c.twin_only = _flag("DIGA_TWIN_ONLY", c.twin_only)
        c.twin_conv3 = e("DIGA_TWIN_CONV3", c.twin_conv3)
        c.relu_bits = _flag("DIGA_RELU_BITS", c.relu_bits)
        c.fuse_bwd = _flag("DIGA_FUSE_BWD", c.fuse_bwd)
        c.junction_chain = _flag("DIGA_JUNCTION_CHAIN", c.junction_chain)
        c.ddp_bucket_mb = int(e("DIGA_DDP_BUCKET_MB", c.ddp_bucket_mb))
        c.ddp_grad_views = _flag("DIGA_DDP_GRAD_VIEWS", c.ddp_grad_views)
        c.ddp_overlap = _flag("DIGA_DDP_OVERLAP", c.ddp_overlap)
        c.ddp_single_rank = _flag("DIGA_DDP_SINGLE_RANK", c.ddp_single_rank)
        c.mit_loss_scale = float(e("DIGA_MIT_LOSS_SCALE", c.mit_loss_scale))
        c.validate()
        return c

    def validate(self):
        if self.centroid_exchange not in ("allgather", "allreduce"):
            raise ValueError(f"centroid_exchange must be 'allgather' or 'allreduce', not {self.centroid_exchange!r}")
        if self.c4_overlap not in (0, 1, 2):
            raise ValueError(f"c4_overlap must be 0, 1 or 2, not {self.c4_overlap!r}")
        if self.wgrad_hold < 0:
            raise ValueError(f"wgrad_hold must be >= 0, not {self.wgrad_hold!r}")
        if self.conv_math not in (0, 1):
            raise ValueError(f"conv_math must be 0 (fp32) or 1 (split bf16), not {self.conv_math!r}")
        if self.winograd_max_tile not in (2, 4, 6):
            raise ValueError(f"winograd_max_tile must be 2, 4 or 6, not {self.winograd_max_tile!r}")
        return self

    def replace(self, **fields):
        return dataclasses.replace(self, **fields).validate()

    def serial_streams(self):
        """A copy with every side stream off (kernel timing, graph capture on one stream, the gloo smoke-test backend)."""
        return self.replace(teacher_stream=False, wgrad_stream=False, c4_overlap=0)


DEFAULTS = StepConfig.from_env()       # the ONE read of the environment
_stack = [DEFAULTS]


def active():
    """The configuration in force: the innermost `use(...)`, else the process defaults."""
    return _stack[-1]


@contextlib.contextmanager
def use(cfg):
    """Run the body under `cfg` (DigaTrainer wraps every step in this).  Not re-entrant across threads by design: one process
    drives one GPU from one thread."""
    _stack.append(cfg)
    try:
        yield cfg
    finally:
        _stack.pop()


@contextlib.contextmanager
def override(**fields):
    """`with override(c4_overlap=0): ...` -- the active configuration with some fields replaced."""
    with use(active().replace(**fields)) as cfg:
        yield cfg
