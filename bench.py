#!/usr/bin/env python3
"""Benchmark of the DiGA training hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N = 1 runs in this process (headline + second arithmetic + the c4 / c5 / c1 legs + CPU baseline + bandwidth kernels + mIoU parity).
N > 1 measures the headline configuration and the second arithmetic on every rank (the legs are 1-GPU results: --multi-gpu-legs runs
them on all ranks too), no CPU-baseline child; rank 0 prints the same compact line.  N > 1: when the process was not started by torch.distributed.run (no WORLD_SIZE in
the environment) it starts N worker processes itself -- before anything touches the GPU, as children, never by
exec -- one per GPU, rendezvous on 127.0.0.1, backend "nccl" (= RCCL over xGMI); under
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` it is one of the ranks.

A step is one full DiGA warm-up iteration (EMA teacher update, ClassMix, student forward on 2B images, teacher
forward on 2B images, fused upsample+CE+distillation, backward, gradient all-reduce, fused SGD) on BASELINE.json
configs[1]: ResNet-101 DeepLabV2, B=8 source crops of 768x768 per GPU, synthetic inputs already resident in HBM.

Output: rank 0 prints ONE compact line of strict JSON (< 4000 bytes, self-checked by `emit`) as the last thing on stdout --
the contract's fields (`value` = source crops/s over all ranks), `roofline` of the dominant kernel family (forward
convolution; `achieved` / `frac` = FLOPs the matrix cores EXECUTE / summed launch time / peak, always <= 1; `frac_algorithmic`
= the direct-convolution FLOPs of SURVEY section 8d over the same time, > 1 where Winograd executes fewer), `cpu_baseline`
(the oracle on this box's host cores), `target` (north_star's 40 crops/s next to the exact-fp32 ceiling) and one-line
summaries `second_precision` / `c4_selftrain` / `c5_segformer`.  Every table -- kernel families of the timed and the
serialised steps, `roofline_other_kernels`, `bandwidth_kernels`, the full c4 / c5 / c1 / translator legs, `miou_parity` --
goes to `bench_detail.json` next to this script (and to gpurun_out/bench_detail.json).

Arithmetic (`--precision`): the convolutions run either in exact fp32 on the fp32 matrix cores ("f32", the default and
the HEADLINE: the reference's own precision; stride-1 3x3 layers through Winograd F(6x6) / F(4x4) / F(2x2)) or
with fp32 operands split into bf16 hi+lo and three bf16 MFMAs per product, fp32 accumulate ("bf16x3").  Everything else is
fp32.  BOTH arithmetics are timed with the same --steps / --warmup, each with its own roofline; the self-training (c4) and
SegFormer (c5) legs run with the same --steps / --warmup as well.
"""
import argparse
import json
import os
import random
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")   # (no op of the step reaches a vendor library; harmless for torch ops of a caller)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
F32_MFMA_PEAK_TFLOPS = 157.3     # v_mfma_f32_32x32x2_f32 dense peak
BF16_MFMA_PEAK_TFLOPS = 2500.0   # v_mfma_f32_32x32x16_bf16 dense peak
SERIAL_STEPS = 2          # steps of the serialised-stream pass that times kernels for the roofline
FWD_GFLOP_768 = 1232.9           # model forward, one 768x768 image
FWD_GFLOP_256 = 142.6
PROFILE_TAGS = ("r06", "r05", "r04", "r03", "r02")    # profiles/<tag>_<precision>_serial_pmc_summary.json feeds roofline.traffic (newest first)

CONFIGS = {
    # name: (arch, batch per GPU, H, W, label block, BASELINE.json entry, description)
    "c2": ("RESNET101", 8, 768, 768, 32, "configs[1]",
           "ResNet-101 DeepLabV2 DiGA warm-up (student + EMA teacher, KL distill), synthetic GTA5-shape"),
    "c1": ("TINY", 2, 256, 256, 16, "configs[0] stand-in",
           "small-backbone (Bottleneck 1-1-2-1, 16..128 planes) DeepLab warm-up step; the reference cannot build a "
           "ResNet-18 (seg_model_noaux.py:253 raises for BasicBlock)"),
    "c5": ("MIT_B5", 8, 768, 768, 32, "configs[4]",
           "SegFormer-B5 (MiT-B5 encoder, fp16 storage / fp32 accumulate; SegFormer all-MLP decode head on the four stages, logits "
           "at 1/4 scale) distillation student + EMA teacher, DiGA warm-up step (the build's own wiring: the reference ships "
           "encoder and head unwired)"),
    "c4": ("RESNET101", 8, 512, 1024, 32, "configs[3]",
           "self-training step (centroid pseudo-labeler + two ClassMix blocks + centroid EMA), synthetic "
           "Cityscapes-shape, B source + B target crops per GPU"),
}
DTYPE = {"f32": "f32",
         "bf16x3": "bf16x3 (conv operands = f32 split into bf16 hi+lo = 16 significand bits per operand, hi*hi + hi*lo + "
                   "lo*hi on the bf16 matrix cores, f32 accumulate; all other kernels f32)"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=None, help="override crops per GPU (debug only)")
    ap.add_argument("--size", type=int, nargs=2, default=None, help="override crop H W (debug only)")
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16x3"])
    ap.add_argument("--no-other-precision", action="store_true", help="skip the run of the other arithmetic")
    ap.add_argument("--other-steps", type=int, default=None, help="steps of the other arithmetic (default: --steps)")
    ap.add_argument("--other-warmup", type=int, default=None, help="warm-ups of the other arithmetic (default: --warmup)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the c4 / c5 / c1 legs")
    ap.add_argument("--multi-gpu-legs", action="store_true",
                    help="N > 1: also run the c4 / c5 legs on every rank (default: N > 1 measures the headline configuration and the "
                         "second arithmetic only -- the scaling curve is about `value`; the legs are 1-GPU results)")
    ap.add_argument("--c4-steps", type=int, default=None, help="steps of the self-training leg (default: --steps)")
    ap.add_argument("--c4-warmup", type=int, default=None, help="warm-ups of the self-training leg (default: --warmup)")
    ap.add_argument("--c5-steps", type=int, default=None, help="steps of the SegFormer leg (default: --steps)")
    ap.add_argument("--c5-head", choices=["segformer", "aspp"], default="segformer",
                    help="decode head of the SegFormer leg: the SegFormer all-MLP head (default) or the round-3 wiring (the DeepLab ASPP "
                         "classifier on the last stage only: a much lighter workload)")
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"),
                    help="file the full tables go to (kernel families, every roofline, bandwidth kernels, mIoU parity)")
    ap.add_argument("--no-graph", action="store_true", help="run the launch-bound legs (c1, c5) eagerly instead of from a HIP graph")
    ap.add_argument("--graph", action="store_true", help="run the MAIN configuration's step from a HIP graph (A/B runs of c5 / c1 as --config)")
    ap.add_argument("--no-bandwidth-kernels", action="store_true")
    ap.add_argument("--no-miou", action="store_true", help="skip the fixed-seed validation-mIoU parity leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="internal: run one CPU measurement and print its JSON")
    ap.add_argument("--cpu-threads", type=int, default=8)
    ap.add_argument("--cpu-part", default="768", choices=["768", "256"])
    ap.add_argument("--no-prof", action="store_true", help="do not bracket kernel families with HIP events")
    ap.add_argument("--lean", action="store_true",
                    help="only the timed steps (= --no-other-precision --no-other-configs --no-bandwidth-kernels "
                         "--no-cpu-baseline): the form rocprofv3 runs use")
    ap.add_argument("--serial-streams", action="store_true",
                    help="run the whole step on one stream (no teacher / weight-gradient side stream): the mode the "
                         "per-kernel roofline durations are measured in; profiles/*_serial_* are rocprofv3 runs of it")
    a = ap.parse_args()
    if a.other_steps is None:
        a.other_steps = a.steps
    if a.other_warmup is None:
        a.other_warmup = a.warmup
    if a.c4_steps is None:
        a.c4_steps = a.steps
    if a.c4_warmup is None:
        a.c4_warmup = a.warmup
    if a.c5_steps is None:
        a.c5_steps = a.steps
    if a.lean:
        a.no_other_precision = a.no_other_configs = a.no_bandwidth_kernels = a.no_cpu_baseline = a.no_miou = True
    return a


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(threads, part):
    """One measurement of the CPU leg in this process (child of cpu_baseline_subprocess): the oracle's warm-up step
    (PyTorch-CPU restatement of the reference, pinned by tests/golden) with `threads` torch threads.
      part "768": ResNet-101, B=1 source crop of 768x768 (= 2 student + 2 teacher images per step) -- the benchmark geometry,
                  one step after a small warm-up of the thread pool, measured, nothing scaled;
      part "256": the B=2 256x256 sample of earlier rounds (cross-check, FLOP-scaled) + BASELINE configs[0] as the build can
                  state it (small backbone, 2x256x256, exact size)."""
    import torch
    from oracle import deeplab as od
    from oracle import detweights, synth
    from oracle import step as ost
    torch.set_num_threads(threads)

    def timed(arch, seed, b, hw, nsteps):
        tr = ost.Trainer(detweights.state_dict(arch), detweights.state_dict(arch), arch=arch)
        rng = random.Random(5)
        dt = None
        for it in range(nsteps):
            batch = synth.warmup_batch(seed + it, b, hw, hw, block=16)
            t0 = time.perf_counter()
            tr.warmup_step(it, *batch, rng)
            dt = time.perf_counter() - t0
        return dt

    if part == "768":
        timed(od.TINY, 330, 1, 128, 1)                         # spin the thread pool / oneDNN primitives up
        t = [timed(od.RESNET101, 320 + i, 1, 768, 1) for i in range(2)]        # two independent steps: the better one counts
        return {"threads": threads, "s_per_step_768": min(t), "s_per_step_768_all": t}
    return {"threads": threads, "s_per_step_256": timed(od.RESNET101, 300, 2, 256, 2), "s_per_step_c1": timed(od.TINY, 310, 2, 256, 2)}


def cpu_baseline_subprocess():
    """The CPU leg: child processes (started before this process touches the GPU) so that a slow or memory-hungry host cannot
    take the GPU measurement down with it.  The 768x768 step is measured with 32 threads (beyond ~32 threads torch's CPU
    convolutions stop scaling on these hosts) and, time permitting, with every core the process may use; `value` is the
    better of the two."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1

    def child(threads, part, limit_s):
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--cpu-threads", str(threads),
                                "--cpu-part", part], capture_output=True, text=True, timeout=limit_s)
            for ln in reversed(r.stdout.splitlines()):
                if ln.startswith("{"):
                    return json.loads(ln)
            return {"error": f"rc={r.returncode}: {r.stderr[-300:]}"}
        except subprocess.TimeoutExpired:
            return {"error": f"not finished within {limit_s} s"}

    base = min(avail, 32)
    t0 = time.perf_counter()
    runs = {base: child(base, "768", 300)}
    if base < avail <= 64 and time.perf_counter() - t0 < 100:
        # (on the 256-core hosts of the pool the all-cores run does not finish in minutes: torch's CPU convolutions
        #  collapse beyond ~32 threads -- measured round 3 -- so it is only attempted on hosts with up to 64 cores)
        runs[avail] = child(avail, "768", 150)
    small = child(base, "256", 120)
    ok = {k: v["s_per_step_768"] for k, v in runs.items() if "s_per_step_768" in v}
    out = {"value": None, "unit": "crops/s", "cores": None, "kind": "port", "host_cores_total": os.cpu_count(),
           "host_cores_available": avail, "runs": {str(k): v for k, v in runs.items()}}
    if ok:
        cores = min(ok, key=ok.get)
        out.update(value=1.0 / ok[cores], cores=cores,
                   sample=f"oracle warm-up step, ResNet-101, B=1 source crop of 768x768 fp32 (2 student + 2 teacher images), best of "
                          f"2 steps, measured (not scaled): {ok[cores]:.2f} s with {cores} threads"
                          + "".join(f"; {v:.2f} s with {k} threads" for k, v in ok.items() if k != cores)
                          + f"; host has {os.cpu_count()} cores")
    else:
        out["sample"] = "768x768 CPU step failed: " + "; ".join(f"{k} threads: {v.get('error')}" for k, v in runs.items())
    if "s_per_step_256" in small:
        out["cross_check_256"] = {"value": (2.0 / small["s_per_step_256"]) * FWD_GFLOP_256 / FWD_GFLOP_768,
                                  "unit": "crops/s (FLOP-scaled)", "cores": base,
                                  "sample": f"B=2 crops of 256x256, 2nd of 2 steps: {small['s_per_step_256']:.2f} s/step, scaled to "
                                            f"768x768 crops by forward FLOPs ({FWD_GFLOP_256}/{FWD_GFLOP_768} GFLOP)"}
        out["c1"] = {"value": 2.0 / small["s_per_step_c1"], "unit": "256x256 crops/s", "cores": base,
                     "sample": f"oracle warm-up step, configs[0] stand-in (small backbone), B=2 crops of 256x256, 2nd of 2 steps: "
                               f"{small['s_per_step_c1']:.3f} s/step (exact size, no scaling)"}
    return out


def miou_parity_subprocess(limit_s=420):
    """Fixed-seed validation mIoU of the build (both conv arithmetics) against the capture of the reference
    (tests/golden/valmiou.npz), and -- north_star's training-level criterion -- ONE seed of the trained-model experiment
    (tests/golden/trainmiou.npz: 300 steps of the warm-up loop on a learnable task, then the two-scale validation) next to the
    reference's three training runs: the checker lives with the tests (it needs the oracle's deterministic weights), so it
    runs as a child process, after the timed regions."""
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "miou_parity.py"), "--trained"], capture_output=True, text=True,
                           timeout=limit_s)
        for ln in reversed(r.stdout.splitlines()):
            if ln.startswith("{"):
                return json.loads(ln)
        return {"error": f"rc={r.returncode}: {r.stderr[-300:]}"}
    except subprocess.TimeoutExpired:
        return {"error": f"no result within {limit_s} s"}


# ------------------------------------------------------------------------------------------------ N > 1 launcher
def rank_threads(local_world):
    """Host threads one rank may use: the step is ~2 600 asynchronous launches from ONE Python thread (+ autograd's backward
    thread); torch's intra-op pool is only touched by the few CPU-side tensor ops of ClassMix.  cores / ranks, capped at 8."""
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cores = os.cpu_count() or 1
    return max(1, min(8, cores // max(local_world, 1)))


def pin_rank_to_cores(local_rank, local_world):
    """Multi-rank host hygiene, called BEFORE the rank touches the GPU: the rank's threads are confined to its own contiguous
    slice of the cores this process may run on (8 ranks x 256 unpinned torch threads on a 2-socket host otherwise migrate across
    sockets while they drive thousands of launches per step), and OMP / torch intra-op threads are capped (`rank_threads`).
    Contiguous slices keep a rank on one socket / NUMA node for the usual core numbering.  DIGA_BENCH_PIN=0 switches it off.
    Returns a dict for the bench line (`host`)."""
    info = {"local_rank": local_rank, "local_world": local_world, "pinned": False}
    threads = rank_threads(local_world)
    os.environ.setdefault("OMP_NUM_THREADS", str(threads))
    info["omp_num_threads"] = int(os.environ["OMP_NUM_THREADS"])
    if local_world <= 1 or os.environ.get("DIGA_BENCH_PIN", "1") == "0":
        return info
    try:
        cores = sorted(os.sched_getaffinity(0))
        per = len(cores) // local_world
        if per >= 1:
            mine = cores[local_rank * per:(local_rank + 1) * per]
            os.sched_setaffinity(0, mine)
            info.update(pinned=True, cores=[mine[0], mine[-1]], n_cores=len(mine))
    except (AttributeError, OSError) as e:          # (not fatal: a container may forbid it)
        info["pin_error"] = str(e)
    return info


def spawn_workers(a):
    """--gpus N > 1 without a torchrun environment: start N fresh worker processes (this process has not touched the
    GPU and never will), one rank per GPU; rank 0 inherits stdout and prints the JSON line."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    threads = rank_threads(a.gpus)
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", str(threads)),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        out = None if r == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in pending:          # a dead rank leaves the others hanging in a collective
                        q.terminate()
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


# ------------------------------------------------------------------------------------------------ timed steps
def geometry(a, config):
    arch_name, B, H, W, block, _, _ = CONFIGS[config]
    if config == a.config:
        B = a.batch or B
        if a.size:
            H, W = a.size
    return arch_name, B, H, W, block


def run_steps(a, config, precision, steps, warmup, rank, world, dev, prof, graph=False):
    """Build student/teacher, run `warmup` + `steps` iterations of `config`; returns (seconds for `steps` = max over
    ranks, kernel families, last losses, parameter counts, geometry)."""
    import torch
    from diga_amd import _lib, ddp, synthetic
    from diga_amd import config as dcfg
    from diga_amd.model import seg_model_noaux as sm
    from diga_amd.model.model_noaux import SegModel
    from diga_amd.train_step import DigaTrainer

    # the step's configuration is explicit (diga_amd/config.py): the process defaults (environment read once at import; gloo runs
    # have had their side streams switched off by ddp.init_from_env) with this leg's arithmetic and stream policy
    step_cfg = dcfg.DEFAULTS.replace(conv_math=1 if precision == "bf16x3" else 0)
    if a.serial_streams:
        step_cfg = step_cfg.serial_streams()
    _lib.set_conv_math(step_cfg.conv_math)       # (model construction and anything outside a step follow the same arithmetic)
    arch_name, B, H, W, block = geometry(a, config)
    torch.manual_seed(0)                       # identical random-init weights on every rank
    if arch_name.startswith("MIT_"):
        from diga_amd.model.segformer import SegFormerStudent
        student = SegFormerStudent(arch_name.lower(), head=a.c5_head).to(dev)
        teacher = SegFormerStudent(arch_name.lower(), head=a.c5_head).to(dev)
    else:
        arch = getattr(sm, arch_name)
        student, teacher = SegModel(arch=arch).to(dev), SegModel(arch=arch).to(dev)
    ddp.broadcast_module(student)
    teacher.train()
    rng = random.Random(1234 + rank)           # ClassMix class choice differs per rank, reproducibly
    # graph=True (the launch-bound legs c1 / c5): the static part of the step is captured into a HIP graph and replayed
    # (diga_amd/train_step.py); kernel-family timings then come from extra eager steps after the timed region
    tr = DigaTrainer(student, teacher, rng=rng, graph=graph, config=step_cfg)
    if config == "c4":
        from diga_amd.calc_centroids import Class_Features
        batch = synthetic.selftrain_batch(1234 + rank, B, H, W, block=block, device=dev)
        cf = Class_Features(numbers=19)
        g7 = torch.Generator(device="cpu")
        g7.manual_seed(7)
        cf.objective_vectors = torch.randn((19, 256), generator=g7).to(dev)

        def one_step(i):
            return tr.selftrain_step(i, *batch, cf)
    else:
        batch = synthetic.warmup_batch(1234 + rank, B, H, W, block=block, device=dev)

        def one_step(i):
            return tr.warmup_step(i, *batch)
    counts = (sum(p.numel() for p in student.parameters() if p.requires_grad),
              sum(p.numel() for p in student.parameters()))

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    labels = batch[3]
    inner_step = one_step

    def one_step(i):                           # noqa: F811
        # the data-loader side of the step: as soon as a step is enqueued, the class lists ClassMix needs for the NEXT batch
        # (here: the same synthetic labels) are fetched on a side stream (DigaTrainer.prefetch_classmix) -- the histogram launch
        # and its D->H copy still happen once per step, they just no longer drain the GPU at the top of the next step
        out_ = inner_step(i)
        if os.environ.get('DIGA_BENCH_NO_PREFETCH') != '1':
            tr.prefetch_classmix(labels)
        return out_

    it = 0
    if graph:
        warmup = max(warmup, 2)                # step 0 eager, step 1 captures: both outside the timed region
    for _ in range(warmup):
        one_step(it)
        it += 1
    torch.cuda.synchronize()
    barrier()
    prof_timed = prof and not graph            # HIP events cannot be recorded into a replayed graph
    if prof_timed:
        _lib.call("diga_prof_reset")
        _lib.call("diga_prof_enable", 1)
    torch.cuda.reset_peak_memory_stats(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = one_step(it)
        it += 1
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    # peak device memory of the timed region: what the step's tensors occupy at their high-water mark (allocated) and what the
    # caching allocator holds from the driver for them (reserved); workspaces and kept Winograd transforms included
    peak_mem = {"allocated": torch.cuda.max_memory_allocated(dev) / 1e9, "reserved": torch.cuda.max_memory_reserved(dev) / 1e9}
    if prof_timed:
        _lib.call("diga_prof_enable", 0)
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    losses = {k: float(v) for k, v in out.items()}
    if not all(v == v and abs(v) < 1e6 for v in losses.values()):
        raise SystemExit(f"non-finite loss in the timed region: {losses}")
    agree = None
    if world > 1:
        # data-parallel invariant: after the all-reduced steps every rank holds the same student, bit for bit
        chk = torch.stack([p.detach().double().sum() for p in student.parameters()]).sum().reshape(1)
        allc = [torch.empty_like(chk) for _ in range(world)]
        torch.distributed.all_gather(allc, chk)
        agree = all(bool(torch.equal(c, allc[0])) for c in allc)
        if not agree:
            raise SystemExit(f"rank {rank}: student parameters differ across ranks after the timed region: {[float(c) for c in allc]}")
    losses["_ranks_agree"] = agree
    losses["_peak_mem_gb"] = peak_mem

    def query(nsteps):
        fam = {}
        for tag in _lib.PROF_TAGS:
            n, ms = _lib.prof_query(tag)
            if n:
                fam[tag] = {"launches": n, "avg_ms": ms / n, "ms_per_step": ms / nsteps,
                            "work_per_step": _lib.prof_work(tag) / nsteps}
        return fam

    families_overlapped = query(steps) if prof_timed else {}
    families = families_overlapped
    if graph:
        tr.use_graph = False                   # the eager steps below time the kernels
    if prof and (graph or not a.serial_streams):
        # Kernel durations for the roofline: the timed region runs the teacher forward and the weight gradients on a
        # second stream, so the HIP events of a launch there span other kernels sharing the GPU.  Two more steps
        # with the streams serialised time every kernel on its own (all ranks, the step has a collective).
        tr.cfg = step_cfg.serial_streams()       # (c4: the one-backward form, no third stream, while kernels are timed)
        try:
            one_step(it)
            it += 1
            torch.cuda.synchronize()
            _lib.call("diga_prof_reset")
            _lib.call("diga_prof_enable", 1)
            from diga_amd.model import conv as _dconv
            _dconv.flop_log = {}
            for _ in range(SERIAL_STEPS):
                one_step(it)
                it += 1
            torch.cuda.synchronize()
            _lib.call("diga_prof_enable", 0)
            families = query(SERIAL_STEPS)
            # what the matrix cores execute next to the FLOPs of the direct convolutions (Winograd layers: 16/36 per 2x2 tile)
            for tag, (direct, executed) in _dconv.flop_log.items():
                if tag in families:
                    families[tag]["direct_flops_per_step"] = direct / SERIAL_STEPS
                    families[tag]["executed_flops_per_step"] = executed / SERIAL_STEPS
            _dconv.flop_log = None
        finally:
            tr.cfg = step_cfg
        barrier()
    if prof:
        _lib.call("diga_prof_reset")
    del tr, student, teacher, batch, one_step, inner_step, labels
    if os.environ.get('DIGA_BENCH_NO_EMPTY_CACHE') != '1':
        torch.cuda.empty_cache()
    return float(t), (families, families_overlapped), losses, counts, (B, H, W, arch_name)


def target_statement(config, precision, families, B, H, W, n_stu, n_tea):
    """north_star asks for >= 40 crops/s/GPU; in exact fp32 the step is capped by the fp32 matrix pipe: the FLOPs the step's
    convolutions execute (after Winograd) / 157.3 TFLOP/s."""
    if config != "c2" or precision != "f32" or (B, H, W) != (8, 768, 768):
        return None
    ex = sum(f.get("executed_flops_per_step", 0.0) for t, f in families.items() if t.startswith("conv_"))
    if not ex:
        return None
    floor_ms = ex / (F32_MFMA_PEAK_TFLOPS * 1e12) * 1e3
    return {"north_star_crops_per_s_per_gpu": 40.0, "fp32_ceiling_crops_per_s": float(f"{B / (floor_ms * 1e-3):.4g}"),
            "executed_conv_tflop_per_step": float(f"{ex / 1e12:.4g}"),
            "note": "40 crops/s needs ~400 TFLOP/s: beyond the fp32 matrix peak (157.3); ceiling = executed conv FLOPs / peak"}


def images_per_step(config, B):
    """Images through the student (forward + backward) and through the teacher (forward) per step and GPU."""
    return (3 * B, 3 * B) if config == "c4" else (2 * B, 2 * B)


def executed_fraction(fam, out, peak):
    """`achieved` / `frac` of an MFMA-bound family = the FLOPs its kernels EXECUTE on the matrix cores / the summed launch
    durations (/ peak): a utilisation, always <= 1.  The stride-1 3x3 layers run as Winograd (csrc/winograd.hip) and execute
    fewer multiplications than the direct convolution SURVEY section 8d prices (16/36 per 2x2 tile, 36/144 per 4x4 tile), so
    the algorithmic figure (direct-convolution FLOPs / the same time, which can exceed the peak) is kept next to it as
    `achieved_algorithmic` / `frac_algorithmic`.  The family's time includes its transform passes."""
    if "executed_flops_per_step" in fam and fam.get("direct_flops_per_step"):
        ex = fam["executed_flops_per_step"] / (fam["ms_per_step"] * 1e-3) / 1e12
        out.update(achieved_algorithmic=out["achieved"], frac_algorithmic=out["frac"],
                   executed_flops_per_step=fam["executed_flops_per_step"], achieved=ex, frac=ex / peak)


def rooflines(config, precision, families, counts, geom):
    """Dominant kernel: the forward implicit-GEMM convolution (MFMA-bound).  Algorithmic work of its launches in
    one step = model forward FLOPs (1232.9 GFLOP per 768x768 image, scaled by area) x the images
    of the student's and the teacher's forward passes; achieved = that / the summed HIP-event durations of those
    launches.  The other families: the algorithmic bytes (or FLOPs) each library call declared / its duration."""
    B, H, W, arch_name = geom
    n_trainable, n_params = counts
    roof, other = None, {}
    fwd_gflop = FWD_GFLOP_768 * (H * W) / (768.0 * 768.0)
    # bf16x3: three bf16 MFMAs (2.5 PFLOP/s dense) per algorithmic multiply-add
    peak = F32_MFMA_PEAK_TFLOPS if precision == "f32" else BF16_MFMA_PEAK_TFLOPS / 3.0
    # HBM traffic per launch comes from separate rocprofv3 --pmc passes of this same command (FETCH_SIZE doubled
    # as MI355X_MICROARCH.md prescribes), condensed by tools/summarize_prof.py into profiles/
    pmc = None
    for tag in PROFILE_TAGS:
        try:
            # (the self-training leg has its own passes since round 5: profiles/r05_c4_f32_serial_pmc_summary.json)
            with open(os.path.join(ROOT, "profiles", f"{tag}_{'c4_' if config == 'c4' else ''}{precision}_serial_pmc_summary.json")) as fh:
                pmc = json.load(fh)["kernels"]
            break
        except (OSError, ValueError, KeyError):
            continue
    n_stu, n_tea = images_per_step(config, B)
    if arch_name.startswith("MIT_") and "mit_gemm" in families:
        # --config c5 as the main configuration: the dominant family is the encoder's fp16 GEMMs (declared 2*M*N*K per call)
        fam = families["mit_gemm"]
        ach = fam["work_per_step"] / (fam["ms_per_step"] * 1e-3) / 1e12
        n_launch = fam["ms_per_step"] / fam["avg_ms"]
        roof = {"kernel": "diga::mit::gemm_nt_kernel (fp16 MFMA, fp32 accumulate: Linear forward / backward-data and the patch-embedding / "
                          "spatial-reduction convolutions of the MiT encoder; all launches of a step)",
                "bound": "mfma", "achieved": ach, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / BF16_MFMA_PEAK_TFLOPS,
                "traffic": None, "algorithmic_flops_per_launch": fam["work_per_step"] / n_launch, "avg_launch_ms": fam["avg_ms"],
                "launches_per_step": n_launch}
    if "conv_fwd" in families and arch_name == "RESNET101":
        fam = families["conv_fwd"]
        flops_step = (n_stu + n_tea) * fwd_gflop * 1e9
        n_launch = fam["ms_per_step"] / fam["avg_ms"]                     # launches per step
        ach = flops_step / (fam["ms_per_step"] * 1e-3) / 1e12
        kname = ("gemm_f32_persistent_kernel + conv_fwd_dma_kernel + Winograd transforms" if precision == "f32"
                 else "conv_fwd_x3w_kernel + conv_fwd_x3t8_kernel")
        traffic = None
        if pmc and ((B, H, W) == (8, 768, 768) and config == "c2" or (B, H, W) == (8, 512, 1024) and config == "c4"):
            # launch-weighted mean over the family's kernels (the 128-column instantiations carry > 95 % of its time;
            # the same kernels also serve backward-data, whose launches are in the PMC averages)
            names = ("diga::gemm_f32_persistent_kernel", "diga::conv_fwd_dma_kernel", "diga::conv_fwd_kernel<2", "diga::wino::winoM_input_kernel",
                     "diga::wino::winoM_output_kernel", "diga::wino::winoM_output_stats_kernel") if precision == "f32" else ("diga::conv_fwd_x3w_kernel<2", "diga::conv_fwd_x3t8_kernel<2", "diga::conv_fwd_x3t_kernel<2")
            ent = [v for k, v in pmc.items() if k.startswith(names)]
            nl = sum(v["launches"] for v in ent)
            traffic = sum(v["hbm_bytes_per_launch_corrected"] * v["launches"] for v in ent) / nl if nl else None
        roof = {"kernel": f"{kname} (all forward-convolution launches of a step, {'fp32' if precision == 'f32' else 'bf16'} MFMA)",
                "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic,
                "algorithmic_flops_per_launch": flops_step / n_launch, "avg_launch_ms": fam["avg_ms"],
                "launches_per_step": n_launch, "declared_flops_per_step": fam["work_per_step"]}
        executed_fraction(fam, roof, peak)
        for tag in ("conv_bwd_data", "conv_bwd_weight"):
            if tag in families:        # backward: one pass each over the student's images
                f = n_stu * fwd_gflop * 1e9
                v = f / (families[tag]["ms_per_step"] * 1e-3) / 1e12
                other[tag] = {"bound": "mfma", "unit": "TFLOP/s", "peak": peak, "achieved": v, "frac": v / peak,
                              "ms_per_step": families[tag]["ms_per_step"]}
                executed_fraction(families[tag], other[tag], peak)
    # per-launch byte counts the host knows (the multi-tensor kernels get device-side size tables)
    known = {"sgd": 20.0 * n_trainable, "ema": 12.0 * n_params}
    for tag, fam in families.items():
        if tag.startswith("conv_"):
            continue
        nbytes = known.get(tag, fam["work_per_step"] / max(fam["ms_per_step"] / fam["avg_ms"], 1e-9))
        if not nbytes:
            continue
        v = nbytes / (fam["avg_ms"] * 1e-3) / 1e9
        other[tag] = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "achieved": v, "frac": v / HBM_PEAK_GBS,
                      "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": fam["avg_ms"], "ms_per_step": fam["ms_per_step"]}
    return roof, other


def translator_leg(dev, precision):
    """the frozen source->target translator dec_s2t(enc_s(x)) that runs every step right before the
    scoped path (warm_up.py:235-237), B = 8 crops of 768x768, no_grad: reflection pads, nearest upsampling and tanh inside
    the conv kernels, InstanceNorm on the HIP kernels -- next to the same modules with explicit pad / upsample / tanh ops."""
    import torch
    from diga_amd import _lib
    from diga_amd.model.model_noaux import ImgDecoder, ImgEncoder
    _lib.set_conv_math(1 if precision == "bf16x3" else 0)
    torch.manual_seed(3)
    enc, dec = ImgEncoder().to(dev).eval(), ImgDecoder().to(dev).eval()
    for m in (enc, dec):
        for p in m.parameters():
            p.requires_grad_(False)
    x = torch.rand((8, 3, 768, 768), device=dev) * 2 - 1
    res = {}
    for name, ctx in (("folded", torch.no_grad), ("explicit_ops", torch.enable_grad)):
        with ctx():
            y = dec(enc(x))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                y = dec(enc(x))
            torch.cuda.synchronize()
            res[name + "_ms"] = 1e3 * (time.perf_counter() - t0) / 3
        del y
    # roofline of the folded pass: FLOPs of its convolutions as direct convolutions (algorithmic) and as executed (the sixteen 3x3
    # 256 -> 256 ResBlock convs run on F(6x6,3x3) Winograd with the reflection padding folded into the input transform since round 5)
    from diga_amd.model import conv as _dconv
    _dconv.flop_log = {}
    with torch.no_grad():
        dec(enc(x))
    torch.cuda.synchronize()
    direct = sum(v[0] for v in _dconv.flop_log.values())
    executed = sum(v[1] for v in _dconv.flop_log.values())
    _dconv.flop_log = None
    peak = F32_MFMA_PEAK_TFLOPS if precision == "f32" else BF16_MFMA_PEAK_TFLOPS / 3.0
    sec = res["folded_ms"] * 1e-3
    res.update(images_per_s=8.0 / sec, dtype=precision,
               workload="dec_s2t(enc_s(x)), x [8,3,768,768], random-init weights, inference",
               roofline={"bound": "mfma", "unit": "TFLOP/s", "peak": peak, "achieved": executed / sec / 1e12, "frac": executed / sec / 1e12 / peak,
                         "achieved_algorithmic": direct / sec / 1e12, "frac_algorithmic": direct / sec / 1e12 / peak,
                         "direct_tflop": direct / 1e12, "executed_tflop": executed / 1e12,
                         "note": "whole pass (convolutions + InstanceNorm passes) timed; FLOPs = its convolutions only"})
    del enc, dec, x
    torch.cuda.empty_cache()
    return res


def bandwidth_kernels(dev):
    """The HBM-bound kernels north_star names, timed on their own at full size (HIP events on the launch stream, 10
    launches after 2 warm-ups each): the loss kernels at the reference's API boundary (full-resolution logits, the path
    an unmodified train_DiGA_*.py takes) at C2 size, the centroid pseudo-labeler and the class-mean pass at C4 size."""
    import torch
    from diga_amd import _lib
    from diga_amd.calc_centroids import Class_Features
    from diga_amd.util import loss as L
    out = {}
    g = torch.Generator(device="cpu")
    g.manual_seed(99)

    def timed(tags, fn, note, rename=None):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        _lib.call("diga_prof_reset")
        _lib.call("diga_prof_enable", 1)
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        _lib.call("diga_prof_enable", 0)
        for tag in tags:
            n, ms = _lib.prof_query(tag)
            if not n:
                continue
            nbytes = _lib.prof_work(tag) / n
            gbs = nbytes / (ms / n * 1e-3) / 1e9
            out[(rename or {}).get(tag, tag)] = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "achieved": gbs, "frac": gbs / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": ms / n, "launches": n, "shape": note}
        _lib.call("diga_prof_reset")

    # C2: B = 8 crops of 768x768, 19 classes; CE on the first B images, distillation on 2B
    B, C, H, W = 8, 19, 768, 768
    stu = torch.randn((2 * B, C, H, W), device=dev).requires_grad_()
    tea = torch.randn((2 * B, C, H, W), device=dev)
    lab = torch.randint(0, 19, (B, H, W), device=dev, dtype=torch.int64)
    lab[:, ::17, ::13] = 255
    timed(["ce2d"], lambda: L.cross_entropy2d(stu[:B], lab), f"cross_entropy2d fwd+bwd, logits [{B},{C},{H},{W}] fp32 + int64 labels")
    timed(["distill"], lambda: L.distillation_loss(tea, stu, 0.5), f"distillation_loss fwd+bwd, teacher/student [{2 * B},{C},{H},{W}] fp32")
    del stu, tea, lab
    from diga_amd.util import augment as A
    aug = A.ExtraAug(seed=1)
    xa = torch.randn((8, 3, 768, 768), device=dev)
    timed(["elementwise"], lambda: aug.view(xa, 0.4, (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)),
          "colour-augmentation view beta*norm(extra_aug(x)) + (1-beta)*x, x [8,3,768,768] fp32, one pass", {"elementwise": "color_aug_view"})
    del xa
    # C4: B = 8 target crops of 512x1024 -> low-res 65x129, 256-channel features
    B, D, h, w, H, W = 8, 256, 65, 129, 512, 1024
    cf = Class_Features(numbers=19)
    cf.objective_vectors = torch.randn((19, D), generator=g).to(dev)
    feat = torch.randn((B, D, h, w), device=dev)
    logit = torch.randn((B, 19, h, w), device=dev)
    pseudo = torch.randint(0, 19, (B, H, W), device=dev, dtype=torch.int64)
    timed(["centroid_weights", "consensus"], lambda: cf.consensus_pseudo_labels(feat, pseudo),
          f"feat [{B},{D},{h},{w}] -> softmax(-dist) weights -> upsample + argmax + consensus on [{B},{H},{W}] int64")
    timed(["class_means", "centroid_apply"], lambda: cf.update_from_batch(feat, logit, labels_full=pseudo),
          f"class sums of feat [{B},{D},{h},{w}] under argmax(out) & nearest-downsampled labels + sequential centroid EMA")
    torch.cuda.empty_cache()
    return out


def main():
    a = parse()
    if a.cpu_baseline_only:
        print(json.dumps(cpu_baseline(a.cpu_threads, a.cpu_part)), flush=True)
        return 0
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        return spawn_workers(a)
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("DIGA_BENCH_FAULT_AFTER"):
        # debugging aid for a stuck multi-rank run: every rank dumps its Python stacks to stderr after N seconds and exits
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["DIGA_BENCH_FAULT_AFTER"]), exit=True)
    cpu_line = None
    if world_env == 1 and not a.no_cpu_baseline:
        cpu_line = cpu_baseline_subprocess()
    host = None
    if world_env > 1:
        # (before `import torch` reads OMP_NUM_THREADS and before anything touches the GPU)
        host = pin_rank_to_cores(int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("LOCAL_WORLD_SIZE", str(world_env))))
    import torch
    from diga_amd import ddp
    if host is not None:
        torch.set_num_threads(host["omp_num_threads"])
    rank, world, local = ddp.init_from_env()
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the hot path has no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    backend = torch.distributed.get_backend() if world > 1 else None
    prof = not a.no_prof

    dt, (families, families_ov), losses, counts, geom = run_steps(a, a.config, a.precision, a.steps, a.warmup, rank, world,
                                                                  dev, prof, graph=bool(a.graph and world == 1))
    B, H, W, _ = geom
    other_line = None
    if not a.no_other_precision:
        # the other arithmetic, same --steps / --warmup, same barrier + synchronize bracket, its own roofline
        oprec = "f32" if a.precision == "bf16x3" else "bf16x3"
        odt, (ofam, _), olosses, _, _ = run_steps(a, a.config, oprec, a.other_steps, a.other_warmup, rank, world, dev, prof)
        oroof, oother = rooflines(a.config, oprec, ofam, counts, geom)
        if oroof is not None:
            oroof["measured"] = f"HIP events around every launch; {SERIAL_STEPS} extra steps with the side streams serialised"
        other_line = {"metric": f"{H}x{W} 19-class crops/sec (DiGA warm-up step)" if a.config != "c4" else
                                f"{H}x{W} 19-class (source,target) crop pairs/sec (DiGA self-training step)",
                      "dtype": DTYPE[oprec], "value": world * B * a.other_steps / odt,
                      "unit": "crops/s" if a.config != "c4" else "pairs/s", "n_gpus": world,
                      "steps": a.other_steps, "warmup": a.other_warmup, "ms_per_step": 1e3 * odt / a.other_steps,
                      "timed_region_s": odt, "peak_mem_gb": olosses.get("_peak_mem_gb"), "roofline": oroof, "roofline_other_kernels": oother,
                      "kernel_families": ofam, "losses_last_step": {k: v for k, v in olosses.items() if not k.startswith("_")}}
    other_cfg = {}
    if not a.no_other_configs and (world == 1 or a.multi_gpu_legs):
        for cfg, (st, wu) in (("c5", (a.c5_steps, max(a.warmup, 2))), ("c4", (a.c4_steps, a.c4_warmup)), ("c1", (5, 2))):
            if cfg == a.config or (cfg == "c1" and world > 1):
                continue
            use_graph = cfg in ("c1", "c5") and not a.no_graph and world == 1
            cdt, (cfam, _), closs, ccounts, cgeom = run_steps(a, cfg, a.precision, st, wu, rank, world, dev, prof, graph=use_graph)
            cB, cH, cW, carch = cgeom
            croof, cother = rooflines(cfg, a.precision, cfam, ccounts, cgeom)
            n_stu, n_tea = images_per_step(cfg, cB)
            other_cfg[cfg] = {
                "workload": f"{CONFIGS[cfg][5]}: {CONFIGS[cfg][6]}, {cH}x{cW}, batch {cB} per GPU",
                "metric": (f"{cH}x{cW} 19-class (source,target) crop pairs/sec (DiGA self-training step)" if cfg == "c4"
                           else f"{cH}x{cW} 19-class crops/sec (DiGA warm-up step)"),
                "value": world * cB * st / cdt, "unit": "pairs/s" if cfg == "c4" else "crops/s", "steps": st, "warmup": wu,
                "ms_per_step": 1e3 * cdt / st, "n_gpus": world, "peak_mem_gb": closs.get("_peak_mem_gb"),
                "dtype": ("fp16 storage / fp32 accumulate (MiT encoder); head convs " + a.precision) if cfg == "c5" else a.precision,
                "kernel_families": cfam if cfg in ("c5", "c4") else None,
                "images_per_step_per_gpu": {"student_fwd_bwd": n_stu, "teacher_fwd": n_tea},
                "hip_graph": bool(use_graph), **({"head": a.c5_head} if cfg == "c5" else {}),
                **({"note": "SegFormer head: linear_fuse (mmcv ConvModule) is pinned to a stand-in conv(bias=False) -> BN -> ReLU = mmcv 1.x's documented "
                            "behaviour, not its code (mmcv absent); MiT encoder pinned to the reference module"} if cfg == "c5" and a.c5_head == "segformer" else {}),
                "roofline_conv_fwd": None if croof is None else {k: croof.get(k) for k in ("achieved", "peak", "unit", "frac", "frac_algorithmic", "traffic")},
                "roofline_other_kernels": cother if cfg == "c4" else None,
                "losses_last_step": {k: v for k, v in closs.items() if not k.startswith("_")}}
            if cfg == "c5":
                # MFMA-bound families of the encoder: declared algorithmic FLOPs / summed HIP-event durations, against the
                # dense fp16 MFMA peak (= bf16's); the HBM-bound ones against 8 TB/s
                rl = {}
                for tag, bound in (("mit_attn_fwd", "mfma"), ("mit_attn_bwd", "mfma"), ("mit_gemm", "mfma"), ("mit_wgrad", "mfma"),
                                   ("mit_norm", "hbm"), ("mit_dwconv", "hbm"), ("mit_misc", "hbm")):
                    if tag in cfam and cfam[tag]["ms_per_step"] > 0:
                        rate = cfam[tag]["work_per_step"] / (cfam[tag]["ms_per_step"] * 1e-3)
                        peak = BF16_MFMA_PEAK_TFLOPS if bound == "mfma" else HBM_PEAK_GBS
                        ach = rate / (1e12 if bound == "mfma" else 1e9)
                        rl[tag] = {"bound": bound, "achieved": ach, "peak": peak, "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
                                   "frac": ach / peak, "ms_per_step": cfam[tag]["ms_per_step"]}
                other_cfg[cfg]["roofline_families"] = rl
    bw = None
    if rank == 0 and not a.no_bandwidth_kernels:
        bw = bandwidth_kernels(dev)
    if rank == 0 and world == 1 and not a.no_other_configs:
        other_cfg["translator"] = translator_leg(dev, a.precision)

    miou = None
    if rank == 0 and world == 1 and not a.no_miou:
        torch.cuda.synchronize()
        miou = miou_parity_subprocess()

    if rank == 0:
        roof, other = rooflines(a.config, a.precision, families, counts, geom)
        if roof is not None:
            roof["measured"] = ("HIP events around every launch, on the launch stream; "
                                + ("timed region (single stream)" if a.serial_streams else
                                   f"{SERIAL_STEPS} extra steps with the teacher / weight-gradient side streams serialised "
                                   "(in the timed region they overlap other kernels; see kernel_families_overlapped)"))
        n_stu, n_tea = images_per_step(a.config, B)
        fwd_tflop = FWD_GFLOP_768 * (H * W) / (768.0 * 768.0) / 1e3
        line = {
            "metric": (f"{H}x{W} 19-class crops/sec (DiGA warm-up step)" if a.config != "c4" else
                       f"{H}x{W} 19-class (source,target) crop pairs/sec (DiGA self-training step)"),
            "value": world * B * a.steps / dt, "unit": "crops/s" if a.config != "c4" else "pairs/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": DTYPE[a.precision], "data": "synthetic",
            # BASELINE.md publishes no number for this metric (vs_baseline stays null); north_star's target next to what exact fp32 allows
            "target": target_statement(a.config, a.precision, families, B, H, W, n_stu, n_tea),
            "config": {"workload": f"{CONFIGS[a.config][5]}: {CONFIGS[a.config][6]} {H}x{W}, batch {B} per GPU",
                       "global_batch": world * B, "crop": [H, W], "parallelism": f"dp{world}",
                       "images_per_step_per_gpu": {"student_fwd_bwd": n_stu, "teacher_fwd": n_tea},
                       # what pins this geometry to the reference: the reference's own forward / backward at 768 x 768 on
                       # 2 and on 8 images (75 272 rows per GEMM; the step's 16-image pass, 150 544 rows, does not fit a CPU capture in this
                       # container's 62 GB) and a 3-step training trajectory at 768 x 768
                       "reference_pins": "full768 (2 img), full768b8 (8 img: half the step's 16-image pass), traj768 (3 steps), trainmiou (300 steps @128)"},
            "rccl_ranks": world, "backend": backend if world > 1 else "none (single process)",
            "ranks_agree": losses.pop("_ranks_agree", None), "peak_mem_gb": losses.pop("_peak_mem_gb", None), "host": host,
            "roofline": roof, "roofline_other_kernels": other, "bandwidth_kernels": bw, "cpu_baseline": cpu_line,
            "second_precision": other_line, "timed_region_s": dt, "other_configs": other_cfg or None, "miou_parity": miou, "kernel_families": families,
            "kernel_families_overlapped": None if a.serial_streams else families_ov, "losses_last_step": losses,
            # student: forward + backward-data + backward-weight (3 passes) ; teacher: forward
            "model_tflop_per_step_per_gpu": (3 * n_stu + n_tea) * fwd_tflop,
        }
        emit(line, a.detail)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    return 0


COMPACT_LIMIT = 4000    # bytes: the driver reads the tail of stdout; a line it cannot hold whole is a line it cannot parse


def compact(line):
    """The ONE stdout line: the contract's fields, the dominant kernel's roofline, the CPU baseline, and one-line summaries of the
    other arithmetic / the self-training (c4) and SegFormer (c5) legs.  Every table stays in the detail file."""
    def num(v, nd=4):
        return None if v is None else float(f"{v:.{nd}g}") if isinstance(v, float) else v

    def pick(d, keys, nd=4):
        return None if not d else {k: (num(d[k], nd) if not isinstance(d[k], str) else d[k][:200]) for k in keys if k in d}

    roof = pick(line.get("roofline"), ("kernel", "bound", "achieved", "peak", "unit", "frac", "frac_algorithmic", "traffic",
                                       "algorithmic_flops_per_launch", "avg_launch_ms", "launches_per_step"), 5)
    cpu = pick(line.get("cpu_baseline"), ("value", "unit", "cores", "kind", "sample"), 4)
    if cpu and "sample" in cpu:
        cpu["sample"] = cpu["sample"][:200]

    def leg(d):
        if not d:
            return None
        r = d.get("roofline") or d.get("roofline_conv_fwd") or {}
        return {"value": num(d["value"]), "unit": d["unit"], "ms_per_step": num(d["ms_per_step"]), "steps": d["steps"],
                "warmup": d["warmup"], "dtype": d["dtype"].split(" (")[0], "frac": num(r.get("frac")),
                "peak_mem_gb": num((d.get("peak_mem_gb") or {}).get("allocated"), 4),
                **({"traffic": num(r.get("traffic"), 4)} if r.get("traffic") else {}),
                **({"head": d["head"]} if "head" in d else {})}

    oc = line.get("other_configs") or {}
    out = {k: (num(line[k], 6) if isinstance(line[k], float) else line[k]) for k in
           ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "timed_region_s", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data")}
    out["dtype"] = out["dtype"].split(" (")[0]
    cfg = line["config"]
    out["config"] = {"workload": cfg["workload"][:220], "global_batch": cfg["global_batch"], "crop": cfg["crop"],
                     "parallelism": cfg["parallelism"]}
    pm = line.get("peak_mem_gb") or {}
    out["peak_mem_gb"] = {k: num(v, 4) for k, v in pm.items()} or None
    mp = line.get("miou_parity") or {}
    tm = mp.get("trained_model") or {}
    if mp and "error" not in mp:
        # val mIoU half of the metric: fixed weights (argmax parity of the validation pass) and a TRAINED model (one seed run live;
        # reference = three training runs of the reference's own loop, tests/golden/trainmiou.npz)
        out["miou_parity"] = {"fixed_weights_delta_points": num((mp.get("f32") or {}).get("miou_delta_points"), 3),
                              "trained_hip_miou": num(tm.get("hip_final_miou"), 4), "trained_reference_same_seed": num(tm.get("reference_same_seed"), 4),
                              "trained_reference_mean": num(tm.get("reference_mean"), 4), "trained_reference_seed_spread": num(tm.get("reference_seed_spread"), 3)}
    out.update(rccl_ranks=line["rccl_ranks"], backend=line.get("backend"), ranks_agree=line.get("ranks_agree"), roofline=roof, cpu_baseline=cpu, target=line.get("target"),
               second_precision=leg(line.get("second_precision")), c4_selftrain=leg(oc.get("c4")), c5_segformer=leg(oc.get("c5")),
               detail=line.get("detail_file"))
    return out


def emit(line, detail_path):
    """Write the full record to `detail_path` (and a copy under gpurun_out/ so that it comes home from a GPU box), then print the
    compact line -- strict JSON, re-parsed and length-checked before it goes out -- as the last thing on stdout."""
    for path in (detail_path, os.path.join(ROOT, "gpurun_out", "bench_detail.json")):
        try:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, "w") as fh:
                json.dump(line, fh, indent=1)
            line.setdefault("detail_file", os.path.relpath(detail_path, ROOT))
        except OSError as e:
            print(f"bench.py: cannot write {path}: {e}", file=sys.stderr)
    text = json.dumps(compact(line), allow_nan=False, separators=(",", ":"))
    back = json.loads(text)
    if len(text.encode()) > COMPACT_LIMIT or "\n" in text or back["value"] is None:
        raise SystemExit(f"bench.py: compact line is {len(text.encode())} bytes (limit {COMPACT_LIMIT}) or malformed")
    sys.stdout.flush()
    print(text, flush=True)


if __name__ == "__main__":
    sys.exit(main())
