#!/usr/bin/env python3
"""Benchmark of the DiGA training hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N>1)

A step is one full DiGA warm-up iteration (EMA teacher update, ClassMix, student forward on 2B images,
teacher forward on 2B images, fused upsample+CE+distillation, backward, gradient all-reduce, fused SGD)
on BASELINE.json configs[1]: ResNet-101 DeepLabV2, B=8 source crops of 768x768 per GPU, synthetic inputs
already resident in HBM.  Rank 0 prints ONE JSON line; `value` is source crops/s over all ranks.

Arithmetic (`--precision`): the convolutions run either in exact fp32 on the fp32 matrix cores ("f32") or
with fp32 operands split into bf16 hi+lo and three bf16 MFMAs per product, fp32 accumulate ("bf16x3",
default: ~1e-5 relative per product, the whole network stays within the path's 1e-3 logit tolerance --
tests/test_gpu_conv.py).  Everything else is fp32.  The other mode is timed too (2 steps) and reported
under "other_precision".
"""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")   # only the two SE linears reach a vendor library

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
F32_MFMA_PEAK_TFLOPS = 157.3     # v_mfma_f32_32x32x2_f32 dense peak
BF16_MFMA_PEAK_TFLOPS = 2500.0   # v_mfma_f32_32x32x16_bf16 dense peak
SERIAL_STEPS = 2          # steps of the serialised-stream pass that times kernels for the roofline
FWD_GFLOP_768 = 1232.9           # model forward, one 768x768 image
FWD_GFLOP_256 = 142.6

CONFIGS = {
    # name: (arch, batch per GPU, H, W, label block, description)
    "c2": ("RESNET101", 8, 768, 768, 32,
           "configs[1]: ResNet-101 DeepLabV2 DiGA warm-up (student + EMA teacher, KL distill), synthetic "
           "GTA5-shape 768x768, batch 8 per GPU"),
    "c1": ("TINY", 2, 256, 256, 16, "configs[0] stand-in: small-backbone DeepLab, 2x256x256 warm-up step"),
    "c4": ("RESNET101", 8, 512, 1024, 32,
           "configs[3]: self-training step (centroid pseudo-labeler + two ClassMix blocks + centroid EMA), synthetic "
           "Cityscapes-shape 512x1024, 8 source + 8 target crops per GPU"),
}
DTYPE = {"f32": "f32",
         "bf16x3": "bf16x3 (conv operands = f32 split into bf16 hi+lo, 3 bf16 MFMAs per product, f32 accumulate; "
                   "all other kernels f32)"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=None, help="override crops per GPU (debug only)")
    ap.add_argument("--size", type=int, nargs=2, default=None, help="override crop H W (debug only)")
    ap.add_argument("--precision", default="bf16x3", choices=["f32", "bf16x3"])
    ap.add_argument("--no-other-precision", action="store_true", help="skip the short run of the other arithmetic")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="internal: run the CPU leg and print its JSON")
    ap.add_argument("--no-prof", action="store_true", help="do not bracket kernel families with HIP events")
    ap.add_argument("--serial-streams", action="store_true",
                    help="run the whole step on one stream (no teacher / weight-gradient side stream): the mode the "
                         "per-kernel roofline durations are measured in; profiles/*_serial_* are rocprofv3 runs of it")
    return ap.parse_args()


def cpu_baseline():
    """The oracle's warm-up step (PyTorch-CPU restatement of the reference, pinned by tests/golden) timed on
    this box's host cores: ResNet-101, B=2 crops of 256x256, second of two steps.  Reported in 768x768-crop
    units by the forward-FLOP ratio."""
    from oracle import detweights, synth
    from oracle import step as ost
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 32))            # beyond ~32 threads torch's CPU convs stop scaling at this size
    torch.set_num_threads(cores)
    tr = ost.Trainer(detweights.state_dict(), detweights.state_dict())
    rng = random.Random(5)
    dt = None
    for it in range(2):
        batch = synth.warmup_batch(300 + it, 2, 256, 256, block=16)
        t0 = time.perf_counter()
        tr.warmup_step(it, *batch, rng)
        dt = time.perf_counter() - t0
    crops256 = 2.0 / dt
    return {"value": crops256 * FWD_GFLOP_256 / FWD_GFLOP_768, "unit": "crops/s", "cores": cores, "kind": "port",
            "sample": f"oracle warm-up step, ResNet-101, B=2 crops of 256x256 fp32, 2nd of 2 steps: {dt:.2f} s/step "
                      f"= {crops256:.3f} 256x256-crops/s; scaled to 768x768 crops by forward FLOPs "
                      f"({FWD_GFLOP_256}/{FWD_GFLOP_768} GFLOP)"}


def cpu_baseline_subprocess(limit_s=240):
    """Run the CPU leg in a child process (started before this process touches the GPU) so that a slow or
    memory-hungry host cannot take the GPU measurement down with it."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"], capture_output=True,
                           text=True, timeout=limit_s)
        for ln in reversed(r.stdout.splitlines()):
            if ln.startswith("{"):
                return json.loads(ln)
        return {"value": None, "unit": "crops/s", "cores": None, "kind": "port",
                "sample": f"CPU leg failed (rc={r.returncode}): {r.stderr[-300:]}"}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "crops/s", "cores": None, "kind": "port",
                "sample": f"CPU leg did not finish two B=2 256x256 oracle steps within {limit_s} s"}


def run_steps(a, precision, steps, warmup, rank, world, dev, prof):
    """Build student/teacher, run `warmup` + `steps` warm-up iterations; returns (seconds for `steps` = max over
    ranks, kernel families, last losses, parameter counts)."""
    from diga_amd import _lib, ddp, synthetic
    from diga_amd.model import seg_model_noaux as sm
    from diga_amd.model.model_noaux import SegModel
    from diga_amd.train_step import DigaTrainer

    _lib.call("diga_set_conv_math", 1 if precision == "bf16x3" else 0)
    arch_name, B, H, W, block, _ = CONFIGS[a.config]
    B = a.batch or B
    if a.size:
        H, W = a.size
    arch = getattr(sm, arch_name)
    torch.manual_seed(0)                       # identical random-init weights on every rank
    student, teacher = SegModel(arch=arch).to(dev), SegModel(arch=arch).to(dev)
    ddp.broadcast_module(student)
    teacher.train()
    rng = random.Random(1234 + rank)           # ClassMix class choice differs per rank, reproducibly
    tr = DigaTrainer(student, teacher, rng=rng)
    if a.config == "c4":
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

    it = 0
    for _ in range(warmup):
        one_step(it)
        it += 1
    torch.cuda.synchronize()
    barrier()
    if prof:
        _lib.call("diga_prof_reset")
        _lib.call("diga_prof_enable", 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = one_step(it)
        it += 1
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if prof:
        _lib.call("diga_prof_enable", 0)
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    losses = {k: float(v) for k, v in out.items()}
    if not all(v == v and abs(v) < 1e6 for v in losses.values()):
        raise SystemExit(f"non-finite loss in the timed region: {losses}")
    def query(nsteps):
        fam = {}
        for tag in _lib.PROF_TAGS:
            n, ms = _lib.prof_query(tag)
            if n:
                fam[tag] = {"launches": n, "avg_ms": ms / n, "ms_per_step": ms / nsteps}
        return fam

    families_overlapped = query(steps) if prof else {}
    families = families_overlapped
    if prof and not a.serial_streams:
        # Kernel durations for the roofline: the timed region runs the teacher forward and the weight gradients on a
        # second stream, so the HIP events of a launch there span other kernels sharing the GPU.  Two more steps
        # with the streams serialised time every kernel on its own (all ranks, the step has a collective).
        saved = {k: os.environ.get(k) for k in ("DIGA_TEACHER_STREAM", "DIGA_WGRAD_STREAM")}
        os.environ.update(DIGA_TEACHER_STREAM="0", DIGA_WGRAD_STREAM="0")
        try:
            one_step(it)
            it += 1
            torch.cuda.synchronize()
            _lib.call("diga_prof_reset")
            _lib.call("diga_prof_enable", 1)
            for _ in range(SERIAL_STEPS):
                one_step(it)
                it += 1
            torch.cuda.synchronize()
            _lib.call("diga_prof_enable", 0)
            families = query(SERIAL_STEPS)
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        barrier()
    del tr, student, teacher, batch, one_step
    torch.cuda.empty_cache()
    return float(t), (families, families_overlapped), losses, counts, (B, H, W, arch_name)


def rooflines(a, precision, families, steps, counts, geom):
    """Dominant kernel: the forward implicit-GEMM convolution (MFMA-bound).  Algorithmic work of its launches in
    one step = model forward FLOPs (1232.9 GFLOP per 768x768 image, scaled by area) x 2B
    student + 2B teacher images; achieved = that / the summed HIP-event durations of those launches."""
    B, H, W, arch_name = geom
    n_trainable, n_params = counts
    roof, other = None, {}
    fwd_gflop = FWD_GFLOP_768 * (H * W) / (768.0 * 768.0)
    # bf16x3: three bf16 MFMAs (2.5 PFLOP/s dense) per algorithmic multiply-add
    peak = F32_MFMA_PEAK_TFLOPS if precision == "f32" else BF16_MFMA_PEAK_TFLOPS / 3.0
    # HBM traffic per launch comes from separate rocprofv3 --pmc passes of this same command (FETCH_SIZE doubled
    # as MI355X_MICROARCH.md prescribes), condensed by tools/summarize_prof.py into profiles/
    pmc = None
    try:
        tag = "r01_f32" if precision == "f32" else "r01_bf16x3_serial"      # rocprofv3 --pmc runs of bench.py --serial-streams
        with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.json")) as fh:
            pmc = json.load(fh)["kernels"]
    except (OSError, ValueError, KeyError):
        pass
    if "conv_fwd" in families and arch_name == "RESNET101":
        fam = families["conv_fwd"]
        imgs_fwd = (2 * (2 * B)) if a.config != "c4" else (2 * (3 * B))      # student + teacher forward images
        flops_step = imgs_fwd * fwd_gflop * 1e9
        n_launch = fam["launches"] / steps
        ach = flops_step / (fam["ms_per_step"] * 1e-3) / 1e12
        kname = "conv_fwd_kernel" if precision == "f32" else "conv_fwd_x3w_kernel + conv_fwd_x3t_kernel"
        traffic = None
        if pmc and (B, H, W) == (8, 768, 768):
            # launch-weighted mean over the family's kernels (the 128-column instantiations carry > 95 % of its time;
            # the same kernels also serve backward-data, whose launches are in the PMC averages)
            names = ("diga::conv_fwd_kernel<2",) if precision == "f32" else ("diga::conv_fwd_x3w_kernel<2", "diga::conv_fwd_x3t_kernel<2")
            ent = [v for k, v in pmc.items() if k.startswith(names)]
            nl = sum(v["launches"] for v in ent)
            traffic = sum(v["hbm_bytes_per_launch_corrected"] * v["launches"] for v in ent) / nl if nl else None
        roof = {"kernel": f"{kname} (implicit-GEMM convolution on the "
                          f"{'fp32' if precision == 'f32' else 'bf16'} matrix cores; all forward-conv launches of a step"
                          f"{'' if precision == 'f32' else ', split-twin conversions of their inputs counted in elementwise'})",
                "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic,
                "algorithmic_flops_per_launch": flops_step / n_launch, "avg_launch_ms": fam["avg_ms"],
                "launches_per_step": n_launch}
        for tag in ("conv_bwd_data", "conv_bwd_weight"):
            if tag in families:        # backward: one pass each over the student's 2B images
                f = ((2 * B) if a.config != "c4" else (3 * B)) * fwd_gflop * 1e9
                v = f / (families[tag]["ms_per_step"] * 1e-3) / 1e12
                other[tag] = {"bound": "mfma", "unit": "TFLOP/s", "peak": peak, "achieved": v, "frac": v / peak}
    for tag, nbytes in (("sgd", 20.0 * n_trainable), ("ema", 12.0 * n_params),
                        ("classmix_paste", 44.0 * B * H * W), ("classmix_hist", 8.0 * B * H * W)):
        if tag in families:
            v = nbytes / (families[tag]["avg_ms"] * 1e-3) / 1e9
            other[tag] = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "achieved": v, "frac": v / HBM_PEAK_GBS,
                          "algorithmic_bytes_per_launch": nbytes}
    return roof, other


def main():
    a = parse()
    if a.cpu_baseline_only:
        print(json.dumps(cpu_baseline()), flush=True)
        return
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    cpu_line = None
    if world_env == 1 and not a.no_cpu_baseline:
        cpu_line = cpu_baseline_subprocess()
    from diga_amd import ddp
    rank, world, local = ddp.init_from_env()
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N>1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the hot path has no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    if a.serial_streams:
        os.environ.update(DIGA_TEACHER_STREAM="0", DIGA_WGRAD_STREAM="0")
    dt, (families, families_ov), losses, counts, geom = run_steps(a, a.precision, a.steps, a.warmup, rank, world, dev,
                                                                  not a.no_prof)
    fam_steps = a.steps if (a.serial_streams or a.no_prof) else SERIAL_STEPS
    B, H, W, _ = geom
    other_line = None
    if not a.no_other_precision and world == 1:
        oprec = "f32" if a.precision == "bf16x3" else "bf16x3"
        odt, (ofam, _), olosses, _, _ = run_steps(a, oprec, 2, 1, rank, world, dev, not a.no_prof)
        oroof, _ = rooflines(a, oprec, ofam, 2 if (a.serial_streams or a.no_prof) else SERIAL_STEPS, counts, geom)
        other_line = {"dtype": DTYPE[oprec], "value": world * B * 2 / odt, "unit": "crops/s", "steps": 2, "warmup": 1,
                      "ms_per_step": 1e3 * odt / 2, "roofline": oroof, "losses_last_step": olosses}

    if rank == 0:
        roof, other = rooflines(a, a.precision, families, fam_steps, counts, geom)
        if roof is not None:
            roof["measured"] = ("HIP events around every launch, on the launch stream; "
                                + ("timed region (single stream)" if a.serial_streams else
                                   f"{SERIAL_STEPS} extra steps with the teacher / weight-gradient side streams serialised "
                                   "(in the timed region they overlap other kernels; see kernel_families_overlapped)"))
        line = {
            "metric": ("768x768 19-class crops/sec (DiGA warm-up step)" if a.config != "c4" else
                       "512x1024 19-class (source,target) crop pairs/sec (DiGA self-training step)"),
            "value": world * B * a.steps / dt, "unit": "crops/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": DTYPE[a.precision], "data": "synthetic",
            "config": {"workload": CONFIGS[a.config][5], "global_batch": world * B, "crop": [H, W],
                       "parallelism": f"dp{world}",
                       "images_per_step_per_gpu": {"student_fwd_bwd": 2 * B, "teacher_fwd": 2 * B}},
            "roofline": roof, "roofline_other_kernels": other, "cpu_baseline": cpu_line,
            "other_precision": other_line, "kernel_families": families,
            "kernel_families_overlapped": None if a.serial_streams else families_ov, "losses_last_step": losses,
            "model_tflop_per_step_per_gpu": (2 * B) * 4 * (FWD_GFLOP_768 * (H * W) / (768.0 * 768.0)) / 1e3,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
