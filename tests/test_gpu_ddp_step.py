"""Two data-parallel ranks (one process each, gloo, both on the one GPU of the test box) run a DiGA warm-up step and a
self-training step on their own shards; the result must equal a single-process emulation:
  * student after the step == one optimizer step on the AVERAGE of the two shards' gradients (each shard's forward /
    backward with its own BatchNorm batch statistics and its own ClassMix draw, as every rank of the reference would);
    the all-reduce sums two addends, the 1/world lives in the SGD kernel's grad_scale -- bit for bit;
  * every rank holds the same parameters afterwards;
  * the centroid bank == the order-dependent EMA (G5/calc_centroids.py:147-156) applied to the class sums of rank 0's
    images then rank 1's (rank-major, image-major, class-minor), target pass before source pass -- checked against
    the CPU oracle on the per-rank sums the workers sent into the all-gather."""
import os
import random
import socket
import subprocess
import sys

import pytest
import torch

from oracle import centroids as oc
from oracle import deeplab as od
from oracle import detweights, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda"


def _run_workers(tmp, math, **extra_env):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   DIGA_TEST_MATH=str(math), **extra_env)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ddp_step_worker.py"), str(tmp)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)[-3000:]
    return [torch.load(os.path.join(tmp, f"rank{r}.pt")) for r in range(2)]


@pytest.mark.timeout(900)
def test_two_rank_steps_equal_single_process_emulation(tmp_path, conv_math):
    from diga_amd.calc_centroids import Class_Features
    from diga_amd.model import seg_model_noaux as sm
    from diga_amd.model.model_noaux import SegModel
    from diga_amd.train_step import DigaTrainer
    from diga_amd.util import loss as L
    from diga_amd.util import utils as U
    r0, r1 = _run_workers(tmp_path, conv_math)
    # every rank ends with the same student
    for tag in ("after1", "after2"):
        for k in r0[tag]:
            assert torch.equal(r0[tag][k], r1[tag][k]), f"{tag}: ranks disagree on {k}"

    # ---- single-process emulation of the warm-up step: per-shard gradients, summed, grad_scale = 1/2
    def make():
        m = SegModel(arch=sm.TINY)
        m.load_state_dict(detweights.state_dict(od.TINY))
        m.final.head[0].p = 0.0
        return m.to(DEV)

    student, teacher = make(), make()
    teacher.train()
    tr = DigaTrainer(student, teacher, rng=random)
    tr.opt.grad_scale = 0.5
    tr._begin(0)
    tr.opt.zero_grad(set_to_none=True)
    bn_state = {k: v.clone() for k, v in student.state_dict().items() if "running" in k or "num_batches" in k}
    t_state = {k: v.clone() for k, v in teacher.state_dict().items() if "running" in k or "num_batches" in k}
    ces = []
    for rank in range(2):
        student.load_state_dict(bn_state, strict=False)          # each rank starts from the same buffers
        teacher.load_state_dict(t_state, strict=False)
        x, x_aug, rec, lab = (t.to(DEV) for t in synth.warmup_batch(900 + rank, 2, 96, 128, block=16))
        with torch.no_grad():
            mix, _ = U.classmix(rec, x_aug, lab, random.Random(500 + rank))
            cat = torch.cat([x, mix])
            t_lr = teacher(cat)[2]
        s_lr = student(cat)[2]
        total, ce, di = L.upsample_ce_distill(s_lr, t_lr, lab, 1.0, 0.5, 0.5)
        total.backward()                                         # accumulates: p.grad = g(shard 0) + g(shard 1)
        ces.append(float(ce))
    tr.opt.step()
    torch.cuda.synchronize()
    assert ces[0] == pytest.approx(r0["log1"]["ce"], rel=1e-6) and ces[1] == pytest.approx(r1["log1"]["ce"], rel=1e-6)
    for k, v in student.named_parameters():
        assert torch.equal(v.detach().cpu(), r0["after1"][k]), f"warm-up step: 2-rank result differs from the emulation at {k}"

    # ---- centroid bank: oracle's sequential EMA over the rank-major concatenation of what the ranks sent
    assert len(r0["sent"]) == 2 and len(r1["sent"]) == 2            # target pass, then source pass
    cents = torch.randn((19, 256), generator=torch.Generator().manual_seed(7))
    nums = torch.zeros(19)
    hw = 13 * 17                                                    # low-res map of a 96x128 crop
    for ps in range(2):
        sums = torch.cat([r0["sent"][ps][0], r1["sent"][ps][0]])    # [2 ranks x 2 images, 19, 256], rank-major
        counts = torch.cat([r0["sent"][ps][1], r1["sent"][ps][1]])
        vectors, ids = [], []
        for n in range(sums.shape[0]):
            for t in range(19):
                c = int(counts[n, t])
                if c == 0 or c < 5:
                    continue
                vectors.append((sums[n, t] / float(hw)) / (float(c) / float(hw)))
                ids.append(t)
        oc.centroid_ema_apply(cents, nums, vectors, ids)
    for r in (r0, r1):
        assert torch.allclose(r["cents"], cents, rtol=1e-6, atol=1e-7)
        assert torch.equal(r["nums"], nums)
    assert float(nums.sum()) > 0                                    # the update did something
    cf = Class_Features(numbers=19)                                 # and hw above is the map size the library used
    f = torch.zeros((1, 256, 13, 17), device=DEV)
    assert cf._class_sums(f, torch.zeros((1, 19, 13, 17), device=DEV), labels_full=torch.zeros((1, 96, 128), dtype=torch.int64, device=DEV))[2] == hw


@pytest.mark.timeout(1500)
def test_two_rank_selftraining_step_same_in_all_three_forms(tmp_path, conv_math):
    """The self-training step under data parallelism in its three forms -- one backward pass with hook-driven buckets (the
    reference's form), cross-mixed forward / backward on a third stream, the whole target branch on a third stream (round 5; the two
    overlapped forms hold the hooks back: the first backward() leaves partial sums, the buckets leave after the join) -- gives
    the same student, centroid bank and losses on both ranks, bit for bit."""
    if conv_math != 0:
        pytest.skip("stream / reducer structure, not arithmetic: once (fp32) is enough -- three pairs of worker processes")
    runs = {}
    # (DIGA_C4_OVERLAP_GLOO: the overlapped forms are off under gloo by default -- slow there, not wrong)
    for tag, env in (("one_backward", dict(DIGA_C4_OVERLAP="0")), ("tail", dict(DIGA_C4_OVERLAP="2", DIGA_C4_OVERLAP_GLOO="1")),
                     ("target_branch", dict(DIGA_C4_OVERLAP="2", DIGA_C4_OVERLAP_GLOO="1", DIGA_TEST_FORCE_STREAMS="1"))):
        d = tmp_path / tag
        d.mkdir()
        runs[tag] = _run_workers(d, conv_math, **env)
    assert runs["one_backward"][0]["forms"] == [] and runs["tail"][0]["forms"] == ["_selftrain_tail_overlapped"]
    assert runs["target_branch"][0]["forms"] == ["_selftrain_rest_overlapped"]
    ref = runs["one_backward"]
    for tag in ("tail", "target_branch"):
        for r in range(2):
            assert runs[tag][r]["log2"] == ref[r]["log2"], (tag, r, runs[tag][r]["log2"], ref[r]["log2"])
            assert torch.equal(runs[tag][r]["cents"], ref[r]["cents"]) and torch.equal(runs[tag][r]["nums"], ref[r]["nums"])
            for k in ref[r]["after2"]:
                assert torch.equal(runs[tag][r]["after2"][k], ref[r]["after2"][k]), (tag, r, k)
        for k in runs[tag][0]["after2"]:
            assert torch.equal(runs[tag][0]["after2"][k], runs[tag][1]["after2"][k]), (tag, k)


@pytest.mark.timeout(1500)
def test_rccl_single_rank_runs_the_production_reducer_path():
    """RCCL code in the driver's GPU test (round 6): a child process brings up a ONE-rank "nccl" process group on the box's GPU with
    the reducer forced active (config.ddp_single_rank) and runs the warm-up step and the self-training step in all three forms --
    hook-driven buckets on RCCL's stream next to the teacher / weight-gradient side streams, gradients living in the bucket slices,
    the three-stream self-training form with the hooks held and per-bucket launches after the join, the class-sum all-gather --
    against the same steps without a reducer: bit-identical students, centroid banks and losses (tests/rccl_single_rank_worker.py;
    it exits non-zero on any difference).  What only > 1 rank can show -- that the sum over ranks is right -- is the gloo tests' job
    above (`test_two_rank_*`)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("DIGA_DDP_BACKEND", "WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_single_rank_worker.py")], env=env, capture_output=True, text=True,
                       timeout=1400, cwd=ROOT)
    print(r.stdout[-3000:])
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "rccl single-rank worker OK" in r.stdout and r.stdout.count("== plain run bit for bit") == 4


@pytest.mark.timeout(1500)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: one RCCL rank per device (the 1-GPU test box skips it)")
def test_bench_two_ranks_over_rccl():
    """The distributed path as the driver launches it, on real RCCL: `bench.py --gpus 2 --lean` (two worker processes, one per
    GPU, backend nccl) must finish, report two RCCL ranks, finite losses and -- checked inside bench.py by an all-gather of a
    parameter checksum -- bit-identical students on both ranks after the all-reduced steps; rank 0 prints the one compact line."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--lean", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=1400, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert len(lines[0]) < 4000
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["backend"] == "nccl" and d["ranks_agree"] is True
    assert d["config"]["global_batch"] == 16 and d["value"] > 0 and d["cpu_baseline"] is None


@pytest.mark.timeout(1500)
def test_bench_launcher_eight_ranks_gloo_toy():
    """Launcher smoke for the 8-GPU node the driver measures on, with what a 1-GPU box has: `bench.py --gpus 8` starts EIGHT worker
    processes (spawned before anything touches the GPU; LOCAL_WORLD_SIZE / OMP_NUM_THREADS set, every rank pinned to its own slice of
    the host's cores) over gloo, all sharing the one device, on the small backbone at 64 x 64: the process group comes up as 8 ranks,
    the hook-driven gradient buckets (gradients living in the buckets: ddp.GradReducer) all-reduce, every rank ends with the same
    student bit for bit (bench.py's own all-gather of a parameter checksum), rank 0 prints the one compact line, exit code 0."""
    import json
    env = dict(os.environ, DIGA_DDP_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--lean", "--config", "c1", "--batch", "1", "--size", "64", "64",
                        "--steps", "2", "--warmup", "1", "--no-graph"], capture_output=True, text=True, timeout=1400, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["rccl_ranks"] == 8 and d["backend"] == "gloo" and d["ranks_agree"] is True
    assert d["config"]["global_batch"] == 8 and d["value"] > 0
    detail = json.load(open(os.path.join(ROOT, "bench_detail.json")))
    host = detail["host"]
    assert host["local_world"] == 8 and host["omp_num_threads"] >= 1
    if (os.cpu_count() or 1) >= 8:
        assert host["pinned"] and host["n_cores"] >= 1
