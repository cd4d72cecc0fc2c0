"""Round 6: what the weight-gradient side stream costs in device memory (reserved within 1.2 x allocated).

The caching allocator returns a block that was `record_stream`-ed to a side stream only once the HOST sees an event of that stream
complete; the host enqueues a backward pass far ahead of the device, so during backward none of those blocks came back and the main
stream's pool grew to 126 GB for tensors peaking at 104 GB (profiles/r06_memory_by_stream.txt).  `_lib.release_to_side` keeps the
tensors alive for `config.wgrad_hold` more layers instead and makes their stream wait ON THE DEVICE for the side stream's event before
dropping them: stream-ordered frees, no deferral.  Checked here: the mechanics, that the step's results do not move by a bit, and the
bound itself on BASELINE configs[1] at full size."""
import json
import os
import random
import subprocess
import sys

import pytest
import torch

from diga_amd import config

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_release_to_side_holds_then_releases_behind_an_event():
    from diga_amd import _lib
    dev = torch.device("cuda", 0)
    main = torch.cuda.current_stream(dev)
    _lib.side_overlap = True
    try:
        with config.override(wgrad_stream=True, wgrad_hold=2, wgrad_hold_batch=1):
            side = _lib.side_stream(dev)
            assert side is not None
            sums = []
            nbytes = 4 << 20
            torch.cuda.synchronize()

            def active():
                return torch.cuda.memory_stats(dev)["active_bytes.all.current"]
            base = active()
            for i in range(5):
                t = torch.full((1 << 20,), float(i), device=dev)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    sums.append(t.sum())                         # the side stream reads t
                _lib.release_to_side(side, (t,))
                del t
                assert _lib.held_for_side() == min(i + 1, 2)    # never more than `wgrad_hold` entries
                # a released block is free in stream order -- back in its pool the moment the queue drops it, not parked behind an event the
                # host has yet to see complete: only the held tensors (and the few bytes of the sums) are active
                assert abs(active() - base - min(i + 1, 2) * nbytes) < (64 << 10), (i, active() - base)
                again = torch.empty((1 << 20,), device=dev)      # ... and may be rewritten at once: its stream waited for the reader
                again.fill_(-1.0)
                del again
            _lib.join_side()
            assert _lib.held_for_side() == 0
            torch.cuda.synchronize()
            assert [float(v) for v in sums] == [float(i) * (1 << 20) for i in range(5)]
            del sums[:]
        with config.override(wgrad_stream=True, wgrad_hold=2, wgrad_hold_batch=3):   # released three at a time behind ONE wait
            seen = []
            for i in range(9):
                t = torch.full((1 << 20,), 1.0, device=dev)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    sums.append(t.sum())
                _lib.release_to_side(side, (t,))
                del t
                seen.append(_lib.held_for_side())
            assert seen == [1, 2, 3, 4, 2, 3, 4, 2, 3], seen
            _lib.join_side()
            assert _lib.held_for_side() == 0
        with config.override(wgrad_stream=True, wgrad_hold=0):      # 0: the allocator's record_stream, nothing held
            t = torch.ones((1 << 20,), device=dev)
            _lib.release_to_side(_lib.side_stream(dev), (t,))
            assert _lib.held_for_side() == 0
            _lib.join_side()
    finally:
        _lib.side_overlap = False
        _lib.join_side()


@pytest.mark.parametrize("arch", ["resnet101"])
def test_warmup_steps_do_not_move_by_a_bit_under_the_hold(arch, conv_math):
    """Three warm-up steps (student + teacher, ClassMix, SGD) with record_stream (hold 0), with the default hold and with everything held
    until the join: losses and every student tensor equal bit for bit.  (The first form of the hold also kept the returned weight
    gradient alive -- AccumulateGrad then CLONED it on the main stream while the side stream was still writing it; this test and
    tests/test_selftrain.py::test_selftrain_overlapped_tail_is_bit_identical are what catches that.)"""
    from diga_amd.model.model_noaux import SegModel
    from diga_amd.train_step import DigaTrainer
    from oracle import deeplab as od
    from oracle import detweights, synth

    def run(hold):
        cfg = config.DEFAULTS.replace(wgrad_stream=True, wgrad_hold=hold, conv_math=conv_math)

        def make():
            m = SegModel()
            m.load_state_dict(detweights.state_dict(od.RESNET101))
            m.final.head[0].p = 0.0
            return m.to("cuda")
        student, teacher = make(), make()
        teacher.train()
        tr = DigaTrainer(student, teacher, rng=random.Random(3), config=cfg)
        losses = []
        for it in range(3):
            batch = [t.to("cuda") for t in synth.warmup_batch(500 + it, 2, 160, 160, block=16)]
            losses.append({k: float(v) for k, v in tr.warmup_step(it, *batch).items()})
        torch.cuda.synchronize()
        return losses, {k: v.clone() for k, v in student.state_dict().items()}

    la, sa = run(0)
    for hold in (config.StepConfig().wgrad_hold, 1, 1000):
        lb, sb = run(hold)
        assert la == lb, (hold, la, lb)
        for k in sa:
            assert torch.equal(sa[k], sb[k]), (hold, k)


@pytest.mark.timeout(900)
def test_c2_reserved_memory_within_1p2_of_allocated():
    """BASELINE configs[1] (ResNet-101, B = 8 crops of 768 x 768, fp32, default streams) in a fresh process: the peak the caching
    allocator holds from the driver over the timed steps stays within 1.2 x the peak of the step's tensors.  (Round 5: 1.54 x.)"""
    torch.cuda.empty_cache()                      # (this process's cache of earlier tests must not crowd the child's 125 GB)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--lean", "--steps", "3", "--warmup", "2"],
                       capture_output=True, text=True, timeout=800, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    mem = d["peak_mem_gb"]
    print("\n[memory] C2 B=8 peak allocated %.1f GB, reserved %.1f GB (%.3f x), %.1f ms/step"
          % (mem["allocated"], mem["reserved"], mem["reserved"] / mem["allocated"], d["ms_per_step"]))
    assert mem["reserved"] <= 1.2 * mem["allocated"], mem
    assert mem["allocated"] < 112.0, mem            # the hold itself costs ~3 GB of live tensors (103.6 -> 106.3), not more


@pytest.mark.timeout(900)
def test_c4_three_stream_form_holds_no_more_than_the_one_backward_form_needs():
    """BASELINE configs[3]'s step (B = 8 + 8 at 512 x 1024, default = target branch on a third stream): per-stream pools keep its
    `reserved` above 1.2 x ITS OWN `allocated` -- it frees the first student graph while the second still runs, 106 GB at the high-water
    mark where the one-backward form has 146 GB -- but what it holds from the driver must stay within 1.2 x of what the one-backward form's
    tensors occupy (measured 169.9 GB against 1.2 x 145.6 = 174.7; with record_stream it was 210 GB)."""
    torch.cuda.empty_cache()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--lean", "--config", "c4", "--steps", "2", "--warmup", "2"],
                       capture_output=True, text=True, timeout=800, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    mem = d["peak_mem_gb"]
    print("\n[memory] c4 B=8+8 peak allocated %.1f GB, reserved %.1f GB, %.1f ms/step" % (mem["allocated"], mem["reserved"], d["ms_per_step"]))
    assert mem["reserved"] <= 1.2 * 145.6, mem
    assert mem["allocated"] < 112.0, mem
