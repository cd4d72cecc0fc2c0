#!/usr/bin/env python3
"""Diagnostic: where the HOST time of a graph-replayed small-backbone step goes (cProfile over 30 steps)."""
import cProfile, os, pstats, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from diga_amd import synthetic
from diga_amd.model import seg_model_noaux as sm
from diga_amd.model.model_noaux import SegModel
from diga_amd.train_step import DigaTrainer
dev = torch.device("cuda")
torch.manual_seed(0)
s, t = SegModel(arch=sm.TINY).to(dev), SegModel(arch=sm.TINY).to(dev)
t.train()
tr = DigaTrainer(s, t, rng=random.Random(1), graph=True)
batch = synthetic.warmup_batch(1, 2, 256, 256, block=16, device=dev)
for i in range(3):
    tr.warmup_step(i, *batch); tr.prefetch_classmix(batch[3])
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for i in range(3, 33):
    tr.warmup_step(i, *batch); tr.prefetch_classmix(batch[3])
pr.disable()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / 30 * 1e3)
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
