#!/usr/bin/env python3
"""Diagnostic: which tensors differ between the eager and the graph-replayed warm-up step after N steps (TINY model)."""
import os, random, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from diga_amd.model import seg_model_noaux as sm
from diga_amd.model.model_noaux import SegModel
from diga_amd.train_step import DigaTrainer
from oracle import deeplab as od, detweights, synth

nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
res = []
for graph in (False, True):
    def make():
        m = SegModel(arch=sm.TINY)
        m.load_state_dict(detweights.state_dict(od.TINY))
        m.final.head[0].p = 0.0
        return m.cuda()
    s, t = make(), make()
    t.train()
    tr = DigaTrainer(s, t, rng=random.Random(11), graph=graph)
    grads = None
    for it in range(nsteps):
        batch = [x.cuda() for x in synth.warmup_batch(900 + it, 2, 96, 128, block=16)]
        out = tr.warmup_step(it, *batch)
        print(graph, it, float(out["ce"]), float(out["distil"]))
    torch.cuda.synchronize()
    res.append(({k: v.clone() for k, v in s.state_dict().items()}, {k: v.clone() for k, v in t.state_dict().items()},
                {n: p.grad.clone() for n, p in s.named_parameters() if p.grad is not None}))
for name, idx in (("student", 0), ("teacher", 1), ("grad", 2)):
    bad = [(k, float((res[0][idx][k].float() - res[1][idx][k].float()).abs().max())) for k in res[0][idx] if k in res[1][idx] and not torch.equal(res[0][idx][k], res[1][idx][k])]
    print(name, "differing tensors:", len(bad), "of", len(res[0][idx]), "equal:", [k for k in res[0][idx] if k in res[1][idx] and torch.equal(res[0][idx][k], res[1][idx][k])][:8] if name == "grad" else "")
    for k, e in bad[:12] + [b for b in bad if b[0].startswith("final")]:
        print("   ", k, e, "rel", e / max(float(res[0][idx][k].float().abs().max()), 1e-30))
