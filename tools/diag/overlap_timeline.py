#!/usr/bin/env python3
"""How much of the step's bandwidth-bound kernel time runs under matrix-core kernels of the other stream: from a rocprofv3
--kernel-trace CSV of the default two-stream step (bench.py --lean --no-prof --steps 3 --warmup 2) take the last step's kernels,
classify them (MFMA = the convolution / GEMM kernels, OTHER = everything else) and integrate over time.

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/p_tl -- python3 $R/bench.py --lean --no-prof --steps 3 --warmup 2
    python tools/diag/overlap_timeline.py gpurun_out/p_tl"""
import csv
import glob
import os
import sys

d = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the optimizer kernel closes a step: take the span between the last two sgd launches
sgd = [i for i, r in enumerate(rows) if "sgd_multi_kernel" in r[2]]
a, b = sgd[-2] + 1, sgd[-1] + 1
step = rows[a:b]
t0, t1 = step[0][0], max(r[1] for r in step)
MFMA = ("conv_fwd", "conv_wgrad", "gemm_f32_persistent", "gemm_nt_kernel", "gemm_tn_kernel", "attn_fwd", "attn_bwd")   # (the last four: the MiT leg)
ev = []
for s, e, n in step:
    k = 0 if any(m in n for m in MFMA) else 1
    ev.append((s, 1, k))
    ev.append((e, -1, k))
ev.sort()
cnt = [0, 0]
last = t0
acc = {"mfma only": 0, "other only": 0, "both": 0, "idle": 0}
for t, dlt, k in ev:
    span = t - last
    if span > 0:
        key = "both" if cnt[0] and cnt[1] else "mfma only" if cnt[0] else "other only" if cnt[1] else "idle"
        acc[key] += span
    cnt[k] += dlt
    last = t
# the same split into the forward part (up to the fused loss kernel) and the backward part
loss = [r for r in step if "upsample_loss_cells" in r[2]]
if loss:
    tl = loss[0][0]
    for name, lo, hi in (("forward", t0, tl), ("backward", tl, t1)):
        c = [0, 0]
        lastt = lo
        a2 = {"mfma only": 0, "other only": 0, "both": 0, "idle": 0}
        for t, dlt, k in ev:
            tt = min(max(t, lo), hi)
            span = tt - lastt
            if span > 0:
                key = "both" if c[0] and c[1] else "mfma only" if c[0] else "other only" if c[1] else "idle"
                a2[key] += span
            c[k] += dlt
            lastt = tt
        print(f"  {name} {(hi - lo) / 1e6:.1f} ms: " + ", ".join(f"{k}: {v / 1e6:.1f}" for k, v in a2.items()))
    # which 'other' kernels run with no MFMA kernel in flight (top 8 by exposed time)
    import collections
    mf = sorted((s_, e_) for s_, e_, n in step if any(m in n for m in MFMA))
    merged = []
    for s_, e_ in mf:
        if merged and s_ <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], e_)
        else:
            merged.append([s_, e_])
    exp = collections.Counter()
    import bisect
    starts = [m[0] for m in merged]
    for s_, e_, n in step:
        if any(m in n for m in MFMA):
            continue
        covered = 0
        i = max(bisect.bisect_right(starts, s_) - 1, 0)
        while i < len(merged) and merged[i][0] < e_:
            covered += max(0, min(e_, merged[i][1]) - max(s_, merged[i][0]))
            i += 1
        exp[n.split("(")[0][-60:]] += (e_ - s_) - covered
    print("  exposed (no MFMA kernel in flight), per kernel: " + "; ".join(f"{k} {v / 1e6:.1f} ms" for k, v in exp.most_common(8)))
tot = t1 - t0
dur = [sum(e - s for s, e, n in step if any(m in n for m in MFMA)), sum(e - s for s, e, n in step if not any(m in n for m in MFMA))]
print(f"step span {tot / 1e6:.1f} ms, {len(step)} kernels; summed durations: MFMA kernels {dur[0] / 1e6:.1f} ms, others {dur[1] / 1e6:.1f} ms")
print("wall time with " + ", ".join(f"{k}: {v / 1e6:.1f} ms" for k, v in acc.items()))
