#!/bin/bash
# Round 6: what bounds the named bandwidth kernels (centroid weights, consensus, class sums, label histogram, upsample loss):
# instruction-issue counters of tools/bench_bw_kernels.py, one pass per counter group (counters only: no tracing domains).
#   gpurun -- 'bash tools/diag/r06_bw_pmc.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/p_bw_a -- python3 $R/tools/bench_bw_kernels.py $R/gpurun_out/bw_pmc_run.json > $R/gpurun_out/p_bw_a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES --output-format csv -d $R/gpurun_out/p_bw_b -- python3 $R/tools/bench_bw_kernels.py $R/gpurun_out/bw_pmc_run.json > $R/gpurun_out/p_bw_b.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/p_bw_c -- python3 $R/tools/bench_bw_kernels.py $R/gpurun_out/bw_pmc_run.json > $R/gpurun_out/p_bw_c.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/p_bw_d -- python3 $R/tools/bench_bw_kernels.py $R/gpurun_out/bw_pmc_run.json > $R/gpurun_out/p_bw_d.log 2>&1
cd $R
python3 - <<'PY' > gpurun_out/r06_bw_kernels_pmc_summary.json
import csv, glob, collections, json
want = ("centroid_weights_kernel", "argmax_consensus", "class_sums_kernel", "class_ids_kernel", "label_hist256_kernel", "classmix_paste_kernel",
        "upsample_loss_cells_kernel", "ce2d_kernel", "distill_kernel", "color_aug_kernel")
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for d in ("p_bw_a", "p_bw_b", "p_bw_c", "p_bw_d"):
    for f in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("diga::", "")
            if not any(w in k for w in want):
                continue
            k = k[:70]
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
out = {}
for k, v in agg.items():
    m = {c: x / cnt[(k, c)] for c, x in v.items()}
    r = dict(m)
    if "FETCH_SIZE" in m: r["hbm_read_MB"] = 2.0 * m["FETCH_SIZE"] * 1024.0 / 1e6      # FETCH_SIZE in KB, doubled on gfx950 (MI355X_MICROARCH.md; calibrated for 16-byte-per-lane streams)
    if "WRITE_SIZE" in m: r["hbm_write_MB"] = m["WRITE_SIZE"] * 1024.0 / 1e6
    if "SQ_ACTIVE_INST_VALU" in m and "SQ_BUSY_CYCLES" in m and m["SQ_BUSY_CYCLES"]:
        r["valu_active_share_of_busy"] = m["SQ_ACTIVE_INST_VALU"] / m["SQ_BUSY_CYCLES"]
    out[k] = r
print(json.dumps(out, indent=1))
PY
rm -rf gpurun_out/p_bw_a gpurun_out/p_bw_b gpurun_out/p_bw_c gpurun_out/p_bw_d
head -c 3000 gpurun_out/r06_bw_kernels_pmc_summary.json
