# Round-3 profiling recipe (run from the repo root on the GPU box via gpurun; outputs under gpurun_out/, summaries are then
# condensed into profiles/ by tools/summarize_prof.py / tools/summarize_mfma.py -- see profiles/README.md).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python3 $R/bench.py --lean --no-prof --steps 2 --warmup 1"
# headline arithmetic (exact fp32): kernel stats of the serialised step, HBM traffic (separate passes), MFMA busy
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_f32_stats -- $B --serial-streams --precision f32 --steps 3 > $R/gpurun_out/p1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/p_f32_fetch -- $B --serial-streams --precision f32 --steps 1 > $R/gpurun_out/p2.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/p_f32_write -- $B --serial-streams --precision f32 --steps 1 > $R/gpurun_out/p3.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/p_f32_mfma -- $B --serial-streams --precision f32 --steps 1 > $R/gpurun_out/p4.log 2>&1
# second arithmetic (split bf16)
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_x3_stats -- $B --serial-streams --precision bf16x3 --steps 3 > $R/gpurun_out/p5.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/p_x3_fetch -- $B --serial-streams --precision bf16x3 --steps 1 > $R/gpurun_out/p6.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/p_x3_write -- $B --serial-streams --precision bf16x3 --steps 1 > $R/gpurun_out/p7.log 2>&1
# MiT-B5 encoder forward + backward (BASELINE configs[4]): kernel stats, MFMA busy, HBM traffic
M="python3 $R/tools/bench_mit.py --batch 16"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_mit_stats -- $M --steps 2 > $R/gpurun_out/p8.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/p_mit_mfma -- $M --steps 1 > $R/gpurun_out/p9.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/p_mit_fetch -- $M --steps 1 > $R/gpurun_out/p10.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/p_mit_write -- $M --steps 1 > $R/gpurun_out/p11.log 2>&1
cd $R
# the raw traces are large: keep the per-kernel CSV summaries and the counter files only
find gpurun_out/p_f32_stats gpurun_out/p_x3_stats gpurun_out/p_mit_stats -name "*kernel_trace.csv" -delete
find gpurun_out/p_* -name "*.csv" -size +30M -delete
du -sh gpurun_out/p_* | tail -12
