cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python3 $R/bench.py --lean --no-prof --steps 2 --warmup 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_stats_serial -- $B --serial-streams --steps 3 > $R/gpurun_out/p1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_stats_overlap -- $B --steps 3 > $R/gpurun_out/p2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/p_fetch -- $B --serial-streams --steps 1 > $R/gpurun_out/p3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/p_write -- $B --serial-streams --steps 1 > $R/gpurun_out/p4.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/p_mfma -- $B --serial-streams --steps 1 > $R/gpurun_out/p5.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_f32_stats -- $B --serial-streams --precision f32 --steps 3 > $R/gpurun_out/p6.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/p_f32_fetch -- $B --serial-streams --precision f32 --steps 1 > $R/gpurun_out/p7.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/p_f32_write -- $B --serial-streams --precision f32 --steps 1 > $R/gpurun_out/p8.log 2>&1
cd $R
# the raw traces are large: keep the per-kernel CSV summaries only
find gpurun_out/p_stats_serial gpurun_out/p_stats_overlap gpurun_out/p_f32_stats -name "*kernel_trace.csv" -delete
ls -la gpurun_out/p_*/*/ | head -40
