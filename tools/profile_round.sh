# Round-4 profiling recipe (run from the repo root on the GPU box via gpurun; raw output under gpurun_out/, condensed on the box by
# tools/summarize_prof.py / tools/summarize_mfma.py into profiles/<tag>_*, which are copied to gpurun_out/profiles_r04/ so that they
# come home -- see profiles/README.md).  Counters are collected in their own passes, never together with tracing domains.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python3 $R/bench.py --lean --no-prof --warmup 1"
# headline arithmetic (exact fp32): kernel stats of the serialised step, HBM traffic (separate passes), MFMA busy
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_f32_stats -- $B --serial-streams --precision f32 --steps 3 > $R/gpurun_out/p1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/p_f32_fetch -- $B --serial-streams --precision f32 --steps 1 > $R/gpurun_out/p2.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/p_f32_write -- $B --serial-streams --precision f32 --steps 1 > $R/gpurun_out/p3.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/p_f32_mfma -- $B --serial-streams --precision f32 --steps 1 > $R/gpurun_out/p4.log 2>&1
# the two-stream step as it is timed (kernel stats only)
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_f32_ov -- $B --precision f32 --steps 3 > $R/gpurun_out/p5.log 2>&1
# self-training step (c4), serialised: kernel stats
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_c4_stats -- $B --config c4 --serial-streams --precision f32 --steps 2 > $R/gpurun_out/p6.log 2>&1
cd $R
python3 tools/summarize_prof.py r04_f32_serial gpurun_out/p_f32_stats gpurun_out/p_f32_fetch gpurun_out/p_f32_write
python3 tools/summarize_mfma.py r04_f32 gpurun_out/p_f32_mfma
mkdir -p gpurun_out/profiles_r04
cp profiles/r04_f32_serial_kernel_stats.csv profiles/r04_f32_serial_pmc_summary.json profiles/r04_f32_mfma_busy_summary.json gpurun_out/profiles_r04/
find gpurun_out/p_f32_ov -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/profiles_r04/r04_f32_overlapped_kernel_stats.csv
find gpurun_out/p_c4_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/profiles_r04/r04_c4_f32_serial_kernel_stats.csv
rm -rf gpurun_out/p_f32_stats gpurun_out/p_f32_fetch gpurun_out/p_f32_write gpurun_out/p_f32_mfma gpurun_out/p_f32_ov gpurun_out/p_c4_stats
ls -la gpurun_out/profiles_r04
