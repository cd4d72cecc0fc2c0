# Round-6 profiling recipe (run from the repo root on the GPU box via gpurun; raw output under gpurun_out/, condensed on the box by
# tools/summarize_prof.py / tools/summarize_mfma.py into profiles/<tag>_*, which are copied to gpurun_out/profiles_r06/ so that they
# come home -- see profiles/README.md).  Counters are collected in their own passes, never together with tracing domains.
#   gpurun --timeout 2400 -- 'bash tools/profile_round6.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=r06
B="python3 $R/bench.py --lean --no-prof --warmup 1"
# headline arithmetic (exact fp32): kernel stats of the serialised step, HBM traffic (separate passes), MFMA busy
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_f32_stats -- $B --serial-streams --precision f32 --steps 3 > $R/gpurun_out/p1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/p_f32_fetch -- $B --serial-streams --precision f32 --steps 1 > $R/gpurun_out/p2.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/p_f32_write -- $B --serial-streams --precision f32 --steps 1 > $R/gpurun_out/p3.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/p_f32_mfma -- $B --serial-streams --precision f32 --steps 1 > $R/gpurun_out/p4.log 2>&1
# the two-stream step as it is timed: kernel stats + the overlap timeline (what runs with no matrix-core kernel in flight)
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_f32_ov -- $B --precision f32 --steps 3 > $R/gpurun_out/p5.log 2>&1
# self-training step (c4), serialised: kernel stats and its own HBM traffic
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_c4_stats -- $B --config c4 --serial-streams --precision f32 --steps 2 > $R/gpurun_out/p6.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/p_c4_fetch -- $B --config c4 --serial-streams --precision f32 --steps 1 > $R/gpurun_out/p7.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/p_c4_write -- $B --config c4 --serial-streams --precision f32 --steps 1 > $R/gpurun_out/p8.log 2>&1
cd $R
python3 tools/summarize_prof.py ${T}_f32_serial gpurun_out/p_f32_stats gpurun_out/p_f32_fetch gpurun_out/p_f32_write
python3 tools/summarize_prof.py ${T}_c4_f32_serial gpurun_out/p_c4_stats gpurun_out/p_c4_fetch gpurun_out/p_c4_write
python3 tools/summarize_mfma.py ${T}_f32 gpurun_out/p_f32_mfma
python3 tools/diag/overlap_timeline.py gpurun_out/p_f32_ov > profiles/${T}_f32_overlap_timeline.txt 2>&1
mkdir -p gpurun_out/profiles_${T}
cp profiles/${T}_* gpurun_out/profiles_${T}/
find gpurun_out/p_f32_ov -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/profiles_${T}/${T}_f32_overlapped_kernel_stats.csv
rm -rf gpurun_out/p_f32_stats gpurun_out/p_f32_fetch gpurun_out/p_f32_write gpurun_out/p_f32_mfma gpurun_out/p_f32_ov gpurun_out/p_c4_stats gpurun_out/p_c4_fetch gpurun_out/p_c4_write
ls -la gpurun_out/profiles_${T}; cat profiles/${T}_f32_overlap_timeline.txt
