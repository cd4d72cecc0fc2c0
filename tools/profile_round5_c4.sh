# Round-5 kernel stats of the self-training leg (c4), serialised (one stream, one-backward form): refresh after the step driver's
# three-stream form became the default (bench.py --serial-streams now also switches the third stream off).
#   gpurun --timeout 900 -- 'bash tools/profile_round5_c4.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=r05
B="python3 $R/bench.py --lean --no-prof --warmup 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_c4_stats -- $B --config c4 --serial-streams --precision f32 --steps 2 > $R/gpurun_out/p6.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_c4_ov -- $B --config c4 --precision f32 --steps 3 > $R/gpurun_out/p6b.log 2>&1
cd $R
mkdir -p gpurun_out/profiles_${T}
find gpurun_out/p_c4_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/profiles_${T}/${T}_c4_f32_serial_kernel_stats.csv
find gpurun_out/p_c4_ov -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/profiles_${T}/${T}_c4_f32_overlapped_kernel_stats.csv
python3 tools/diag/overlap_timeline.py gpurun_out/p_c4_ov > gpurun_out/profiles_${T}/${T}_c4_f32_overlap_timeline.txt 2>&1
tail -2 gpurun_out/p6.log; tail -2 gpurun_out/p6b.log
rm -rf gpurun_out/p_c4_stats gpurun_out/p_c4_ov
cat gpurun_out/profiles_${T}/${T}_c4_f32_overlap_timeline.txt | tail -12
