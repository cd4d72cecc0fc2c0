# Round-5 profile of the SegFormer-B5 leg (BASELINE configs[4] on one GPU): kernel stats of the serialised eager step, and the
# kernel trace of the step as it is timed (replayed from its HIP graph, teacher forward + weight gradients on forked streams).
#   gpurun --timeout 900 -- 'bash tools/profile_round5_c5.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=r05
B="python3 $R/bench.py --lean --no-prof --warmup 2 --config c5"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_c5_stats -- $B --no-graph --serial-streams --steps 3 > $R/gpurun_out/pc5a.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_c5_graph -- $B --graph --steps 3 > $R/gpurun_out/pc5b.log 2>&1
cd $R
mkdir -p gpurun_out/profiles_${T}
find gpurun_out/p_c5_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/profiles_${T}/${T}_c5_segformer_serial_kernel_stats.csv
find gpurun_out/p_c5_graph -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/profiles_${T}/${T}_c5_segformer_graph_kernel_stats.csv
python3 tools/diag/overlap_timeline.py gpurun_out/p_c5_graph > gpurun_out/profiles_${T}/${T}_c5_graph_overlap_timeline.txt 2>&1
tail -3 gpurun_out/pc5a.log; tail -3 gpurun_out/pc5b.log
rm -rf gpurun_out/p_c5_stats gpurun_out/p_c5_graph
cat gpurun_out/profiles_${T}/${T}_c5_graph_overlap_timeline.txt | tail -40
