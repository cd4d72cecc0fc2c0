# round-4 counters of the MiT-B5 encoder (forward + backward, 16 crops of 768x768) with the row-sliding depthwise kernels:
# kernel stats, FETCH_SIZE and WRITE_SIZE in separate passes (tools/summarize_prof.py condenses them)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
M="python3 $R/tools/bench_mit.py --batch 16"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_mit_stats -- $M --steps 2 > $R/gpurun_out/pm1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/p_mit_fetch -- $M --steps 1 > $R/gpurun_out/pm2.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/p_mit_write -- $M --steps 1 > $R/gpurun_out/pm3.log 2>&1
cd $R
python3 tools/summarize_prof.py r04_mit_b5 gpurun_out/p_mit_stats gpurun_out/p_mit_fetch gpurun_out/p_mit_write
mkdir -p gpurun_out/profiles_r04
cp profiles/r04_mit_b5_kernel_stats.csv profiles/r04_mit_b5_pmc_summary.json gpurun_out/profiles_r04/
rm -rf gpurun_out/p_mit_stats gpurun_out/p_mit_fetch gpurun_out/p_mit_write
tail -2 gpurun_out/pm1.log | cut -c1-300
python3 - <<'PY'
import json
p=json.load(open('profiles/r04_mit_b5_pmc_summary.json'))['kernels']
for k,v in p.items():
    if 'dwconv' in k: print(k[:70], v['launches'], round(v['hbm_bytes_per_launch_corrected']/1e6,1),'MB/launch')
PY
