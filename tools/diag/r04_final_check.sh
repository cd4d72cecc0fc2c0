cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r04_final_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_final_smoke.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_final.log 2> gpurun_out/r04_bench_final.err
cat gpurun_out/r04_final_tests.log; tail -4 gpurun_out/r04_final_smoke.log; tail -1 gpurun_out/r04_bench_final.log | cut -c1-400
