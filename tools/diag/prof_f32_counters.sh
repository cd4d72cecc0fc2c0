cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python3 $R/bench.py --lean --no-prof --steps 1 --warmup 1 --serial-streams --precision f32"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/p_f32_mfma -- $B > $R/gpurun_out/p4.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/p_f32_lds -- $B > $R/gpurun_out/p5.log 2>&1
cd $R
find gpurun_out/p_* -name "*.csv" -size +30M -delete
du -sh gpurun_out/p_*
