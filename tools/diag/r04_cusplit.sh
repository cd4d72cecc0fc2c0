cd $GRAFT_REPO_ROOT
for ab in "all all" "lo hi" "even odd" "lo3q hi1q" "x3q x1q" "all hi1q" "all x1q"; do
  set -- $ab
  timeout 300 python tools/diag/cu_split_probe.py --a $1 --b $2 2>&1 | grep -E "CU split|Error|error|assert" | tail -2 >> gpurun_out/r04_cusplit.log
done
cat gpurun_out/r04_cusplit.log
