mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_hip_t2s_lowp.py -q -s 2>&1 | grep -E "^tests|Error|assert |layers:|tokens equal|identical|max / mean|fp8 bs|passed|failed" ) > gpurun_out/lowp3.log
cd /tmp && export TMPDIR=/tmp
for B in 64 12; do
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof$B -o p$B -- python $GRAFT_REPO_ROOT/tools/step_time.py $B bf16 > /tmp/prof$B.log 2>&1
db=$(find /tmp/prof$B -name "*.db" | head -1)
echo "== B=$B $db" ; tail -2 /tmp/prof$B.log; python $GRAFT_REPO_ROOT/tools/prof_kernel_table.py $db 0.3
done > $GRAFT_REPO_ROOT/gpurun_out/prof_batched.txt 2>&1
cd $GRAFT_REPO_ROOT
cat gpurun_out/lowp3.log gpurun_out/prof_batched.txt
