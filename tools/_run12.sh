mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for B in 32 16; do
GSV_BATCHED_MIN=10000 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof$B -o p$B -- python $GRAFT_REPO_ROOT/tools/step_time.py $B bf16 > /tmp/prof$B.log 2>&1
db=$(find /tmp/prof$B -name "*.db" | head -1)
echo "== B=$B"; grep step /tmp/prof$B.log; python $GRAFT_REPO_ROOT/tools/prof_kernel_table.py $db 0.3
done > $GRAFT_REPO_ROOT/gpurun_out/prof12.txt 2>&1
cd $GRAFT_REPO_ROOT; cat gpurun_out/prof12.txt
