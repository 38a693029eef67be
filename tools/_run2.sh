mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_hip_t2s_lowp.py -q -s 2>&1 | grep -v "^$" | tail -60) > gpurun_out/lowp2.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof64 -o p64 -- python $GRAFT_REPO_ROOT/tools/step_time.py 64 bf16 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof12 -o p12 -- python $GRAFT_REPO_ROOT/tools/step_time.py 12 bf16 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
for d in 64 12; do f=$(find /tmp/prof$d -name "*kernel_stats.csv" | head -1); echo "== $d $f"; head -25 "$f" | cut -d, -f1-8 | cut -c1-200; done > gpurun_out/prof_batched.txt
cat gpurun_out/lowp2.log gpurun_out/prof_batched.txt
