#!/bin/bash
# Round-end measurement sweep on one MI355X (run on the GPU box through tools/gpu.sh): writes everything under gpurun_out/sweep/.
# Copy what is to be kept into profiles/ afterwards.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/sweep; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { echo "== $*" >&2; "$@"; }
# 1. default bench line (configs[1]) and its rocprofv3 summary
run timeout 900 python $R/bench.py > $O/bench_line.json 2> $O/bench_line.log
rm -rf /tmp/p1; run timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-cb32 > /dev/null 2> $O/prof_bench.log
f=$(find /tmp/p1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f" > $O/rocprofv3_kernel_stats.csv
# 2. HBM / fabric traffic: two PMC passes (counters only, no tracing beyond kernel dispatch)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm_$c; run timeout 900 rocprofv3 --pmc $c --output-format csv -d /tmp/pm_$c -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-cb32 > /dev/null 2> $O/pmc_$c.log
done
# ... and the batched decode step of the cb32 record (32 sequences, kv ~ 350-450)
for c in FETCH_SIZE WRITE_SIZE; do
  # eager launches, 20 steps: the FETCH_SIZE pass of the graph-replayed chain hangs under the profiler (r04, r05)
  rm -rf /tmp/ps_$c; GSV_NO_GRAPH=1 GSV_STEPS=20 GSV_PROMPT_TOK=250 run timeout 200 rocprofv3 --pmc $c --output-format csv -d /tmp/ps_$c -- python $R/tools/step_time.py 32 bf16 > /dev/null 2> $O/pmc_step_$c.log
done
ff=$(find /tmp/pm_FETCH_SIZE -name "*counter_collection.csv" | head -1); fw=$(find /tmp/pm_WRITE_SIZE -name "*counter_collection.csv" | head -1)
sf=$(find /tmp/ps_FETCH_SIZE -name "*counter_collection.csv" | head -1); sw=$(find /tmp/ps_WRITE_SIZE -name "*counter_collection.csv" | head -1)
[ -n "$ff" ] && [ -n "$fw" ] && python $R/tools/pmc_traffic.py "$ff" "$fw" $O/traffic.json $sf $sw > $O/traffic_table.txt 2>&1
# 3. continuous batching lines
run timeout 900 python $R/bench.py --workload cb --version v2ProPlus > $O/cb_configs2.json 2> $O/cb_configs2.log
run timeout 900 python $R/bench.py --workload cb --version v2ProPlus --sync-refill --no-cpu-baseline > $O/cb_configs2_sync_refill.json 2> /dev/null
run timeout 900 python $R/bench.py --workload cb --version v2ProPlus --lpt-budget --no-cpu-baseline > $O/cb_configs2_lpt_budget.json 2> /dev/null
run timeout 900 python $R/bench.py --workload cb --version v2Pro --no-cpu-baseline > $O/cb_v2pro_bs32.json 2> /dev/null
GSV_REFILL_AHEAD=0 run timeout 900 python $R/bench.py --workload cb --version v2Pro --no-cpu-baseline > $O/cb_v2pro_bs32_staged_loop.json 2> /dev/null
GSV_TAIL_LEVELS=0 run timeout 900 python $R/bench.py --workload cb --version v2Pro --no-cpu-baseline > $O/cb_v2pro_bs32_no_tail_compaction.json 2> /dev/null
run timeout 900 python $R/bench.py --workload cb --version v2ProPlus --slots 64 --no-cpu-baseline > $O/cb_bf16_bs64.json 2> /dev/null
run timeout 900 python $R/bench.py --workload cb --version v2ProPlus --slots 64 --dtype fp8 --no-cpu-baseline > $O/cb_fp8_bs64.json 2> /dev/null
# N > 1 through the bench's own launcher (no torchrun around it): two ranks on this box's one GPU, gloo
run timeout 900 python $R/bench.py --gpus 2 --workload cb --share-gpu --dist-backend gloo --steps 1 --warmup 1 --requests 64 --no-cpu-baseline 2> $O/cb_2ranks.log | grep '^{' > $O/cb_2ranks_bare_launch.json
rm -rf /tmp/p2; run timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -- python $R/bench.py --workload cb --version v2ProPlus --steps 1 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
f=$(find /tmp/p2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f" > $O/rocprofv3_kernel_stats_cb_configs2.csv
# 4. raw step times
( for b in 1 2 4 8 16 17 24 32 33 40 64 128 256; do timeout 300 python $R/tools/step_time.py $b bf16 | grep step; done
  for b in 64 256; do timeout 300 python $R/tools/step_time.py $b fp8 | grep step; done
  for b in 1 2 4; do timeout 300 python $R/tools/step_time.py $b fp32 | grep step; done
  GSV_PROMPT_TOK=250 timeout 300 python $R/tools/step_time.py 32 bf16 | grep step | sed 's/$/   <- kv 350-450 (GSV_PROMPT_TOK=250): the kv the cb32 slot loop runs at/' ) > $O/step_time.txt 2>&1
# 5. vocoder: pass times and per-kernel timelines
timeout 600 python $R/tools/voc_time.py 2>&1 | grep "T=" > $O/voc_time.txt
GSV_CGEMM_NO_SPLIT=1 timeout 600 python $R/tools/voc_time.py v2ProPlus 2>&1 | grep "T=" | sed 's/$/   <- GSV_CGEMM_NO_SPLIT=1/' >> $O/voc_time.txt
timeout 300 python $R/tools/voc_gap.py 2>&1 | grep "T=" > $O/voc_gap.txt
GSV_VOC_DTYPE=fp32 timeout 600 python $R/tools/voc_time.py v2Pro 2>&1 | grep "T=" | sed 's/$/   <- fp32 handle/' >> $O/voc_time.txt
( timeout 120 python $R/tools/prefill_time.py 1; GSV_DTYPE=fp32 timeout 120 python $R/tools/prefill_time.py 1 | sed 's/$/   <- fp32 handle/'; timeout 120 python $R/tools/prefill_time.py 32 ) 2>&1 | grep rows > $O/prefill_time.txt
for v in v2Pro v2ProPlus; do
  rm -rf /tmp/pv_$v; timeout 600 rocprofv3 --kernel-trace -d /tmp/pv_$v -- python $R/tools/voc_time.py $v > /dev/null 2>&1
  db=$(find /tmp/pv_$v -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/prof_timeline.py "$db" vocpass > $O/vocoder_timeline_$v.txt 2>&1
done
timeout 300 python $R/tools/sample_speed.py 2>&1 | grep token > $O/sample_speed.txt
# 6. placement spread (ten model instances per setting), the fused-resblock and conv-GEMM micro-benchmarks
( echo "arena (default), instances kept alive"; KEEP=1 timeout 300 python $R/tools/placement_ab.py 10 | grep step
  echo "separate allocations (GSV_NO_ARENA=1)"; GSV_NO_ARENA=1 KEEP=1 timeout 300 python $R/tools/placement_ab.py 10 | grep step ) > $O/placement_ab.txt 2>&1
[ -x $R/tools/rb_bench ] && ( $R/tools/rb_bench 16 5000; $R/tools/rb_bench 32 3000 24; $R/tools/rb_bench 16 320000; $R/tools/rb_bench 32 160000; $R/tools/rb_bench 16 3200000; $R/tools/rb_bench 32 1600000 ) 2>&1 | grep -v stamps > $O/rbfuse_bench.txt
[ -x $R/tools/cg_bench ] && ( for sp in 1,1,1 2,2,1 3,2,1 3,3,1; do CG_SPLIT=$sp timeout 30 $R/tools/cg_bench 384 5003 5 1 128; done; for sp in 1,1,1 3,2,1; do CG_SPLIT=$sp timeout 30 $R/tools/cg_bench 384 500 5 1 128; done ) 2>&1 | grep -v "block 0" > $O/cgemm_ksplit.txt
[ -x $R/tools/cg_bench ] && ( for c in 384 256 192 128; do $R/tools/cg_bench $c 5003 5 1 128; done; $R/tools/cg_bench 384 50000 5 1 128; $R/tools/cg_bench 256 50000 5 1 128; $R/tools/cg_bench 192 400000 5 1 128; $R/tools/cg_bench 128 400000 5 1 128 ) 2>&1 | grep -v "block 0" > $O/cgemm_bench.txt
ls -la $O >&2
