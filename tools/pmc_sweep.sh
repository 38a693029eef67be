#!/bin/bash
# MFMA utilisation / LDS bank-conflict share per kernel at HEAD: for each workload one un-instrumented kernel trace (durations) and
# one counter pass (rocprofv3 --pmc only, as gpurun requires) -> gpurun_out/pmc/<name>_mfma_lds.txt  (tools/pmc_table.py).
# Workloads: the vocoder pass (v2Pro, v2ProPlus; eager launches so that every kernel is its own dispatch), the batched decode chain
# at 32 and 64 sequences (bf16) and 64 (e4m3), the prompt pass.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
one() {   # name, command...
  local name=$1; shift
  rm -rf /tmp/pt_$name /tmp/pc_$name
  timeout 600 rocprofv3 --kernel-trace -d /tmp/pt_$name -- "$@" > /dev/null 2> $O/${name}_trace.log
  local db=$(find /tmp/pt_$name -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/prof_summary.py "$db" 60 > $O/${name}_kernel_trace.txt
  timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pc_$name -- "$@" > /dev/null 2> $O/${name}_pmc.log
  local cc=$(find /tmp/pc_$name -name "*counter_collection.csv" | head -1)
  [ -n "$cc" ] && [ -n "$db" ] && python $R/tools/pmc_table.py "$cc" $O/${name}_kernel_trace.txt > $O/${name}_mfma_lds.txt
  echo "== $name" >&2; head -30 $O/${name}_mfma_lds.txt >&2
}
one voc_v2Pro python $R/tools/voc_time.py v2Pro
one voc_v2ProPlus python $R/tools/voc_time.py v2ProPlus
one chain_b32 python $R/tools/step_time.py 32 bf16
one chain_b64 python $R/tools/step_time.py 64 bf16
one chain_b64_fp8 python $R/tools/step_time.py 64 fp8
one prompt_pass python $R/tools/prefill_time.py 32
