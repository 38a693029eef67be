mkdir -p gpurun_out
for B in 16 32 64 256; do timeout 120 python tools/step_time.py $B bf16 2>&1 | grep "step"; done > gpurun_out/step10.log
for B in 8 12 16 24; do GSV_BATCHED_MIN=10000 timeout 120 python tools/step_time.py $B bf16 2>&1 | grep "step"; done >> gpurun_out/step10.log
for B in 64; do timeout 120 python tools/step_time.py $B fp8 2>&1 | grep "step"; done >> gpurun_out/step10.log
cat gpurun_out/step10.log
