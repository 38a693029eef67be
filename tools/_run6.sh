mkdir -p gpurun_out
for skip in 0 2; do for B in 64 32 256; do GSV_BSTEP_SKIP=$skip timeout 120 python tools/step_time.py $B bf16 2>&1 | grep "step" | sed "s/^/skip=$skip /"; done; done > gpurun_out/skip6.log
for B in 1 4 8 16; do GSV_BATCHED_MIN=10000 timeout 120 python tools/step_time.py $B bf16 2>&1 | grep step; done >> gpurun_out/skip6.log
(timeout 1500 python -m pytest tests/test_hip_t2s_lowp.py tests/test_hip_t2s.py -q -s 2>&1 | grep -E "^tests|Error|assert |layers:|tokens equal|identical|max, mean|fp8 bs|passed|failed" ) > gpurun_out/lowp6.log
cat gpurun_out/skip6.log gpurun_out/lowp6.log
