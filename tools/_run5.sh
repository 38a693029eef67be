mkdir -p gpurun_out
for skip in 0 2; do for B in 64 32; do GSV_BSTEP_SKIP=$skip timeout 120 python tools/step_time.py $B bf16 2>&1 | grep "step" | sed "s/^/skip=$skip /"; done; done > gpurun_out/skip5.log
(timeout 1500 python -m pytest tests/test_hip_t2s_lowp.py tests/test_hip_t2s.py -q -s 2>&1 | grep -E "^tests|Error|assert |layers:|tokens equal|identical|max, mean|fp8 bs|passed|failed" ) > gpurun_out/lowp5.log
cat gpurun_out/skip5.log gpurun_out/lowp5.log
