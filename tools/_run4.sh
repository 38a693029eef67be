mkdir -p gpurun_out
for skip in 0 1 2 4 8 16 31; do for B in 64 32; do GSV_BSTEP_SKIP=$skip timeout 120 python tools/step_time.py $B bf16 2>&1 | grep "step" | sed "s/^/skip=$skip /"; done; done > gpurun_out/skip4.log
(timeout 1500 python -m pytest tests/test_hip_t2s_lowp.py tests/test_hip_engine.py -q -s 2>&1 | grep -E "^tests|Error|assert |layers:|tokens equal|identical|max / mean|fp8 bs|passed|failed" ) > gpurun_out/lowp4.log
cat gpurun_out/skip4.log gpurun_out/lowp4.log
