mkdir -p gpurun_out
for B in 8 12 16 24 32 40 48; do GSV_BATCHED_MIN=10000 timeout 120 python tools/step_time.py $B bf16 2>&1 | grep "step"; done > gpurun_out/step11.log
for B in 16 32; do GSV_BATCHED_MIN=10000 timeout 120 python tools/step_time.py $B fp32 2>&1 | grep "step"; done >> gpurun_out/step11.log
(timeout 1500 python -m pytest tests/test_hip_t2s.py tests/test_hip_t2s_lowp.py tests/test_hip_engine.py -q -x 2>&1 | tail -8) >> gpurun_out/step11.log
cat gpurun_out/step11.log
