set -x
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_hip_t2s_lowp.py -x -q -s 2>&1 | tail -40) > gpurun_out/lowp1.log
for B in 4 8 12 16 24 32 64 128 256; do
  timeout 120 python tools/step_time.py $B bf16
  GSV_BATCHED_MIN=10000 timeout 120 python tools/step_time.py $B bf16
done > gpurun_out/step1.log 2>&1
for B in 32 64 256; do timeout 120 python tools/step_time.py $B fp8; done >> gpurun_out/step1.log 2>&1
(timeout 1200 python -m pytest tests/test_hip_t2s.py -x -q 2>&1 | tail -15) > gpurun_out/t2s1.log
cat gpurun_out/lowp1.log gpurun_out/step1.log gpurun_out/t2s1.log
