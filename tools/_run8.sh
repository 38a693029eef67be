mkdir -p gpurun_out
for B in 16 24 32 64 256; do timeout 120 python tools/step_time.py $B bf16 2>&1 | grep "step"; done > gpurun_out/step8.log
for B in 32 64 256; do timeout 120 python tools/step_time.py $B fp8 2>&1 | grep "step"; done >> gpurun_out/step8.log
(timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -15) > gpurun_out/gpu8.log
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench8.json 2> gpurun_out/bench8.err
cat gpurun_out/step8.log gpurun_out/gpu8.log; tail -2 gpurun_out/bench8.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench8.json').read().strip().splitlines()[-1])
for k in ('value','ar_ms_per_token','vocoder_ms','ttft_ms_p50','ttfa_ms_p50'): print(k, d.get(k))
print(d.get('roofline')); print(d.get('roofline_step'))
PY
