mkdir -p gpurun_out
(timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -6) > gpurun_out/gpu13.log
timeout 600 python bench.py --workload cb --version v2ProPlus --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/cb13.json 2> gpurun_out/cb13.err
timeout 600 python bench.py --workload cb --version v2ProPlus --dtype fp8 --slots 64 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/cb13_fp8.json 2> gpurun_out/cb13_fp8.err
timeout 600 python bench.py --workload cb --version v2ProPlus --slots 64 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/cb13_64.json 2> gpurun_out/cb13_64.err
cat gpurun_out/gpu13.log; python - <<'PY'
import json
for f in ('cb13','cb13_fp8','cb13_64'):
    try:
        d=json.loads(open('gpurun_out/%s.json'%f).read().strip().splitlines()[-1])
        print(f, 'value %.0f AR %.0f voc %.0f slotloop %.3f ms roof %.3f' % (d['value'], d['rank0_ar_tokens_per_s'], d['rank0_vocoder_audio_s_per_s'], d['roofline']['ms_per_step_of_the_slot_loop'], d['roofline']['frac']))
    except Exception as e: print(f, 'ERR', e, open('gpurun_out/%s.err'%f).read()[-500:])
PY
