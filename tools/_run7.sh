mkdir -p gpurun_out
(for cfg in "64 512 200 300" "256 512 200 300" "32 512 200 300" "64 1024 300 600" "64 512 400 500"; do timeout 120 tools/battn_bench $cfg; done) > gpurun_out/battn7.log 2>&1
cat gpurun_out/battn7.log
