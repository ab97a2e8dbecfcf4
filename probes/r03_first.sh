#!/bin/bash
# round 3, first GPU call: whole GPU suite, the default bench line, the product's RCCL path with a one-rank communicator, the C4 workload, fit timings
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r03_gputests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03_gputests.log
tail -25 gpurun_out/r03_gputests.log
timeout -k 10 300 python bench.py > gpurun_out/r03_bench_c4r_v1.json 2> gpurun_out/r03_bench_c4r_v1.err && echo "bench default ok" &&
timeout -k 10 300 python bench.py --via capi --comm-at-one --no-cpu-baseline --steps 10 > gpurun_out/r03_bench_c4r_capi1.json 2> gpurun_out/r03_bench_c4r_capi1.err && echo "bench capi ok" &&
timeout -k 10 300 python bench.py --workload C4 --via capi --comm-at-one --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r03_bench_c4_v1.json 2> gpurun_out/r03_bench_c4_v1.err && echo "bench C4 ok" &&
timeout -k 10 300 python probes/fit_timing.py both 256 1024 2048 4096 8192 > gpurun_out/r03_fit_timing_v1.log 2>&1 && cat gpurun_out/r03_fit_timing_v1.log
