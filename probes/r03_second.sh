#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=8 --deselect tests/test_gpu_adapters.py > gpurun_out/r03_gputests2.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03_gputests2.log
tail -16 gpurun_out/r03_gputests2.log
timeout -k 10 300 python bench.py --via capi --comm-at-one --no-cpu-baseline --steps 10 > gpurun_out/r03_bench_c4r_capi1.json 2> gpurun_out/r03_bench_c4r_capi1.err && echo "bench capi ok" &&
timeout -k 10 300 python bench.py --workload C4 --via capi --comm-at-one --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r03_bench_c4_v1.json 2> gpurun_out/r03_bench_c4_v1.err && echo "bench C4 ok" &&
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 > gpurun_out/r03_bench_c4r_v2.json 2> gpurun_out/r03_bench_c4r_v2.err && echo "bench default ok"
