#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=10 > gpurun_out/r03_gputests_full.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03_gputests_full.log
tail -18 gpurun_out/r03_gputests_full.log
timeout -k 10 300 python bench.py --workload C2step3 --steps 3 --warmup 1 > gpurun_out/r03_bench_c2step3_v1.json 2> gpurun_out/r03_bench_c2step3_v1.err && python -c 'import json; d=json.load(open("gpurun_out/r03_bench_c2step3_v1.json")); print("C2step3", d["value"], d["roofline"]["frac"], d["population_after"])' &&
timeout -k 10 300 python bench.py --workload C2step --steps 3 --warmup 1 > gpurun_out/r03_bench_c2step_v1.json 2>/dev/null && python -c 'import json; d=json.load(open("gpurun_out/r03_bench_c2step_v1.json")); print("C2step", d["value"], d["roofline"]["frac"], d["population_after"])' &&
timeout -k 10 600 python bench.py --workload C5 --via capi --comm-at-one --no-cpu-baseline --steps 1 --warmup 1 > gpurun_out/r03_bench_c5_v1.json 2> gpurun_out/r03_bench_c5_v1.err && python -c 'import json; d=json.load(open("gpurun_out/r03_bench_c5_v1.json")); print("C5", d["value"], d["roofline"]["frac"], d["phases_ms"], d["mfma_frac_step"])'
