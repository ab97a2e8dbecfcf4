#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_campaign.py tests/test_gpu_configs.py -x -q -k "complex or loose or c4 or C4" > gpurun_out/r03_gputests4.log 2>&1; tail -4 gpurun_out/r03_gputests4.log
python bench.py --workload C4opt --opt-only 1 --steps 5 --warmup 2 > gpurun_out/r03_bench_c4opt_only1_v2.json 2>/dev/null && python -c 'import json; d=json.load(open("gpurun_out/r03_bench_c4opt_only1_v2.json")); print("only1", d["value"], d["roofline"]["achieved"], d["phases_ms"])' &&
python bench.py --workload C4opt --steps 5 --warmup 2 > gpurun_out/r03_bench_c4opt_v2.json 2>/dev/null && python -c 'import json; d=json.load(open("gpurun_out/r03_bench_c4opt_v2.json")); print("C4opt", d["value"], d["roofline"]["achieved"], d["mfma_frac_step"], d["phases_ms"])' &&
timeout -k 10 900 python -m pytest tests/test_optimization.py tests/test_gpu_adapters.py -x -q -m gpu --durations=5 2>&1 | tail -12
