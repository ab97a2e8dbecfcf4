#!/bin/bash
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_optimization.py tests/test_gpu_campaign.py -x -q -m gpu 2>&1 | tail -4
python bench.py --workload C4opt --steps 5 --warmup 2 > gpurun_out/r03_bench_c4opt_v3.json 2>/dev/null && python -c 'import json; d=json.load(open("gpurun_out/r03_bench_c4opt_v3.json")); print("C4opt", d["value"], d["roofline"]["achieved"], d["mfma_frac_step"], d["phases_ms"])'
python probes/objective_eval_timing.py 1024 4096 2>&1 | grep N= && GPLE_PREDICT_SKIP=0 python probes/objective_eval_timing.py 1024 4096 2>&1 | grep N= | sed "s/^/skip off: /"
