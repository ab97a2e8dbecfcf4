#!/bin/bash
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -k "short_factor or c1_exact or pruning" 2>&1 | tail -4 || exit 1
P='import json,sys; d=json.load(sys.stdin); print(sys.argv[1], d["value"], d["roofline"]["achieved"], d["roofline"]["kernel_ms"], d["phases_ms"])'
python bench.py --workload C1 --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | python -c "$P" C1_short
GPLE_ROWNORM_SHORT=0 python bench.py --workload C1 --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | python -c "$P" C1_general
