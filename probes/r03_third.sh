#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
B="python bench.py --via capi --comm-at-one --no-cpu-baseline --steps 10"
P='import json,sys; d=json.load(sys.stdin); print(sys.argv[1], d["value"], d["phases_ms"])'
$B 2>/dev/null | python -c "$P" capi_base &&
GPLE_CHOL_OVERLAP_MIN_N=100000 $B 2>/dev/null | python -c "$P" capi_no_side_stream &&
BENCH_OWN_STREAM=1 $B 2>/dev/null | python -c "$P" capi_own_stream &&
BENCH_COMM_UNUSED=1 $B 2>/dev/null | python -c "$P" comm_unused &&
GPLE_CHOL_SIDE_PRIORITY=0 $B 2>/dev/null | python -c "$P" capi_side_prio0 &&
GPU_MAX_HW_QUEUES=8 $B 2>/dev/null | python -c "$P" capi_hwq8 &&
timeout -k 10 600 python -m pytest tests/test_gpu_step_loop.py tests/test_gpu_adapters.py -x -q > gpurun_out/r03_gputests3.log 2>&1; tail -5 gpurun_out/r03_gputests3.log
