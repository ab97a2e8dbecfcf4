#!/bin/bash
# after pick_side_stream: the collision cases of r03_hwq.log again, then the capi bench, then the affected tests
for args in "0 1" "3 0" "4 1" "6 1"; do
  python probes/hwqueue_fit_probe.py $args 2>&1 | grep pre_streams || exit 1
  GPLE_CHOL_SIDE_PRIORITY=0 python probes/hwqueue_fit_probe.py $args 2>&1 | grep pre_streams || exit 1
done
python bench.py --via capi --comm-at-one --no-cpu-baseline --steps 10 2>/dev/null | python -c 'import json,sys; d=json.load(sys.stdin); print("capi", d["value"], d["phases_ms"])' &&
python bench.py --workload C4 --via capi --comm-at-one --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r03_bench_c4_v2.json 2>/dev/null && python -c 'import json; d=json.load(open("gpurun_out/r03_bench_c4_v2.json")); print("C4", d["value"], d["phases_ms"], d["roofline"]["frac"])' &&
timeout -k 10 600 python -m pytest tests/test_gpu_chol_diag.py tests/test_gpu_step_loop.py tests/test_gpu_sharded.py -x -q 2>&1 | tail -5
