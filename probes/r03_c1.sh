#!/bin/bash
P='import json,sys; d=json.load(sys.stdin); print(sys.argv[1], d["value"], d["roofline"]["achieved"], d["roofline"]["kernel_ms"], d["phases_ms"], d.get("pruned",{}).get("ms_per_step"))'
for wl in C1 C2; do
python bench.py --workload $wl --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | python -c "$P" ${wl}_default
GPLE_PREDICT_SMALL_M=0 python bench.py --workload $wl --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | python -c "$P" ${wl}_rownorm
GPLE_PREDICT_SMALL_M=1 python bench.py --workload $wl --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | python -c "$P" ${wl}_gemm
done
GPLE_ROWNORM_VARIANT=2 python bench.py --workload C2 --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | python -c "$P" C2_variant44
