#!/bin/bash
# round 4, second evidence pass (after the contraction's final schedule): profiler passes and plain bench lines of the configs whose dominant kernel changed
set -o pipefail
R=$(pwd)
mkdir -p $R/gpurun_out/r04_prof $R/gpurun_out/r04_bench
for WL in C1 C2 C3 C4r C4c C5r; do bash probes/r04_profile.sh $WL || exit 1; done
cd $R
for WL in C1 C2 C3 C4r C4c C5r; do
  timeout -k 10 300 python bench.py --workload $WL --steps 20 --warmup 3 > gpurun_out/r04_bench/$WL.json 2> gpurun_out/r04_bench/$WL.err || exit 1
done
for e in 0/2 0/4 0/8 3/8; do
  timeout -k 10 300 python bench.py --via capi --emulate-rank $e --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/r04_bench/emu_${e/\//_}.json 2>/dev/null || exit 1
done
timeout -k 10 300 python bench.py --workload C4 --via capi --comm-at-one --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/r04_bench/C4.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --workload C2step --steps 20 --warmup 3 > gpurun_out/r04_bench/C2step.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --workload C5step --steps 2 --warmup 1 > gpurun_out/r04_bench/C5step.json 2>/dev/null || exit 1
echo evidence2-done
