#!/bin/bash
# A/B of the contraction kernels on one box: rownorm2_kernel (GPLE_ROWNORM_PIPE=0) against rownormp_kernel (default), plain bench runs
#   probes/r04_rownorm_ab.sh [workloads] [pipe values]
set -e
mkdir -p gpurun_out/r04_ab
for w in ${1:-C4r C2 C1 C3 C5r}; do
  for pipe in ${2:-0 1}; do
    GPLE_ROWNORM_PIPE=$pipe python bench.py --workload $w --no-cpu-baseline > gpurun_out/r04_ab/${w}_pipe${pipe}.json 2> gpurun_out/r04_ab/${w}_pipe${pipe}.err
    python - <<PY
import json
d = json.load(open("gpurun_out/r04_ab/${w}_pipe${pipe}.json"))
print("${w} pipe=${pipe}", d["value"], "ms/step  frac", d["roofline"]["frac"], "kernel_ms", d["roofline"].get("kernel_ms"), flush=True)
PY
  done
done
