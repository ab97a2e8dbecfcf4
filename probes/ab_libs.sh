#!/bin/bash
# same-box A/B of library builds: probes/ab/lib*.so are copied over csrc/libgple_hip.so in turn and the given workloads benched
#   probes/ab_libs.sh "A B" "C4r C2" [rounds]
set -e
C=gaussian_process_liouville_equation_amd/csrc
mkdir -p gpurun_out/ab
for r in $(seq 1 ${3:-2}); do
for v in $1; do
  cp probes/ab/lib$v.so $C/libgple_hip.so
  for w in $2; do
    python bench.py --workload $w --no-cpu-baseline > gpurun_out/ab/${w}_$v.json 2> gpurun_out/ab/${w}_$v.err
    python - <<PY
import json
d = json.load(open("gpurun_out/ab/${w}_$v.json"))
print("round $r lib $v ${w}", d["value"], "ms/step  frac", d["roofline"]["frac"], "kernel_ms", d["roofline"].get("kernel_ms"), flush=True)
PY
  done
done
done
