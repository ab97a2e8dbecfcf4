#!/bin/bash
# rocprofv3 passes of the default bench (C4r) — kernel stats, then the PMC passes one counter group at a time
# (gpurun refuses --pmc combined with trace domains other than --kernel-trace).  Run from the repo root on the GPU box.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02_prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
WL=${1:-C4r}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$WL -- python3 $R/bench.py --workload $WL --steps 10 --warmup 2 --no-cpu-baseline > $OUT/stats_$WL.log 2>&1 || exit 1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-include-regex "rownorm|kstar_gen" --output-format csv -d $OUT/pmc_${C}_$WL -- python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_${C}_$WL.log 2>&1 || exit 1
done
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-include-regex "rownorm" --output-format csv -d $OUT/pmc_SQ_$WL -- python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_SQ_$WL.log 2>&1 || exit 1
find $OUT -name "*.csv" | head -50
