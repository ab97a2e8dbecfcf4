#!/bin/bash
# rocprofv3 passes of one bench workload — kernel stats, then the PMC passes one counter group at a time (gpurun refuses --pmc combined
# with trace domains other than --kernel-trace).  Run from the repo root on the GPU box: probes/r04_profile.sh <workload> [extra bench args]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
WL=${1:-C4r}; shift
TAG=${TAG:-$WL}
RX="rownorm|kstar_gen|predict_fused"; [ "$WL" = "C4opt" ] && RX="gemm_f64"
B="python3 $R/bench.py --workload $WL --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$TAG -- $B --steps 5 --warmup 2 > $OUT/stats_$TAG.log 2>&1 || exit 1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-include-regex "$RX" --output-format csv -d $OUT/pmc_${C}_$TAG -- $B --steps 2 --warmup 1 > $OUT/pmc_${C}_$TAG.log 2>&1 || exit 1
done
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-include-regex "$RX" --output-format csv -d $OUT/pmc_SQ_$TAG -- $B --steps 2 --warmup 1 > $OUT/pmc_SQ_$TAG.log 2>&1 || exit 1
tail -1 $OUT/stats_$TAG.log | cut -c1-300
