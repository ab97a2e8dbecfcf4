#!/bin/bash
# kernel trace of a few fits at size $1 (default 1024): timeline of one fit, condensed by probes/fit_trace_summary.py
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/fit_trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t_$1 -- python3 $R/probes/fit_timing.py real ${1:-1024} > $OUT/t_$1.log 2>&1 || exit 1
python3 $R/probes/fit_trace_summary.py $(find $OUT/t_$1 -name "*kernel_trace.csv" | head -1) > $OUT/summary_$1.txt
tail -3 $OUT/t_$1.log
