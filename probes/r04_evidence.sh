#!/bin/bash
# round 4 evidence: rocprofv3 kernel stats + PMC passes of every benchable workload, the fit's kernel traces with the GEMM launch log,
# and the plain bench lines (no profiler) of every workload
set -o pipefail
R=$(pwd)
OUT=$R/gpurun_out/r04_prof
mkdir -p $OUT $R/gpurun_out/r04_bench
for WL in C1 C2 C3 C4r C4c C5r; do bash probes/r04_profile.sh $WL || exit 1; done
TAG=C4opt_only1 bash probes/r04_profile.sh C4opt --opt-only 1 || exit 1
TAG=C4opt_only0 bash probes/r04_profile.sh C4opt --opt-only 0 || exit 1
bash probes/r04_profile.sh C4opt || exit 1
cd /tmp && export TMPDIR=/tmp
for N in 1024 4096; do
  GPLE_GEMM_LOG=$OUT/gemm_log_$N.txt rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fit_$N -- python3 $R/probes/fit_timing.py real $N > $OUT/fit_$N.log 2>&1 || exit 1
done
cd $R
for WL in C1 C2 C3 C4r C4c C5r; do
  timeout -k 10 300 python bench.py --workload $WL --steps 20 --warmup 3 > gpurun_out/r04_bench/$WL.json 2> gpurun_out/r04_bench/$WL.err || exit 1
done
timeout -k 10 300 python bench.py --workload C4opt --steps 10 --warmup 2 > gpurun_out/r04_bench/C4opt.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --workload C4opt --opt-only 1 --steps 10 --warmup 2 > gpurun_out/r04_bench/C4opt_only1.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --workload C4 --via capi --comm-at-one --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/r04_bench/C4.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --workload C2step --steps 20 --warmup 3 > gpurun_out/r04_bench/C2step.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --workload C5step --steps 2 --warmup 1 > gpurun_out/r04_bench/C5step.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --workload C2step3 --steps 10 --warmup 3 > gpurun_out/r04_bench/C2step3.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --via capi --comm-at-one --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/r04_bench/C4r_capi_one_rank.json 2>/dev/null || exit 1
for e in 0/2 0/4 0/8 3/8; do
  timeout -k 10 300 python bench.py --via capi --emulate-rank $e --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/r04_bench/emu_${e/\//_}.json 2>/dev/null || exit 1
done
for WL in NLML1024 NLML4096; do
  timeout -k 10 600 python bench.py --workload $WL --steps 10 --warmup 2 > gpurun_out/r04_bench/$WL.json 2>/dev/null || exit 1
done
python probes/fit_timing.py both 256 1024 2048 4096 8192 > gpurun_out/r04_bench/fit_timing.log 2>&1
python probes/n1_latency.py > gpurun_out/r04_bench/n1_latency.log 2>&1
echo evidence-done
