#!/bin/bash
# the multi-rank code paths of bench.py that do not need two devices, on the one GPU of the box: (1) bench.py starting its own ranks (no launcher
# around it) over gloo, (2) the same under torch.distributed.run, (3) the library-side RCCL path with a one-rank communicator
set -o pipefail
mkdir -p gpurun_out/r04_bench
P='import json,sys; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], d["value"], d["n_gpus"], d["config"]["parallelism"][:90], d["config"].get("via"), d["config"].get("collective","")[:80])'
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --steps 3 --warmup 1 --no-cpu-baseline 2>gpurun_out/r04_bench/rehearse_self_launch.err | tee gpurun_out/r04_bench/rehearse_c4r_2rank_gloo_self_launched.json | python -c "$P" C4r_gloo2_self_launched &&
T="python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1" &&
timeout -k 10 300 $T --master-port 29512 bench.py --gpus 2 --backend gloo --workload C4opt --steps 2 --warmup 1 2>/dev/null | tee gpurun_out/r04_bench/rehearse_c4opt_2rank_gloo.json | python -c "$P" C4opt_gloo2 &&
timeout -k 10 300 $T --master-port 29513 bench.py --gpus 2 --backend gloo --workload C4 --via torch --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tee gpurun_out/r04_bench/rehearse_c4_2rank_gloo.json | python -c "$P" C4_gloo2 &&
timeout -k 10 300 python bench.py --via capi --comm-at-one --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "$P" C4r_rccl_one_rank
