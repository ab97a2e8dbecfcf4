#!/bin/bash
# two ranks on the one GPU of the box over gloo: the multi-rank code paths of bench.py that do not need two devices
set -o pipefail
mkdir -p gpurun_out/r03_bench
T="python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1"
P='import json,sys; d=json.load(sys.stdin); print(sys.argv[1], d["value"], d["n_gpus"], d["config"]["parallelism"][:90], d["config"].get("via"))'
timeout -k 10 300 $T --master-port 29511 bench.py --gpus 2 --backend gloo --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tee gpurun_out/r03_bench/rehearse_c4r_2rank_gloo.json | python -c "$P" C4r_gloo2 &&
timeout -k 10 300 $T --master-port 29512 bench.py --gpus 2 --backend gloo --workload C4opt --steps 2 --warmup 1 2>/dev/null | tee gpurun_out/r03_bench/rehearse_c4opt_2rank_gloo.json | python -c "$P" C4opt_gloo2 &&
timeout -k 10 300 $T --master-port 29513 bench.py --gpus 2 --backend gloo --workload C4 --via torch --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tee gpurun_out/r03_bench/rehearse_c4_2rank_gloo.json | python -c "$P" C4_gloo2 &&
timeout -k 10 300 $T --master-port 29514 bench.py --gpus 2 --backend gloo --workload C4 --via torch --plan hybrid --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "$P" C4_hybrid_gloo2 &&
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -x -q -k "parts" 2>&1 | tail -3
