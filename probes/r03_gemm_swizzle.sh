#!/bin/bash
set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_parity.py -x -q -k "gemm or complex or deriv" 2>&1 | tail -3
TAG=C4opt_only1 bash probes/r03_profile.sh C4opt --opt-only 1 || exit 1
