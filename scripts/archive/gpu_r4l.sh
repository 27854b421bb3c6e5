#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -q -p no:cacheprovider --timeout=300 -x -k "transposed_split or blocked_solves or triangular_inverse or int8_residual or active or cholesky or full_forest or append" > gpurun_out/r4l_pytest.log 2>&1
echo "pytest exit=$?"; tail -5 gpurun_out/r4l_pytest.log
CFGS="cfg3" bash scripts/gpu_ab.sh 9=8 9=0 9=8 9=0 2>&1 | tee gpurun_out/r4l_ab.log
CFGS="cfg2 cfg5 cfg4" bash scripts/gpu_ab.sh 9=8 9=0 2>&1 | tee -a gpurun_out/r4l_ab.log
