#!/bin/bash
# round 4, third GPU session: chain one step ahead (next-column launch split off the far chunk), inverses under the tail
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider --timeout=300 -k "cholesky or potrf or lookahead or blocked_solves or full_forest or large_size or lookahead_sizes" > gpurun_out/r4c_pytest.log 2>&1
echo "pytest exit=$?"; tail -5 gpurun_out/r4c_pytest.log
CFGS="cfg3" bash scripts/gpu_ab.sh 8=1,9=1 8=0 8=16 8=8 8=32 8=48 9=1 8=1,9=1 8=0 2>&1 | tee gpurun_out/r4c_ab.log
CFGS="cfg2 cfg5" bash scripts/gpu_ab.sh 8=1,9=1 8=0 2>&1 | tee -a gpurun_out/r4c_ab.log
bash scripts/gpu_trace.sh 8=0 2>&1 | tee gpurun_out/r4c_trace.log
find gpurun_out -name "*kernel_trace.csv" -size +30M -delete
