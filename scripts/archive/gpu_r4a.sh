#!/bin/bash
# round 4, first GPU session: the new Cholesky schedule -- parity of the schedules, A/B timing, one timeline
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider --timeout=300 -k "cholesky or potrf or lookahead or right_looking" > gpurun_out/r4a_pytest.log 2>&1
echo "pytest exit=$?"; tail -5 gpurun_out/r4a_pytest.log
CFGS="cfg3" bash scripts/gpu_ab.sh 8=1 8=0 8=4 8=2 8=1 8=0 2>&1 | tee gpurun_out/r4a_ab.log
bash scripts/gpu_trace.sh 8=0 2>&1 | tee gpurun_out/r4a_trace.log
find gpurun_out -name "*kernel_trace.csv" -size +30M -delete
