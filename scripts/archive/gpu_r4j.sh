#!/bin/bash
# round 4: defaults restored (round-3 Cholesky schedule, 1024-blocks), triangular diag multiplies; full GPU suite + A/B
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 -x > gpurun_out/r4j_pytest.log 2>&1
echo "pytest exit=$?"; tail -5 gpurun_out/r4j_pytest.log
CFGS="cfg3" bash scripts/gpu_ab.sh 9=4 9=0 9=4 9=0 9=2 2>&1 | tee gpurun_out/r4j_ab.log
CFGS="cfg2 cfg5" bash scripts/gpu_ab.sh 9=4 9=0 2>&1 | tee -a gpurun_out/r4j_ab.log
