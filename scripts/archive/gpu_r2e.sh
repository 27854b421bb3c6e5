#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "kernel_build or fit_predict or edge or power_of_two" -p no:cacheprovider --timeout=300 > gpurun_out/pytest_gpu_r2e_k1.log 2>&1
echo "k1 pytest exit=$?"; tail -3 gpurun_out/pytest_gpu_r2e_k1.log | cut -c1-300
timeout -k 10 200 python scripts/k1_ablate.py 2>&1 | tail -1
timeout -k 10 300 python scripts/k1_study.py 2>/dev/null | tr -d "\n " | cut -c1-1500; echo
