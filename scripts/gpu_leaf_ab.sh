#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -x -q > gpurun_out/leaf_pytest.log 2>&1 || { tail -30 gpurun_out/leaf_pytest.log; exit 1; }
tail -2 gpurun_out/leaf_pytest.log
CFGS="cfg2 cfg3 cfg5" bash scripts/gpu_ab.sh "3=2" "3=0" "3=2" "3=0"
