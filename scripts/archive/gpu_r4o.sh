#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -q -p no:cacheprovider --timeout=300 -x -k "fit_predict or early_stopped or append or random_sweep or checkpoint" 2>&1 | tail -3
for V in 14=1 14=0 14=1 14=0; do
  echo "== cov_alone NNGP_DEBUG=$V"
  NNGP_DEBUG=$V timeout -k 10 300 python scripts/cov_alone.py 2>/dev/null | tail -1
done
CFGS="cfg3 cfg2" bash scripts/gpu_ab.sh 14=1 14=0 14=1 14=0
