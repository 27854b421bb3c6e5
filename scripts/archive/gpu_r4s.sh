#!/bin/bash
# round 4: digit planes of K cut from the lower triangle (key 5 = 62: row by row)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -p no:cacheprovider --timeout=600 -x 2>&1 | tail -5 || exit 1
CFGS="cfg3 cfg5 cfg2" bash scripts/gpu_ab.sh 0=0 5=62 0=0 5=62
