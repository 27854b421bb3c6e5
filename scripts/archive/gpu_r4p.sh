#!/bin/bash
# round 4: z of a first residual rounded to three digit planes (12 plane products instead of 15); key 5 = 58 restores five
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_configs.py -m gpu -q -p no:cacheprovider --timeout=600 -x 2>&1 | tail -5 || exit 1
CFGS="cfg3 cfg2 cfg5" bash scripts/gpu_ab.sh 5=58 0=0 5=58 0=0
