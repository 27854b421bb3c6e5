#!/bin/bash
# round 4: fused combination pass (row statistics + float32 copy), one K chunk when <= 3 pairs per diagonal; A/B keys 5 = 59 / 60 / 58
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_configs.py -m gpu -q -p no:cacheprovider --timeout=600 -x 2>&1 | tail -5 || exit 1
CFGS="cfg3" bash scripts/gpu_ab.sh 0=0 5=59 5=60 5=58 0=0 5=59 5=60
CFGS="cfg2 cfg5" bash scripts/gpu_ab.sh 0=0 5=59 0=0 5=59
