#!/bin/bash
# round 4: the composite ReLU map in the kernel build (key 3 = 5: per-layer recursion)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 -x 2>&1 | tail -6 || exit 1
CFGS="cfg3 cfg2 cfg4" bash scripts/gpu_ab.sh 0=0 3=5 0=0 3=5
