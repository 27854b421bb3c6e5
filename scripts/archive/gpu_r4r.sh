#!/bin/bash
# round 4: the factor's inverted blocks built lazily / beside the cross-kernel build of the first predict (key 5 = 61: in line)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 -x 2>&1 | tail -5 || exit 1
CFGS="cfg3 cfg2" bash scripts/gpu_ab.sh 0=0 5=61 0=0 5=61
