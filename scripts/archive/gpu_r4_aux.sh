#!/bin/bash
# round 4 (late): an NTK model's NNGP kernel written by the fit's own kernel build (key 5 = 65: a build of its own inside the predict)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 -x 2>&1 | tail -4 || exit 1
CFGS="cfg5" bash scripts/gpu_ab.sh 0=0 5=65 0=0 5=65
