#!/bin/bash
# round 4: where the deferred alpha CG starts (key 2: 0 = int8 product start, 8 = predict start, 9 = product end, 10 = backward solve start)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for V in 0=0 2=10 2=8 2=9 0=0 2=10; do
  echo "== cov_alone NNGP_DEBUG=$V"
  NNGP_DEBUG=$V timeout -k 10 300 python scripts/cov_alone.py 2>/dev/null | tail -1
done
CFGS="cfg3" bash scripts/gpu_ab.sh 0=0 2=10 2=8 0=0 2=10 2=8
