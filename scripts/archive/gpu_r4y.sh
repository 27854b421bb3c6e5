#!/bin/bash
# round 4: composite ReLU map A/B on a side-effect-free key (5 = 63: per-layer recursion), with the fit's CG figures
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for C in cfg3 cfg4; do
for V in 0=0 5=63 0=0 5=63; do
  echo "== $C NNGP_DEBUG=$V"
  NNGP_DEBUG=$V timeout -k 10 300 python bench.py --config $C --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stages_ms'], d['fit_info'])"
done
done
