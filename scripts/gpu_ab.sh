#!/bin/bash
# A/B timing of NNGP_DEBUG variants on the bench (timing experiments only)
export TMPDIR=/tmp
CFGS=${CFGS:-cfg3}
for C in $CFGS; do
for V in "$@"; do
  echo "== $C NNGP_DEBUG=$V"
  NNGP_DEBUG=$V timeout -k 10 300 python bench.py --config $C --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stages_ms'], d['fit_info']['cg_iters'], d['fit_info']['alpha_l2'])"
done
done
