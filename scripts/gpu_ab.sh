#!/bin/bash
# A/B timing via NNGP_DEBUG switches on bench configs (correctness tests first).
export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
for C in cfg3 cfg2; do
for V in "0=0" "7=2"; do
  echo "== $C NNGP_DEBUG=$V"
  NNGP_DEBUG=$V timeout -k 10 300 python bench.py --config $C --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stages_ms'], d['fit_info']['cg_iters'], d['fit_info']['alpha_l2'])"
done
done
