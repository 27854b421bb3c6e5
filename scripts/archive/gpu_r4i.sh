#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for V in 8=32,12=1 8=32,12=1,4=24 8=32,12=1,4=16 8=32,12=1,4=40 8=1,9=1,12=1 8=48,12=1; do
  echo "== $V"
  NNGP_DEBUG=$V timeout -k 10 300 python bench.py --config cfg3 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep -a "v4 timing\|ms_per_step" | tail -2 | cut -c1-330 | sed 's/"metric.*"ms_per_step"/ms_per_step/'
done
for C in cfg2 cfg5 cfg4; do
for V in 8=1 8=32; do
  echo "== $C $V"
  NNGP_DEBUG=$V timeout -k 10 400 python bench.py --config $C --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stages_ms'], d['fit_info']['cg_iters'])"
done
done
