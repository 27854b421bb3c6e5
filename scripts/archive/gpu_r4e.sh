#!/bin/bash
# round 4: why is the Cholesky 11 ms slower with the side-stream inverses when NOT profiled?
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for V in 8=32 8=0 8=0,10=3 8=0,11=1 8=0,11=2 8=0,10=12; do
  echo "== $V"
  NNGP_DEBUG=$V timeout -k 10 300 python bench.py --config cfg3 --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stages_ms'], d['fit_info']['cg_iters']); print('   h3', d['roofline'])"
done
