#!/bin/bash
export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
for C in cfg3 cfg2; do
for V in "0=0" "1=768" "1=1536"; do
  echo "== $C NNGP_DEBUG=$V"
  NNGP_DEBUG=$V timeout -k 10 300 python bench.py --config $C --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stages_ms'], d['fit_info']['cg_iters'], d['fit_info']['alpha_l2'])"
done
done
