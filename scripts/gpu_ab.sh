#!/bin/bash
# A/B timing of Cholesky variants on cfg3 (and correctness tests first).
export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
for V in "0=0" "2=1" "5=1" "4=16" "4=64" "1=2048" "1=2048,4=16"; do
  echo "== NNGP_DEBUG=$V"
  NNGP_DEBUG=$V timeout -k 10 300 python bench.py --config cfg3 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stages_ms'], d['fit_info'])"
done

