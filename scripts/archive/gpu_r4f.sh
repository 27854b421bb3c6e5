#!/bin/bash
# round 4: hardware-queue oversubscription? HIP multiplexes streams onto GPU_MAX_HW_QUEUES (default 4) HSA queues
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for Q in 4 8 12; do
for V in 8=1,9=1 8=32 8=0 8=8; do
  echo "== GPU_MAX_HW_QUEUES=$Q $V"
  GPU_MAX_HW_QUEUES=$Q NNGP_DEBUG=$V timeout -k 10 300 python bench.py --config cfg3 --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stages_ms'], d['fit_info']['cg_iters'], d['roofline']['ms_per_step_in_kernel'])"
done
done
