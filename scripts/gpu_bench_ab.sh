#!/bin/bash
# A/B of environment switches on the default bench step: gpu_bench_ab.sh "<env assignments>" ["<env assignments>" ...]
R=$GRAFT_REPO_ROOT
for V in "$@"; do
  for rep in 1 2; do
    env $V python3 $R/bench.py --config cfg3 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$V', 'step %.2f' % d['ms_per_step'], 'stages', {k: round(v, 2) for k, v in d['stages_ms'].items() if v > 0.05}, 'solves %.2f' % d['roofline_solves']['ms_per_step'], flush=True)"
  done
done
