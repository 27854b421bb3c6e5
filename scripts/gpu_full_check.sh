#!/bin/bash
# whole GPU suite, then the bench on the configs given (default cfg3)
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/full_pytest.log 2>&1 || { tail -40 gpurun_out/full_pytest.log; exit 1; }
tail -3 gpurun_out/full_pytest.log
for C in ${CFGS:-cfg3}; do
  python bench.py --config $C --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$C', d['ms_per_step'], d['stages_ms'], d['fit_info']['cg_iters'], d['fit_info']['alpha_l2'])"
done
