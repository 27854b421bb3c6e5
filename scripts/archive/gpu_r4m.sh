#!/bin/bash
# round 4: full GPU suite on the consolidated tree + bench lines of every config
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 > gpurun_out/r4m_pytest.log 2>&1
echo "pytest exit=$?"; tail -4 gpurun_out/r4m_pytest.log
for C in cfg1 cfg2 cfg3 cfg5; do
  timeout -k 10 400 python bench.py --config $C --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r4_bench_$C.json 2> gpurun_out/r4_bench_$C.err || tail -3 gpurun_out/r4_bench_$C.err
  python -c "
import json,sys
d=json.loads(open('gpurun_out/r4_bench_$C.json').read().strip().splitlines()[-1]); print('$C', d['ms_per_step'], d['stages_ms'], d['fit_info']['cg_iters'], d['roofline'].get('frac'))"
done
