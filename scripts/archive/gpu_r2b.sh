#!/bin/bash
# Round-2 GPU session B: whole GPU suite on the new default (level-1 variance), variance study, bench lines, multi-rank rehearsal.
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -q -s -p no:cacheprovider --timeout=600 > gpurun_out/pytest_gpu_r2b.log 2>&1
echo "pytest exit=$?"; grep -a "CONFIG_CHECK\|passed\|failed\|^FAILED\|^ERROR" gpurun_out/pytest_gpu_r2b.log | cut -c1-700 | tail -20
VAR_STUDY_N=32768 timeout -k 10 300 python scripts/var_study.py > gpurun_out/var_study_r2b_32768.json 2> gpurun_out/var_study_r2b.err; cat gpurun_out/var_study_r2b_32768.json | tr -d "\n "; echo
for C in cfg3 cfg2 cfg5; do
  timeout -k 10 400 python bench.py --config $C --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_${C}_r2b.json 2> gpurun_out/bench_${C}_r2b.err || tail -5 gpurun_out/bench_${C}_r2b.err
  python -c "
import json,sys
d=json.loads(open('gpurun_out/bench_${C}_r2b.json').read().strip().splitlines()[-1]); print('$C', d['ms_per_step'], d['stages_ms'], d['roofline']['frac'], d['roofline_posterior']['frac'], d['roofline_k1']['frac'])"
done
bash scripts/gpu_multirank_rehearsal.sh cfg2 2>&1 | tail -8
