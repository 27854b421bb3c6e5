#!/bin/bash
# Round-2 GPU session D: the MFMA kernel build (K1 v2): parity tests, A/B against the round-1 kernel, bench.
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "kernel_build or fit_predict or edge or power_of_two" -p no:cacheprovider --timeout=300 > gpurun_out/pytest_gpu_r2d_k1.log 2>&1
echo "k1 pytest exit=$?"; tail -3 gpurun_out/pytest_gpu_r2d_k1.log | cut -c1-300
timeout -k 10 300 python scripts/k1_study.py > gpurun_out/k1_study_r2d.json 2> gpurun_out/k1_study_r2d.err; cat gpurun_out/k1_study_r2d.json | tr -d "\n " | cut -c1-1500; echo
timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 > gpurun_out/pytest_gpu_r2d.log 2>&1
echo "pytest exit=$?"; grep -a "passed\|failed\|^FAILED\|^ERROR" gpurun_out/pytest_gpu_r2d.log | cut -c1-300 | tail -12
for C in cfg3 cfg2 cfg5 cfg4; do
  timeout -k 10 400 python bench.py --config $C --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_${C}_r2d.json 2> gpurun_out/bench_${C}_r2d.err || tail -5 gpurun_out/bench_${C}_r2d.err
  python -c "
import json,sys
d=json.loads(open('gpurun_out/bench_${C}_r2d.json').read().strip().splitlines()[-1]); print('$C', d['ms_per_step'], d['stages_ms'], d['roofline']['frac'], d['roofline_posterior']['frac'], d['roofline_k1']['frac'])"
done
