#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout=600 > gpurun_out/pytest_gpu_r2h.log 2>&1
echo "pytest exit=$?"; grep -a "passed\|failed\|^FAILED\|^ERROR" gpurun_out/pytest_gpu_r2h.log | cut -c1-300 | tail -8
ab() {
  NNGP_DEBUG=$1 timeout -k 10 300 python bench.py --config $2 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('debug=$1 $2', d['ms_per_step'], d['stages_ms'], d['fit_info']['cg_iters'], d['fit_info']['alpha_l2'])"
}
ab "2=0" cfg3
ab "2=5" cfg3
ab "2=4" cfg3
ab "2=0" cfg2
ab "2=0" cfg4
ab "2=5" cfg4
VAR_STUDY_N=32768 timeout -k 10 300 python scripts/var_study.py 2>/dev/null | tr -d "\n " | cut -c1-700; echo
