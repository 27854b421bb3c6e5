#!/bin/bash
# Round-2 GPU session A: full-size config tests, variance-level study, posterior A/B on the cfg3 bench.
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q -x -s -p no:cacheprovider --timeout=600 > gpurun_out/pytest_gpu_r2a.log 2>&1
echo "pytest exit=$?"; grep -a "CONFIG_CHECK\|passed\|failed\|Error" gpurun_out/pytest_gpu_r2a.log | cut -c1-600 | tail -12
VAR_STUDY_N=32768 timeout -k 10 300 python scripts/var_study.py > gpurun_out/var_study_r2a_32768.json 2> gpurun_out/var_study_r2a.err; cat gpurun_out/var_study_r2a_32768.json | tr -d "\n "; echo
VAR_STUDY_N=16384 timeout -k 10 300 python scripts/var_study.py > gpurun_out/var_study_r2a_16384.json 2>> gpurun_out/var_study_r2a.err; cat gpurun_out/var_study_r2a_16384.json | tr -d "\n "; echo
ab() {
  echo "== refine=$1 debug=$2 cfg=$3"
  NNGP_REFINE=$1 NNGP_DEBUG=$2 timeout -k 10 300 python bench.py --config $3 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stages_ms'], d['fit_info']['cg_iters'], d['fit_info']['alpha_l2'])"
}
ab 2 "0=0" cfg3
ab 1 "0=0" cfg3
ab 1 "0=32" cfg3
ab 2 "0=32" cfg3
ab 1 "0=32" cfg2
ab 2 "0=0" cfg2
ab 1 "0=32" cfg4
ab 2 "0=0" cfg4
