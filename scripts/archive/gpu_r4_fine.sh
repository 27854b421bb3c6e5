#!/bin/bash
# round 4 (late): FINE int8 products with z rounded to five planes (key 5 = 64: seven) -- full suite, the hard-case accuracy study, cfg5 A/B
cd "$(dirname "$0")/.."
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 -x 2>&1 | tail -4 || exit 1
timeout -k 10 600 python scripts/i8s_hard_case.py > gpurun_out/r4_i8s_hard_cases.jsonl 2> gpurun_out/r4_i8s_hard_cases.err; echo "hard cases rc=$?"
CFGS="cfg5" bash scripts/gpu_ab.sh 0=0 5=64 0=0 5=64
