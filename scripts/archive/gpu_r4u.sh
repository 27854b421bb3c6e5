#!/bin/bash
# round 4: the 150-case random parity sweep on the final tree (regression record: z rounded to three digit planes)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -f gpurun_out/parity_sweep.jsonl
NNGP_SWEEP_CASES=150 timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider --timeout=600 -k random_sweep 2>&1 | tail -4
cp gpurun_out/parity_sweep.jsonl gpurun_out/r4_parity_sweep_150.jsonl
