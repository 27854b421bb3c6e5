#!/bin/bash
# round 4: the large-N random sweep (N = 5000 .. 14000, NNGP and NTK) on the final tree -- regression record for the rounded z and the composite map
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -f gpurun_out/parity_sweep.jsonl
NNGP_SWEEP_CASES=24 NNGP_SWEEP_NMIN=5000 NNGP_SWEEP_NMAX=14000 NNGP_SWEEP_NTK=1 timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider --timeout=900 -k random_sweep 2>&1 | tail -4
cp gpurun_out/parity_sweep.jsonl gpurun_out/r4_parity_sweep_n5000_14000.jsonl
