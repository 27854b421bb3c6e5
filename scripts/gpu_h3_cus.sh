#!/bin/bash
# split-float16 GEMM time against the number of compute units its persistent grid uses (debug key 4 = units left free)
export TMPDIR=/tmp; mkdir -p gpurun_out; rm -rf gpurun_out/h3ab
H3_KEY=4 H3_VARIANTS=${1:-8,32,64,96,128} timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/h3ab -- python3 scripts/h3_ab.py > gpurun_out/h3ab.log 2>&1 || { tail -20 gpurun_out/h3ab.log; exit 1; }
python scripts/h3_ab_report.py | tee gpurun_out/h3_cus.txt
