#!/bin/bash
# round 3: correctness of both split-float16 GEMM forms, then interleaved timings of H3_CONFIGS from a kernel trace
export TMPDIR=/tmp; mkdir -p gpurun_out; T=${1:-r3}
timeout -k 10 300 python3 scripts/h3_lab.py check > gpurun_out/${T}_h3lab_check.log 2>&1 || { tail -30 gpurun_out/${T}_h3lab_check.log; exit 1; }
tail -2 gpurun_out/${T}_h3lab_check.log
rm -rf gpurun_out/h3lab
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/h3lab -- python3 scripts/h3_lab.py time > gpurun_out/${T}_h3lab_time.log 2>&1 || { tail -20 gpurun_out/${T}_h3lab_time.log; exit 1; }
python3 scripts/h3_lab_report.py gpurun_out/h3lab | tee gpurun_out/${T}_h3lab_report.txt
