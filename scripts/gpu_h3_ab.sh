#!/bin/bash
# correctness of the split-float16 GEMM, then variant timings from a kernel trace, then the bench with the old / new kernel
export TMPDIR=/tmp; mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "h3 or split" > gpurun_out/h3ab_pytest.log 2>&1 || { tail -30 gpurun_out/h3ab_pytest.log; exit 1; }
tail -2 gpurun_out/h3ab_pytest.log
rm -rf gpurun_out/h3ab
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/h3ab -- python3 scripts/h3_ab.py > gpurun_out/h3ab.log 2>&1 || { tail -20 gpurun_out/h3ab.log; exit 1; }
python scripts/h3_ab_report.py
bash scripts/gpu_ab.sh "5=31" "5=0" "5=31" "5=0"
